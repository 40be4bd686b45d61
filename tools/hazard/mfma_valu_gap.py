"""Audit of the MFMA -> VALU hand-over in a gfx950 assembly listing: for every v_mfma, the wait states
along the straight-line path up to the first non-MFMA instruction that READS one of the registers it wrote (VGPR or AGPR),
counted the way LLVM's hazard recogniser counts them (every instruction 1, `s_nop N` N + 1). What GCNHazardRecognizer asks for
on gfx950 for "MFMA write VGPR -> VALU / memory / export read": XDL ops (the bf16 / f16 / i8 / fp8 shapes) passes + 3 + 1,
single-precision MFMAs (SMFMA) passes + 2 -- 8 states behind the 4-pass v_mfma_f32_16x16x32_bf16, 10 behind the 8-pass
v_mfma_f32_16x16x4_f32, 18 behind the 16-pass v_mfma_f32_32x32x2_f32. Prints per kernel and opcode the smallest gap found.

usage:  hipcc -O3 --offload-arch=gfx950 -S -o - --cuda-device-only -I include elimrec_amd/csrc/head.hip [flags of the Makefile] \
            | python tools/hazard/mfma_valu_gap.py            (dev tool; nothing of the product imports it)"""
import re, sys
from collections import defaultdict

NEED = {"v_mfma_f32_16x16x32_bf16": 8, "v_mfma_f32_16x16x4_f32": 10, "v_mfma_f32_32x32x2_f32": 18}
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")               # VGPRs and AGPRs (accumulators of the larger kernels)


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def main():
    kernel, block = None, []
    worst = defaultdict(lambda: [10 ** 9, 0, None])              # (kernel, opcode) -> [min gap, count, example]

    def flush():
        for i, (op, args) in enumerate(block):
            if not op.startswith("v_mfma"):
                continue
            dst = regs(args.split(",")[0])
            gap = 0
            for op2, args2 in block[i + 1:]:
                if op2 == "s_nop":
                    gap += int(args2.strip() or 0) + 1
                    continue
                parts = args2.split(",")
                reads = regs(",".join(parts[1:])) if not op2.startswith(("global_store", "buffer_store", "ds_write", "ds_store", "flat_store", "scratch_store")) else regs(args2)
                if reads & dst:
                    if op2.startswith("v_mfma"):
                        break                                       # SrcC chaining of the matrix pipe: its own (shorter) rule, not audited here
                    w = worst[(kernel, op)]
                    w[1] += 1
                    if gap < w[0]:
                        w[0], w[2] = gap, op2
                    break
                if regs(parts[0]) & dst and not op2.startswith("v_mfma"):
                    break                                           # overwritten before any read
                gap += 1
        block.clear()

    for line in sys.stdin:
        line = line.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(\S+):\s*$", line)
        if m:
            if not m.group(1).startswith((".L", "$")):
                flush()
                kernel = m.group(1)
            continue
        if not line.startswith(("\t", " ")) or line.strip().startswith("."):
            continue
        toks = line.strip().split(None, 1)
        op, args = toks[0], toks[1] if len(toks) > 1 else ""
        block.append((op, args))
        if op.startswith(("s_branch", "s_endpgm", "s_setpc")):          # (a conditional branch falls through: the path goes on)
            flush()
    flush()
    bad = 0
    for (k, op), (gap, n, ex) in sorted(worst.items(), key=lambda kv: kv[1][0]):
        need = NEED.get(op)
        mark = "" if need is None or gap >= need else "   <-- BELOW THE TABLE"
        bad += bool(mark)
        print("%3d wait states (need %s) over %4d hand-overs  %-28s first reader %-18s %s%s" % (gap, need, n, op, ex, (k or "?")[:90], mark))
    print("hand-overs below the table: %d" % bad)


if __name__ == "__main__":
    main()
