#!/bin/bash
# On the GPU box: counters of the step's O(B) kernels at a large batch (B env, default 32768): where their time goes.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_bigb; rm -rf $O; mkdir -p $O
export B=${B:-32768}
P="python3 $R/tools/step_trace.py 14"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/mem -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/inst -o p -- $P > /dev/null 2>&1 < /dev/null
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_bigb"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "mem", "fetch", "write", "inst"):
    for f in glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("elimrec::", "")[:40]
            if any(t in k for t in ("head_", "bpr_head", "sell_tier_bwdw", "sell_tier_adam")):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v[-8:]) / len(v[-8:]) for n, v in c.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    print("%-40s dur~%.0f us  wave_cyc/CU/dur %.1f  wait_inst %.2f  active_any %.2f  valu %.2f vmem %.2f lds %.2f  mfma_busy %.3f  TA %.2f  tcp_stall %.2f  L2hit %.2f  fetch %.0f MB write %.0f MB  waves %d  lds_conf %.2f" % (
        k, gui / 2400.0, m.get("SQ_WAVE_CYCLES", 0) / 256 / (m.get("SQ_BUSY_CYCLES", 1) / 32), m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
        m.get("SQ_ACTIVE_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
        m.get("SQ_ACTIVE_INST_VMEM", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_ACTIVE_INST_LDS", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
        m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (m.get("SQ_BUSY_CYCLES", 1) / 32), m.get("TA_BUSY_avr", 0) / gui, m.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / gui,
        m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1), 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e6, m.get("WRITE_SIZE", 0) * 1024 / 1e6,
        m.get("SQ_WAVES", 0), m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
