"""Stress: native-program steps vs Python-issued steps, many repetitions, dirty allocator (dev tool)."""
import os, sys
import numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from helpers import build_model_from_fixture, load_golden
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
DEV = "cuda:0"
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
bad = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["gcmc", "ml3", "kwai"]):
        g = load_golden(name)
        junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(8)]      # dirty the allocator's free lists
        del junk
        base = [tuple(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")) for t in (1, 2)]
        n0 = min(len(b[0]) for b in base)
        gen = torch.Generator().manual_seed(rep)
        batches = []
        for s in range(120):
            b = base[s % 2]
            perm = torch.randperm(n0, generator=gen).to(DEV)
            size = n0 - 5 if s in (25, 60, 61) else n0
            batches.append(tuple(x[:n0][perm][:size].clone() for x in b))
        out = {}
        for native in ("0", "1"):
            os.environ["ELIMREC_NATIVE_STEP"] = native
            model, _ = build_model_from_fixture(g, DEV)
            opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
            eng = ColumnShardEngine(model)
            tr = ColumnShardTrainer(eng, opt)
            out[native] = torch.stack([tr.step(*b) for b in batches]).cpu().numpy()
        d = np.nonzero(out["0"] != out["1"])[0]
        if len(d):
            bad += 1
            print("MISMATCH", rep, name, "first steps", d[:6], out["0"][d[:3]], out["1"][d[:3]], flush=True)
print("done, mismatching runs:", bad)
