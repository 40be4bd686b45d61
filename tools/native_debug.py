import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from helpers import build_model_from_fixture, load_golden
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
DEV = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "gcmc"
g = load_golden(name)
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
base = [tuple(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")) for t in (1, 2)]
n0 = min(len(b[0]) for b in base)
gen = torch.Generator().manual_seed(0)
batches = []
for s in range(40):
    b = base[s % 2]
    perm = torch.randperm(n0, generator=gen).to(DEV)
    size = n0 - 5 if (s == 25 and os.environ.get("RAGGED", "1") == "1") else n0
    batches.append(tuple(x[:n0][perm][:size].clone() for x in b))
out = {}
for native in ("0", "1"):
    os.environ["ELIMREC_NATIVE_STEP"] = native
    model, _ = build_model_from_fixture(g, DEV)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt)
    losses = []
    for b in batches:
        losses.append(tr.step(*b).clone())
        if os.environ.get("SYNC", "1") == "1":
            torch.cuda.synchronize()
    out[native] = torch.stack(losses).cpu().numpy()
    st = tr._native_state()
    print(native, st["failed"], st["native_steps"])
    if native == "1":
        for key, progs in st["programs"].items():
            for p in progs:
                print(key[0], {k: v for k, v in p.slots.items()}, p.n_ops)
d = out["0"] != out["1"]
print("first differing step:", np.nonzero(d)[0][:5], out["0"][d][:3], out["1"][d][:3])
