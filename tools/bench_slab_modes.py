"""Hop time by launch form (0: hop + fix-up kernels, 1: persistent with in-launch combine, 2: persistent + fix-up launch),
split threshold and geometry (dev tool, GPU)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, ops, slab, _lib
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
d = 64
U, I = 36656, 76085
ds = SyntheticDataset(U, I, 720829, feat_dims=(4, 4, 4), seed=0)
adj = create_adj_mat(*ds.get_train_interactions(), U, I, "pre").tocsr()
N = adj.shape[0]
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
torch.manual_seed(0)
X = torch.randn(N, d, device=dev)
lib = _lib.load()
geoms = [("bf16 w32 gs2", 32, 2, 2, 0, True), ("bf16 w16 gs4", 16, 4, 4, 0, True), ("f32 w32 gs2", 32, 2, 2, 0, False), ("f32 w64 gs1", 64, 1, 1, 0, False), ("bf16 w64 gs1", 64, 1, 1, 0, True),
         ("f32 shard8", 8, 1, 1, 8, False), ("f32 shard16", 16, 1, 1, 16, False), ("bf16 shard8", 8, 1, 1, 8, True)]
for T in (64, 32):
    for name, w, ns, gs, shard, bf in geoms:
        plan = slab.SellPlan(adj, dev, threshold=T, side_split=U, tiered=False)
        ipw = 64 // max(1, (ns // gs) * (w // (8 if bf else 4)))
        tplan = slab.SellPlan(adj, dev, threshold=T, side_split=U, tiered=True, ipw=ipw)
        xs = slab.SlabTable(N, ns, w, dev).from_rows(X, col0=shard)
        if bf: xs = xs.to_bf16(xs.like(torch.bfloat16))
        y1, y2 = xs.like(), xs.like()
        res = []
        for mode in (0, 1, 2):
            lib.elimrec_slab_set_stream(mode)
            def chain():
                slab.hop(plan, xs, y1, gs=gs); slab.hop(plan, y1, y2, gs=gs); slab.hop(plan, y2, y1, gs=gs)
            res.append(timeit(chain) / 3)
        def tchain():
            slab.hop(tplan, xs, y1, gs=gs); slab.hop(tplan, y1, y2, gs=gs); slab.hop(tplan, y2, y1, gs=gs)
        res.append(timeit(tchain) / 3)
        print("T=%2d %-12s us/hop by form [2 kernels, in-launch, persistent+fixup, TIERED one launch] = %s  (segs %d -> %d, wave rows %d, wg rows %d)"
              % (T, name, " ".join("%.1f" % t for t in res), plan.n_seg, tplan.n_seg, tplan.n_w1, tplan.n_w4))
