"""Hop time vs work-item order / split threshold / inner-loop variant (dev tool, GPU)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, ops, slab, _lib
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
d = 64
U, I = 36656, 76085
ds = SyntheticDataset(U, I, 720829, feat_dims=(4, 4, 4), seed=0)
tu, ti = ds.get_train_interactions()
adj = create_adj_mat(tu, ti, U, I, "pre").tocsr()
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
A = ops.Csr.from_scipy(adj, dev, C=256)
Yref = torch.empty_like(X); ops.block_spmm(A, X, Xout=Yref)
lib = _lib.load()
geoms = [(64, 1, 1, 0), (32, 2, 2, 0), (16, 4, 4, 0), (8, 1, 1, 8), (16, 1, 1, 16)]   # (w, ns, gs, col0>0: column shard)
for T in (64, 32, 16):
    for side in (None, U):
        plan = slab.SellPlan(adj, dev, threshold=T, side_split=side)
        for (w, ns, gs, shard) in geoms:
            xs = slab.SlabTable(N, ns, w, dev).from_rows(X, col0=shard)
            y1, y2 = xs.like(), xs.like()
            res = []
            for variant in (0, 1, 2, 3):
                lib.elimrec_slab_set_variant(variant)
                slab.hop(plan, xs, y1, gs=gs)
                err = (y1.dense() - Yref[:, shard:shard + ns * w]).abs().max().item()
                assert err < 1e-5, err
                def chain():
                    slab.hop(plan, xs, y1, gs=gs); slab.hop(plan, y1, y2, gs=gs); slab.hop(plan, y2, y1, gs=gs)
                res.append(timeit(chain) / 3)
            print("T=%2d side=%-5s w=%2d ns=%d gs=%d%s: us/hop by variant [U8, U4+pf, U8+pf, U4] = %s   (segs %d)"
                  % (T, side is not None, w, ns, gs, " shard" if shard else "      ", " ".join("%.1f" % t for t in res), plan.n_seg))
