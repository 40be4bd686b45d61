"""d-column hop time as a function of the table width (column-slab what-if). Dev tool."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, ops
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(4, 4, 4), seed=0)
tu, ti = ds.get_train_interactions()
adj = create_adj_mat(tu, ti, ds.num_users, ds.num_items, "pre").tocsr()
N = adj.shape[0]
A = ops.Csr.from_scipy(adj, dev, C=256)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for d in (64, 32, 16, 128):
    X = torch.randn(N, d, device=dev); Y = torch.empty_like(X)
    print("d = %3d (%5.1f MB table): %.1f us" % (d, N * d * 4 / 1e6, timeit(lambda: ops.block_spmm(A, X, Xout=Y))))
