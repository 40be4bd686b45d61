"""Does column-slab tiling of a wide half hop pay (L2 reuse of the source table)? Dev tool."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, ops
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(4, 4, 4), seed=0)
tu, ti = ds.get_train_interactions()
adj = create_adj_mat(tu, ti, ds.num_users, ds.num_items, "pre")
U, I, C = ds.num_users, ds.num_items, 256
P = ops.Csr.from_scipy(adj[:U, U:], dev, C=C)
Q = ops.Csr.from_scipy(adj[U:, :U], dev, C=C)
g = torch.Generator(device=dev).manual_seed(0)
Xu = torch.randn(U, C, device=dev, generator=g); Xi = torch.randn(I, C, device=dev, generator=g)
Ou = torch.empty(U, C, device=dev); Oi = torch.empty(I, C, device=dev)
Au = torch.randn(U, C, device=dev, generator=g); Ai = torch.randn(I, C, device=dev, generator=g)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, A, Xin, Xout, Add in (("Q (items<-users)", Q, Xu, Oi, Ai), ("P (users<-items)", P, Xi, Ou, Au)):
    ref = torch.empty_like(Xout)
    ops.block_spmm(A, Xin, Xout=ref, add1=Add, acc_out=Xout, scale=0.25)
    for slab in (256, 128, 64, 32):
        def run():
            for c0 in range(0, C, slab):
                ops.block_spmm(A, Xin[:, c0:c0 + slab], Xout=ref[:, c0:c0 + slab], add1=Add[:, c0:c0 + slab],
                               acc_out=Xout[:, c0:c0 + slab], scale=0.25)
        print("%s slab=%3d cols: %.1f us" % (name, slab, timeit(run)))
