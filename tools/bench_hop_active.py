"""What a last forward hop restricted to the batch's active rows could cost: same hop kernel on the adjacency with all
other rows emptied (split rows kept / dropped). Dev tool."""
import os, sys, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, ops, PairwiseSamplerV2
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(4, 4, 4), seed=0)
tu, ti = ds.get_train_interactions()
adj = create_adj_mat(tu, ti, ds.num_users, ds.num_items, "pre").tocsr()
N, d, U = adj.shape[0], 64, ds.num_users
deg = np.diff(adj.indptr)
smp = PairwiseSamplerV2(ds, batch_size=2048, device=dev, seed=1)
u, p, n = (t[:2048].cpu().numpy() for t in smp.sample_epoch())
act = np.unique(np.concatenate([u, U + p, U + n]))
print("active rows %d, their nnz %d of %d (split rows among them: %d with %d nnz)" % (
    len(act), deg[act].sum(), adj.nnz, (deg[act] > 64).sum(), deg[act][deg[act] > 64].sum()))
X = torch.randn(N, d, device=dev); Y = torch.empty_like(X)
def timeit(fn, k=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k
def make(m):
    A = ops.Csr(torch.from_numpy(m.indptr.astype(np.int32)).to(dev), torch.from_numpy(m.indices.astype(np.int32)).to(dev),
                torch.from_numpy(m.data.astype(np.float32)).to(dev), N)
    A.build_split(256)
    return A
keep = np.zeros(N, dtype=np.float32); keep[act] = 1
keep_l = keep.copy(); keep_l[deg > 64] = 1
keep_s = keep.copy(); keep_s[deg > 64] = 0
for name, k in (("all rows", np.ones(N, dtype=np.float32)), ("active rows", keep), ("active + every split row", keep_l),
                ("active, unsplit only", keep_s)):
    m = (sp.diags(k) @ adj).tocsr(); m.eliminate_zeros()
    A = make(m)
    print("%-28s nnz %7d: %.1f us" % (name, m.nnz, timeit(lambda: ops.block_spmm(A, X, Xout=Y))))
