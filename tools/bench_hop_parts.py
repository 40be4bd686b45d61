"""Which part of the d-column hop costs what: the unsplit rows alone, the split (long) rows alone, both. Dev tool."""
import os, sys, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, ops
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(4, 4, 4), seed=0)
tu, ti = ds.get_train_interactions()
adj = create_adj_mat(tu, ti, ds.num_users, ds.num_items, "pre").tocsr()
N, d = adj.shape[0], 64
deg = np.diff(adj.indptr)
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(N, d, device=dev, generator=g); Y = torch.empty_like(X)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def make(m):
    A = ops.Csr(torch.from_numpy(m.indptr.astype(np.int32)).to(dev), torch.from_numpy(m.indices.astype(np.int32)).to(dev),
                torch.from_numpy(m.data.astype(np.float32)).to(dev), N)
    A.build_split(256)
    return A
long_mask = deg > 64
D_short = sp.diags((~long_mask).astype(np.float32)); D_long = sp.diags(long_mask.astype(np.float32))
parts = {"all rows": adj, "unsplit rows only (%d nnz)" % adj[~long_mask].nnz: (D_short @ adj).tocsr(),
         "split rows only (%d rows, %d nnz)" % (long_mask.sum(), adj[long_mask].nnz): (D_long @ adj).tocsr()}
for name, m in parts.items():
    m.eliminate_zeros()
    A = make(m)
    print("%-48s %.1f us" % (name, timeit(lambda: ops.block_spmm(A, X, Xout=Y))))
# no output write / no gather variants via a tiny matrix with the same row count
E = sp.csr_matrix((N, N), dtype=np.float32)
print("%-48s %.1f us" % ("empty matrix (row loop + output only)", timeit(lambda: ops.block_spmm(make(E), X, Xout=Y))))
