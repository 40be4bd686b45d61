"""How much of the d-column hop's time is L2 misses? Same adjacency structure (row lengths, nnz), columns folded into
a window of W source rows so the gathered table fits (or not) in the 4 MB XCD L2s. Dev tool."""
import os, sys, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, ops
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(4, 4, 4), seed=0)
tu, ti = ds.get_train_interactions()
adj = create_adj_mat(tu, ti, ds.num_users, ds.num_items, "pre").tocsr()
N, d = adj.shape[0], 64
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
for W in (N, 65536, 32768, 16384, 8192, 2048):
    m = adj.copy()
    m.indices = (m.indices % W).astype(np.int32)      # duplicates inside a row are fine for timing
    A = ops.Csr.from_scipy(m, dev, C=256, canonical=False) if "canonical" in ops.Csr.from_scipy.__code__.co_varnames else None
    if A is None:
        A = ops.Csr(torch.from_numpy(m.indptr.astype(np.int32)).to(dev), torch.from_numpy(m.indices).to(dev),
                    torch.from_numpy(m.data.astype(np.float32)).to(dev), N)
        A.build_split(256)
    t = timeit(lambda: ops.block_spmm(A, X, Xout=Y))
    print("source window %6d rows (%5.1f MB): %.1f us" % (W, W * d * 4 / 1e6, t))
