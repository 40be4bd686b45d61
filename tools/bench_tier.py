"""Tiered hop at the Tiktok shape: full hop, long-rows-only hop, masked first adjoint hop, by ELIMREC_TIER_U / ELIMREC_TIER_LDSMASK
(dev tool, GPU)."""
import os, subprocess, sys
if len(sys.argv) == 1:
    subprocess.run([sys.executable, __file__, "x"])
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, slab
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
d, U, I = 64, 36656, 76085
ds = SyntheticDataset(U, I, 720829, feat_dims=(4, 4, 4), seed=0)
adj = create_adj_mat(*ds.get_train_interactions(), U, I, "pre").tocsr()
N = adj.shape[0]
def timeit(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
torch.manual_seed(0)
X = torch.randn(N, d, device=dev)
T = int(os.environ.get("T", 64))
plan = slab.SellPlan(adj, dev, threshold=T, side_split=U, tiered=True, ipw=8)
xs = slab.SlabTable(N, 2, 32, dev).from_rows(X)
y1, y2 = xs.like(), xs.like()
long_tab = torch.empty(2 * max(plan.n_long, 1) * 32, dtype=torch.float32, device=dev)
rows = torch.cat([torch.randperm(U)[:2048], U + torch.randperm(I)[:4096]]).numpy()
bits = np.zeros((N + 31) // 32 + 2, np.uint32)
np.bitwise_or.at(bits, rows >> 5, (np.uint32(1) << (rows & 31).astype(np.uint32)))
mask = torch.from_numpy(bits.view(np.int32)).to(dev)
src = xs.like(); src.data.zero_()
full = timeit(lambda: (slab.hop(plan, xs, y1, gs=2), slab.hop(plan, y1, y2, gs=2))) / 2
seg = timeit(lambda: slab.hop(plan, xs, long_tab, gs=2, seg_only=True))
msk = timeit(lambda: slab.hop(plan, src, y1, gs=2, src_mask=mask, add=xs, add_mask=mask, scale=1.0))
print("dbg=%s U=%s ldsmask=%s T=%d: full hop %.1f us, long rows only %.1f us, masked hop %.1f us  (wave rows %d, wg rows %d, segs %d)"
      % (os.environ.get("ELIMREC_TIER_DBG"), os.environ.get("ELIMREC_TIER_U"), os.environ.get("ELIMREC_TIER_LDSMASK"), T, full, seg, msk, plan.n_w1, plan.n_w4, plan.n_seg))
# where the time goes: the same graph without its long rows; the masked hop with an empty mask
import scipy.sparse as sp
deg = np.diff(adj.indptr)
keep = sp.diags((deg <= T).astype(np.float32)) @ adj
plan2 = slab.SellPlan(keep.tocsr(), dev, threshold=T, side_split=U, tiered=True, ipw=8)
short = timeit(lambda: (slab.hop(plan2, xs, y1, gs=2), slab.hop(plan2, y1, y2, gs=2))) / 2
zero = torch.zeros_like(mask)
m0 = timeit(lambda: slab.hop(plan, src, y1, gs=2, src_mask=zero, add=xs, add_mask=mask, scale=1.0))
m0s = timeit(lambda: slab.hop(plan2, src, y1, gs=2, src_mask=zero, add=xs, add_mask=mask, scale=1.0))
print("   short rows only (%.0f%% of the non-zeros): full hop %.1f us; masked hop with an empty mask %.1f us, same without the long rows %.1f us"
      % (100.0 * keep.nnz / adj.nnz, short, m0, m0s))
