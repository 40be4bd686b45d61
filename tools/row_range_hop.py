"""What ONE rank of a ROW-partitioned hop computes (the partition north_star names for the graph tables too, not
built): rank r owns the output rows r, r + W, r + 2W, ... of the propagation matrix (a balanced mix of user and item rows) and
gathers full-width rows of the whole source table -- a rectangular hop [N / W x N] . [N x d]. Time of that launch on one MI355X
for W = 1 / 2 / 4 / 8 beside the column shards' hop (all rows, d / W columns). usage: row_range_hop.py [d]  (SHAPE=c4: configs[3])"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, slab
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
U, I, E = (36656, 1217360, 16 * 720829) if os.environ.get("SHAPE") == "c4" else (36656, 76085, 720829)
ds = SyntheticDataset(U, I, E, feat_dims=(4, 4, 4), seed=0)
adj = create_adj_mat(*ds.get_train_interactions(), U, I, "pre").tocsr()
N = adj.shape[0]
torch.manual_seed(0)


def timed(fn, n=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for W in (1, 2, 4, 8):
    # row partition: rank 0's rows, all d columns
    ns, w = slab.choose_slabs(d, N)
    gs = slab.choose_groups(ns)
    ipw = 64 // ((ns // gs) * (w // 4))
    rows = adj[0::W].tocsr()
    rows.sort_indices()
    plan = slab.SellPlan(rows, dev, side_split=None, tiered=True, threshold=64, ipw=ipw)
    src = [slab.SlabTable(N, ns, w, dev).from_rows(torch.randn(N, d, device=dev)) for _ in range(2)]
    out = [slab.SlabTable(rows.shape[0], ns, w, dev) for _ in range(2)]
    k = [0]

    def row_hop():
        slab.hop(plan, src[k[0] & 1], out[k[0] & 1], gs=gs)
        k[0] += 1
    t_row = timed(row_hop)
    # column partition: all rows, d / W columns
    dl = d // W
    t_col = None
    if dl % 4 == 0 and dl >= 4:
        ns2, w2 = slab.choose_slabs(dl, N)
        gs2 = slab.choose_groups(ns2)
        plan2 = slab.SellPlan(adj, dev, side_split=U, tiered=True, threshold=64, ipw=64 // ((ns2 // gs2) * (w2 // 4)))
        tabs = [slab.SlabTable(N, ns2, w2, dev).from_rows(torch.randn(N, dl, device=dev)) for _ in range(2)]
        j = [0]

        def col_hop():
            slab.hop(plan2, tabs[j[0] & 1], tabs[1 - (j[0] & 1)], gs=gs2)
            j[0] += 1
        t_col = timed(col_hop)
    print("W=%d  row partition (%d of %d rows x %d columns, %d non-zeros): %.1f us   column partition (all rows x %d columns): %s"
          % (W, rows.shape[0], N, d, rows.nnz, t_row, dl, "%.1f us" % t_col if t_col else "-"), flush=True)
