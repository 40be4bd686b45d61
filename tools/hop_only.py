"""N full hops of a slab-major table at the Tiktok shape (SHAPE=c4: Tiktok x16 items), for rocprofv3 (--pmc / --kernel-trace). Dev tool.
usage: hop_only.py [d] [hops]   env: SLAB_W / SLAB_GS choose the geometry (slab.SLAB_W_CAP / slab.SLAB_GROUPS)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, slab
from elimrec_amd.model import create_adj_mat
slab.SLAB_W_CAP = int(os.environ.get("SLAB_W", os.environ.get("ELIMREC_SLAB_W", "0"))) or 32
slab.SLAB_GROUPS = int(os.environ.get("SLAB_GS", os.environ.get("ELIMREC_SLAB_GS", "0")))
dev = "cuda:0"
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hops = int(sys.argv[2]) if len(sys.argv) > 2 else 12
U, I, E = (36656, 1217360, 16 * 720829) if os.environ.get("SHAPE") == "c4" else (36656, 76085, 720829)   # SHAPE=c4: BASELINE configs[3]
ds = SyntheticDataset(U, I, E, feat_dims=(4, 4, 4), seed=0)
adj = create_adj_mat(*ds.get_train_interactions(), U, I, "pre").tocsr()
N = adj.shape[0]
if os.environ.get("MIRROR") == "1":         # the two sides swapped (items first): SWEEP=1 then sweeps the ITEM rows over windows of USER rows
    import numpy as np
    perm = np.concatenate([np.arange(U, N), np.arange(U)])
    adj = adj[perm][:, perm].tocsr()
    adj.sort_indices()
    U, I = I, U
if os.environ.get("RELABEL") == "1":        # node ids = the plan's processing order: item rows then user rows, by decreasing degree
    import numpy as np
    deg = np.diff(adj.indptr)
    order = np.lexsort((-deg, np.arange(N) < U))          # same key as SellPlan's side_split order
    inv = np.empty(N, np.int64); inv[order] = np.arange(N)
    adj = adj[order][:, order].tocsr()
    adj.sort_indices()
phase, n_out = None, N
if os.environ.get("SPLIT"):                 # user rows cut into K virtual rows by item range: what an L2-sized source window would cost
    import numpy as np, scipy.sparse as sp
    K = int(os.environ["SPLIT"])
    coo = adj.tocoo()
    is_user_row = coo.row < U
    part = np.where(is_user_row, np.minimum(K - 1, (coo.col - U) * K // I), 0)
    new_row = np.where(part == 0, coo.row, N + (part - 1) * U + coo.row)
    n_out = N + (K - 1) * U
    adj = sp.csr_matrix((coo.data, (new_row, coo.col)), shape=(n_out, N))
    adj.sort_indices()
    phase = np.zeros(n_out, np.int64)
    phase[:U] = 1                                              # items (phase 0), then user parts 1..K
    for k in range(1, K):
        phase[N + (k - 1) * U:N + k * U] = 1 + k
ns, w = slab.choose_slabs(d, N)
gs = slab.choose_groups(ns)
tiered = os.environ.get("TIERED", "1") == "1"
T = int(os.environ.get("T", 64 if tiered else 32))
plan = slab.SellPlan(adj, dev, phase=phase, side_split=None if os.environ.get("RELABEL") == "1" else U, tiered=tiered, threshold=T, ipw=64 // ((ns // gs) * (w // 4)))
torch.manual_seed(0)
tabs = [slab.SlabTable(N, ns, w, dev).from_rows(torch.randn(N, d, device=dev)) for _ in range(3)]
if os.environ.get("SWEEP") == "1":           # the user rows by the window sweep (csrc/sweep.hip), the item rows by a tile plan of their own
    ref = slab.SlabTable(N, ns, w, dev)
    slab.hop(plan, tabs[0], ref, gs=gs)
    plan.sweep = slab.SweepPlan(plan, adj, U, dev, threshold=T, ipw=64 // ((ns // gs) * (w // 4)))
    got = slab.SlabTable(N, ns, w, dev)
    slab.hop(plan, tabs[0], got, gs=gs)
    a, b = ref.dense(), got.dense()
    print("sweep vs tile hop: max |diff| users %.3e items %.3e (max |value| %.3e); blocks %s window %d" % (
        (a[:U] - b[:U]).abs().max().item(), (a[U:] - b[U:]).abs().max().item(), a.abs().max().item(),
        {k: (round(v, 3) if isinstance(v, float) else v) for k, v in plan.sweep.geometry(ns, w).items() if not hasattr(v, "shape")}, plan.sweep.window(w)))
if n_out != N:                              # rectangular: every hop reads table 0 and writes an [n_out x d] table
    outs = [slab.SlabTable(n_out, ns, w, dev) for _ in range(2)]
src, dst = tabs[0], tabs[1]
def run(n):
    global src, dst
    for i in range(n):
        if n_out != N:
            slab.hop(plan, tabs[i % 3], outs[i % 2], gs=gs)
            continue
        slab.hop(plan, src, dst, gs=gs)
        src, dst = dst, (tabs[2] if dst is tabs[1] else tabs[1])
run(hops)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run(40)
e1.record()
torch.cuda.synchronize()
print("%.2f us per hop" % (e0.elapsed_time(e1) * 1e3 / 40))
if getattr(plan, "sweep", None) is not None:
    for name, fn in (("tile hop over the item rows", lambda: slab.hop(plan.sweep.items, tabs[0], tabs[1], gs=gs)),
                     ("window sweep over the user rows", lambda: plan.sweep.hop(tabs[0], tabs[1]))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20): fn()
        e1.record()
        torch.cuda.synchronize()
        print("  %s: %.2f us" % (name, e0.elapsed_time(e1) * 1e3 / 20))
if getattr(plan, "sweep", None) is not None and os.environ.get("EXPT") == "1":
    # where the in-situ item hop loses 100 us against its isolated timing: rotating tables, the other order, two streams
    def chain(fn, n):
        s_, d_ = tabs[0], tabs[1]
        for _ in range(n):
            fn(s_, d_)
            s_, d_ = d_, (tabs[2] if d_ is tabs[1] else tabs[1])
    side = torch.cuda.Stream()
    def both_streams(a, b):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            plan.sweep.hop(a, b)
        slab.hop(plan.sweep.items, a, b, gs=gs)
        main.wait_stream(side)
    for name, fn in (("item rows only, rotating tables", lambda a, b: slab.hop(plan.sweep.items, a, b, gs=gs)),
                     ("user rows only, rotating tables", lambda a, b: plan.sweep.hop(a, b)),
                     ("items then users", lambda a, b: (slab.hop(plan.sweep.items, a, b, gs=gs), plan.sweep.hop(a, b))),
                     ("users then items", lambda a, b: (plan.sweep.hop(a, b), slab.hop(plan.sweep.items, a, b, gs=gs))),
                     ("both on two streams", both_streams)):
        chain(fn, 4)
        torch.cuda.synchronize()
        e0.record()
        chain(fn, 30)
        e1.record()
        torch.cuda.synchronize()
        print("  EXPT %s: %.2f us per hop" % (name, e0.elapsed_time(e1) * 1e3 / 30))
if os.environ.get("MASKED"):                 # the first adjoint hop alone: a source table of which MASKED random rows are active
    import numpy as np
    if os.environ["MASKED"] == "batch":      # the rows of a sampled training batch of 2048 triplets (positives drawn by interaction)
        from elimrec_amd import PairwiseSamplerV2
        bu, bp, bn = PairwiseSamplerV2(ds, batch_size=2048, device=dev).sample_epoch()
        k = torch.cat([bu[:2048], U + bp[:2048], U + bn[:2048]]).unique().to(torch.int32)
        R = int(k.numel())
        keys = k.view(1, -1).contiguous()
    else:
        R = int(os.environ["MASKED"])
        keys = torch.from_numpy(np.random.default_rng(0).choice(N, R, replace=False).astype(np.int32)).to(dev).view(1, -1).contiguous()
    mask = torch.zeros((N + 31) // 32, dtype=torch.int32, device=dev)
    slab.rows_bitmap(keys, N, mask)
    slab.source_bits(plan, ns, w, gs, mask)
    forms = [("masked hop, source bits ready", plan, dict(src_mask=mask, bits_ready=True)), ("full hop", plan, {})]
    if getattr(plan, "sweep", None) is not None:
        slab.source_bits(plan.sweep.items, ns, w, gs, mask)
        forms.append(("masked tile hop over the ITEM rows only", plan.sweep.items, dict(src_mask=mask, bits_ready=True)))
    for name, pl, kw in forms:
        for _ in range(3): slab.hop(pl, tabs[0], tabs[1], gs=gs, **kw)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(40): slab.hop(pl, tabs[0], tabs[1], gs=gs, **kw)
        e1.record()
        torch.cuda.synchronize()
        print("  %s (%d active rows of %d): %.2f us" % (name, R, N, e0.elapsed_time(e1) * 1e3 / 40))
print("geometry ns=%d w=%d gs=%d, %d hops, index bytes %d" % (ns, w, gs, hops, plan.index_bytes()))
