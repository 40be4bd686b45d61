"""Per-GPU device memory of a column-sharded job with lean tables, from the allocators' own formulas.

`plan()` adds up, term by term, what ColumnShardEngine.cs_setup / _workspace, EliMRec._workspace (lean form), lookup.py and
shard_eval.py allocate -- the same expressions those allocators use, fed with the shape instead of live tensors -- so that
a configuration can be sized before it is run (BASELINE.json configs[4]: 100 M items on 8 GPUs).
tests/test_capacity_gpu.py holds it against the device allocator on a shape that runs.

The index plan's size depends on the degree distribution (padding of the wave tiles, rows that get a wave / a workgroup /
segments): pass the numbers of a real plan (`plan_stats=stats_of(engine.plan)`) or let the estimate use the per-non-zero
ratios measured on the synthetic generator's graphs (PLAN_BYTES_PER_NNZ / PLAN_BYTES_PER_ROW).
"""
from . import ops, slab

PLAN_BYTES_PER_NNZ = 17.6      # tile_col + tile_val (padded, 8 B / entry) + csr_col + csr_val (8 B / non-zero); measured 17.2-17.6
PLAN_BYTES_PER_ROW = 13.0      # rowptr (8) + long_index (4) + tile records amortised
SWEEP_BYTES_PER_NNZ = 23.0     # column slices beyond the Infinity Cache (slab.sweep_wanted): 80-B step records of the user rows' half
                               # of the non-zeros (~11-12.5 B each at 0.8-0.9 fill) + a tile plan of the item rows' half (short rows:
                               # padded tiles, ~30 B each); measured 23.0 per non-zero of the whole graph at the scaled configs[4] shape


def stats_of(plan):
    """The data-dependent numbers of a live slab.SellPlan."""
    t = sum(v.numel() * v.element_size() for v in plan.t.values())
    parts = sum(v.numel() * v.element_size() for v in plan._partials.values())
    sweep = plan.sweep.device_bytes() if getattr(plan, "sweep", None) is not None else 0
    return dict(index_bytes=int(t), partial_bytes=int(parts), nnz=int(plan.nnz), sweep_bytes=int(sweep))


def plan(U, I, interactions, d, dims, world=1, layers=3, batch=2048, mods=None, symmetric=True, feature_dtype="f32",
         eval_users=8192, topk=10, plan_stats=None, fused_head=None, materialize_rows=1 << 18, wide=False):
    """Bytes per GPU, by component. interactions = unique (user, item) training pairs (the adjacency has twice as many
    non-zeros); dims = widths of the feature tables; batch = triplets per GPU and step. wide: an adjacency with a diagonal
    (adj_type norm / mean + I; N more non-zeros) -- the propagated tables are 2 dl columns wide and all L of them are kept."""
    N, W, L = U + I, int(world), int(layers)
    M = 1 + (len(dims) if mods is None else int(mods))
    C, Cy, dl = M * d, M * d, d // W
    nnz = 2 * int(interactions)
    R = 3 * batch
    sumD = sum(dims)
    es = 4 if feature_dtype == "f32" else 2
    row_bytes = ((sumD + (1 if es == 4 else 2)) * es + 15) // 16 * 16
    rows_loc = -(-U // W) + -(-I // W)
    out = {}
    # ---- ColumnShardEngine.cs_setup: slab-major tables of this rank's dl columns
    if wide:        # master x2, gradient, Adam m / v narrow; layer 0, X^1..X^L, the adjoint source [H | G], scratch x2 wide
        n_tables = 5 + 2 * (L + 4)
        nnz += N
    else:
        n_tables = 2 + (L - 1) + 1 + 2 + 2 + 2       # master x2, X^1..X^(L-1), gradient, Adam m / v, srcA / srcB, tmp x2
    out["graph tables (%d x [N x %d] fp32: master x2, layers, gradient, Adam m/v, adjoint sources, scratch)" % (n_tables, dl)] = n_tables * N * dl * 4
    out["hop-L table for evaluation (lazily, [N x %d])" % dl] = N * dl * 4
    out["row bitmap of the adjoint sources"] = ((N + 31) // 32 + 2) * 4
    if plan_stats is None:
        idx = int(PLAN_BYTES_PER_NNZ * nnz + PLAN_BYTES_PER_ROW * N)
        part = 0
    else:
        idx, part = plan_stats["index_bytes"], plan_stats["partial_bytes"]
    out["adjacency plan (wave tiles + CSR; replicated: every hop of a column slice needs the whole graph)"] = idx * (1 if symmetric else 2)
    out["split rows' partial sums"] = part
    if plan_stats is not None:
        sweep = plan_stats.get("sweep_bytes", 0)
    else:
        ns_, w_ = slab.choose_slabs(dl)
        sweep = int(SWEEP_BYTES_PER_NNZ * nnz) if (not wide and w_ in (16, 32) and slab.sweep_tiles_xcds(ns_)
                                                   and slab.sweep_wanted(N, dl, nnz)) else 0
    if sweep:
        out["window-sweep plan of the user rows + tile plan of the item rows (whole hops of a slice beyond the caches)"] = sweep * (1 if symmetric else 2)
    # ---- lookup.py: this rank's rows of [S_1 | .. | S_n | c]
    out["folded constants, my rows (%d x %d B, %s)" % (rows_loc, row_bytes, feature_dtype)] = rows_loc * row_bytes
    out["looked-up rows of a batch (fp32 [R x %d] + c + iota)" % sumD] = R * sumD * 4 + R * 8
    if W > 1:
        out["lookup exchange buffers (send: W x R rows worst case, recv: R rows)"] = (W * R + R) * row_bytes
        out["layer-mean / adjoint-source exchange (send + recv, forward + backward)"] = 4 * W * R * 2 * dl * 4 + W * R * 4
    # ---- EliMRec._workspace (lean): projection weights and the batch's rows
    p_tail = sum(d * D + d for D in dims) + 2 * (d * C + d) + 3 * (d * d + d)
    out["projection weights: parameters, gradients, snapshot, Adam m / v"] = 5 * p_tail * 4
    shapes = [(R, d, C), (R, d, C)] + [(R, d, d)] * (M - 1) + [(R, d, D) for D in dims]
    bwd_ws = ops.linear_bwd_w_batched_workspace(shapes)
    out["batch rows (OutAct, YAct, dY, dOutR, gradient rows, shared part, plans)"] = (
        R * C * 4 * 2 + R * Cy * 4 * 3 + R * d * 4 + 3 * batch * 2 * d * 4 + ops.segment_plan_workspace(R) + 6 * R * 4 + batch * 4)
    out["weight-gradient partial sums"] = bwd_ws
    if fused_head if fused_head is not None else d == 64:
        out["packed head weights"] = ops.head_pack_floats(list(dims)) * 4
    train = sum(out.values())
    # ---- shard_eval.py / _materialize_item_shard: the cached tables, item-sharded
    ev = {}
    i_loc = -(-I // W)
    ev["cached rows [all users ; my items] x Cy fp32 + squared block norms"] = (U + i_loc) * (Cy + M) * 4
    chunk = min(rows_loc, materialize_rows)
    ev["transient while materialising (chunks of %d of every owner's rows: Out, S in fp32, the column transpose; the users' Y gather)" % chunk] = (
        chunk * (C + sumD + 1 + d) * 4 + 2 * W * chunk * 2 * dl * 4 + (-(-U // W) * (W + 1) + U) * Cy * 4)
    ev["score workspace (%d users per launch, chunked top-%d)" % (eval_users, topk)] = ops.score_workspace(eval_users, U, i_loc, M - 1, topk, topk_only=True, d=d)
    return dict(components=out, training_bytes=int(train), eval_components=ev, eval_bytes=int(sum(ev.values())),
                total_bytes=int(train + sum(ev.values())), per_gpu_GiB=round((train + sum(ev.values())) / 2 ** 30, 2))


def table(p):
    """Markdown table of a plan()."""
    lines = ["| component | GiB |", "|---|---|"]
    for k, v in list(p["components"].items()) + list(p["eval_components"].items()):
        lines.append("| %s | %.2f |" % (k, v / 2 ** 30))
    lines.append("| **training total** | **%.2f** |" % (p["training_bytes"] / 2 ** 30))
    lines.append("| **with the evaluator's tables** | **%.2f** |" % (p["total_bytes"] / 2 ** 30))
    return "\n".join(lines)
