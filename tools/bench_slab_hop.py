"""Slab-major hop vs the row-major hop at the Tiktok shape: correctness against block_spmm + timings per geometry.
Dev tool (GPU). usage: bench_slab_hop.py [d]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import SyntheticDataset, ops, slab
from elimrec_amd.model import create_adj_mat
dev = "cuda:0"
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
U, I = 36656, 76085
ds = SyntheticDataset(U, I, 720829, feat_dims=(4, 4, 4), seed=0)
tu, ti = ds.get_train_interactions()
adj = create_adj_mat(tu, ti, U, I, "pre").tocsr()
N = adj.shape[0]
A = ops.Csr.from_scipy(adj, dev, C=256)
plan = slab.SellPlan(adj, dev)
print("N %d nnz %d; SELL entries %d, items %d, segments %d, split rows %d" % (N, plan.nnz, plan.sell_entries, plan.n_items, plan.n_seg, plan.n_long))

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
Yref = torch.empty_like(X)
ops.block_spmm(A, X, Xout=Yref)
# a 3-table chain like the real forward: X -> T1 -> T2 -> T1 ...
T1, T2 = torch.empty_like(X), torch.empty_like(X)
def old_chain():
    ops.block_spmm(A, X, Xout=T1); ops.block_spmm(A, T1, Xout=T2); ops.block_spmm(A, T2, Xout=T1)
print("row-major hop d=%d: %.1f us (single table), %.1f us/hop in a 3-hop chain" % (d, timeit(lambda: ops.block_spmm(A, X, Xout=Yref)), timeit(old_chain) / 3))

long_mask = torch.zeros(N, dtype=torch.bool, device=dev)
long_mask[plan.t["long_rows"][:plan.n_long].long()] = True
for w in (4, 8, 16, 32, 64):
    if d % w: continue
    ns = d // w
    xs = slab.SlabTable(N, ns, w, dev).from_rows(X)
    assert torch.equal(xs.dense(), X)
    y1, y2 = xs.like(), xs.like()
    for gs in (8, 4, 2, 1):
        if ns % gs or (ns // gs) * (w // 4) > 64 or ((ns // gs) & (ns // gs - 1)): continue
        y1.data.fill_(float("nan"))
        slab.hop(plan, xs, y1, gs=gs)
        got = y1.dense()
        err = (got - Yref).abs().max().item()
        eq_short = torch.equal(got[~long_mask], Yref[~long_mask])
        def chain():
            slab.hop(plan, xs, y1, gs=gs); slab.hop(plan, y1, y2, gs=gs); slab.hop(plan, y2, y1, gs=gs)
        t1 = timeit(lambda: slab.hop(plan, xs, y1, gs=gs))
        t3 = timeit(chain) / 3
        print("slab w=%2d ns=%2d gs=%d (LPR %2d): %.1f us single, %.1f us/hop chain; max|diff| %.2e, unsplit rows bitwise %s"
              % (w, ns, gs, (ns // gs) * (w // 4), t1, t3, err, eq_short))
        assert err < 1e-5 and eq_short

# masked hop + epilogue: out = (A . (mask * S) + [mask] Add) * scale
w, ns = (8, d // 8)
n_act = 5500
act = torch.randperm(N, device=dev)[:n_act]
bm = torch.zeros(N, dtype=torch.bool, device=dev); bm[act] = True
words = torch.zeros((N + 31) // 32 + 2, dtype=torch.int32, device=dev)
bits = torch.zeros(words.numel() * 32, dtype=torch.int64, device=dev); bits[:N] = bm.long()
words.copy_((bits.view(-1, 32) << torch.arange(32, device=dev)).sum(1).to(torch.int32))   # wraps into the sign bit as intended
S = torch.randn(N, d, device=dev)
Sm = S * bm[:, None]
ref = torch.empty_like(X); ops.block_spmm(A, Sm, Xout=ref)
ref = (ref + Sm) * 0.25
ss = slab.SlabTable(N, ns, w, dev).from_rows(S)      # garbage outside the mask on purpose
out = ss.like()
for gs in (8, 1):
    out.data.fill_(float("nan"))
    slab.hop(plan, ss, out, gs=gs, src_mask=words, add=ss, add_mask=words, scale=0.25)
    err = (out.dense() - ref).abs().max().item()
    print("masked hop gs=%d: %.1f us; max|diff| %.2e" % (gs, timeit(lambda: slab.hop(plan, ss, out, gs=gs, src_mask=words, add=ss, add_mask=words, scale=0.25)), err))
    assert err < 1e-5

# layer means at listed rows, last hop inline
L = 3
x0 = slab.SlabTable(N, ns, w, dev).from_rows(X)
x1, x2, x3 = x0.like(), x0.like(), x0.like()
slab.hop(plan, x0, x1); slab.hop(plan, x1, x2); slab.hop(plan, x2, x3)
D = [t.dense() for t in (x0, x1, x2, x3)]
mean = (((D[0] + D[1]) + D[2]) + D[3]) * 0.25
nar = torch.cat([(D[0][:U] + D[2][:U]) * 0.25, (D[3][U:] + D[1][U:]) * 0.25])
R = 3 * 2048
rows_ = torch.sort(torch.randperm(N, device=dev)[:n_act])[0].int()
rows_ = torch.cat([rows_, torch.full((R - n_act,), -1, dtype=torch.int32, device=dev)])
cnt = torch.tensor([n_act], dtype=torch.int32, device=dev)
long_tab = torch.empty(ns * max(plan.n_long, 1) * w, device=dev)
out0 = torch.zeros(R, 4 * d, device=dev); narrow = torch.zeros(N, d, device=dev)
def rows_inline():
    slab.hop(plan, x2, long_tab, seg_only=True)
    slab.rows(plan, ns, w, L, U, [x0.data, x1.data, x2.data, None], long_tab, rows_, cnt, R, 1, out0[:, :d], narrow, True)
rows_inline()
r = rows_[:n_act].long()
e0 = (out0[:n_act, :d] - mean[r]).abs().max().item(); e1 = (narrow[r] - nar[r]).abs().max().item()
print("slab_rows (inline last hop at %d rows): %.1f us; bitwise out0 %s narrow %s (max diff %.1e %.1e)"
      % (n_act, timeit(rows_inline), torch.equal(out0[:n_act, :d], mean[r]), torch.equal(narrow[r], nar[r]), e0, e1))
assert e0 < 1e-6 and e1 < 1e-6
full0 = torch.empty(N, d, device=dev); fulln = torch.empty(N, d, device=dev)
slab.rows(plan, ns, w, L, U, [x0.data, x1.data, x2.data, x3.data], None, None, None, N, 1, full0, fulln, False)
print("slab_rows all rows: %.1f us; bitwise %s %s" % (timeit(lambda: slab.rows(plan, ns, w, L, U, [x0.data, x1.data, x2.data, x3.data], None, None, None, N, 1, full0, fulln, False)),
      torch.equal(full0, mean), torch.equal(fulln, nar)))
assert torch.equal(full0, mean) and torch.equal(fulln, nar)

# merge of two ranks' [H | G] rows
Rk = 4096
k0 = torch.sort(torch.randperm(N, device=dev)[:3000])[0].int(); k1 = torch.sort(torch.randperm(N, device=dev)[:3500])[0].int()
padk = lambda k: torch.cat([k, torch.full((Rk - k.numel(),), -(1 << 30), dtype=torch.int32, device=dev)])
keys = torch.cat([padk(k0), padk(k1)])
rws = torch.randn(2 * Rk, 2 * d, device=dev)
sa, sb = slab.SlabTable(N, ns, w, dev), slab.SlabTable(N, ns, w, dev)
sa.data.zero_(); sb.data.zero_()
mk = torch.zeros((N + 31) // 32 + 2, dtype=torch.int32, device=dev)
slab.merge_rows(rws, keys, 2, U, I, sa, sb, mk)
Hh = torch.zeros(N, d, device=dev); Gg = torch.zeros(N, d, device=dev)
Hh[k0.long()] += rws[:3000, :d]; Gg[k0.long()] += rws[:3000, d:]
Hh[k1.long()] += rws[Rk:Rk + 3500, :d]; Gg[k1.long()] += rws[Rk:Rk + 3500, d:]
refA = torch.cat([Hh[:U], Gg[U:]]); refB = torch.cat([Gg[:U], Hh[U:]])
print("merge_rows: %.1f us; SrcA ok %s SrcB ok %s" % (timeit(lambda: slab.merge_rows(rws, keys, 2, U, I, sa, sb, mk)), torch.equal(sa.dense(), refA), torch.equal(sb.dense(), refB)))
assert torch.equal(sa.dense(), refA) and torch.equal(sb.dense(), refB)
print("transposes: from_rows %.1f us, to_rows %.1f us" % (timeit(lambda: x0.from_rows(X)), timeit(lambda: x0.to_rows(T1))))

# column shards (multi-GPU per-rank work): dl columns of the d
for dl in (8, 16, 32):
    if dl >= d: continue
    nsl, wl = slab.choose_slabs(dl, N)
    xs = slab.SlabTable(N, nsl, wl, dev).from_rows(X, col0=dl)
    y1, y2 = xs.like(), xs.like()
    slab.hop(plan, xs, y1)
    err = (y1.dense() - Yref[:, dl:2 * dl]).abs().max().item()
    def chain():
        slab.hop(plan, xs, y1); slab.hop(plan, y1, y2); slab.hop(plan, y2, y1)
    print("column shard dl=%2d (ns=%d w=%d gs=%d): %.1f us/hop chain; max|diff| %.2e" % (dl, nsl, wl, slab.choose_groups(nsl), timeit(chain) / 3, err))
    assert err < 1e-5
print("ALL OK")
