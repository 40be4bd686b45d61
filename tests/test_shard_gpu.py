"""The column-sharded (slab-major) engine on a real MI355X, through the C ABI: op-level checks of the slab kernels
against torch, the trainer at world 1 against the reference's golden vectors, W ranks emulated on one GPU against
the single-rank step on the concatenated batch, and one step at the C4 shape (Tiktok x16 items, d = 128) against the
oracle. Needs a GPU: `-m gpu`."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from helpers import assert_grad_close, build_model_from_fixture, load_golden, make_config, rel_err, sub

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _random_graph(n, seed, hot=6, hot_deg=400):
    """Ragged rows: empty rows, rows of a few entries and `hot` rows far above the split threshold."""
    rng = np.random.RandomState(seed)
    deg = rng.poisson(6, n)
    deg[rng.rand(n) < 0.05] = 0
    deg[rng.choice(n, hot, replace=False)] = hot_deg + rng.randint(0, 70, hot)
    rows = np.repeat(np.arange(n), deg)
    cols = rng.randint(0, n, len(rows))
    m = sp.csr_matrix((rng.rand(len(rows)).astype(np.float32) + 0.1, (rows, cols)), shape=(n, n))
    m.sum_duplicates()
    m.sort_indices()
    return m


def _bitmap(flags):
    n = flags.numel()
    words = torch.zeros((n + 31) // 32 + 2, dtype=torch.int64, device=DEV)
    bits = torch.zeros(words.numel() * 32, dtype=torch.int64, device=DEV)
    bits[:n] = flags.long()
    words.copy_((bits.view(-1, 32) << torch.arange(32, device=DEV)).sum(1))
    return words.to(torch.int32)                      # bit 31 wraps into the sign bit


@pytest.mark.parametrize("d,w,gs", [(64, 32, 2), (64, 8, 8), (64, 64, 1), (16, 16, 1), (8, 8, 1), (4, 4, 1), (128, 32, 4),
                                    (128, 32, 1), (48, 16, 3), (96, 32, 3)])
def test_slab_hop_vs_torch(d, w, gs):
    """elimrec_slab_hop on every lane-group width (1..64 lanes per work item), plain and with the adjoint's options
    (row-sparse source behind a bitmap, masked addend, scale): 1e-5 against an fp64 product; unsplit rows bitwise equal
    to the row-major kernel (same fmaf chain)."""
    from elimrec_amd import ops, slab
    n = 3000
    m = _random_graph(n, d + w)
    ns = d // w
    if ns % gs or (ns // gs) * (w // 4) > 64:
        pytest.skip("geometry not representable")
    plan = slab.SellPlan(m, DEV, threshold=32, side_split=1200)
    assert plan.n_long > 0
    torch.manual_seed(d)
    X = torch.randn(n, d, device=DEV)
    xs = slab.SlabTable(n, ns, w, DEV).from_rows(X)
    assert torch.equal(xs.dense(), X)
    y = xs.like()
    y.data.fill_(float("nan"))
    slab.hop(plan, xs, y, gs=gs)
    A64 = torch.from_numpy(m.astype(np.float64).toarray()).to(DEV)
    want = A64 @ X.double()
    got = y.dense()
    assert (got.double() - want).abs().max().item() < 1e-5
    if d % 4 == 0 and d <= 256:
        ref = torch.empty_like(X)
        ops.block_spmm(ops.Csr.from_scipy(m, DEV, C=d, threshold=32), X, Xout=ref)
        short = torch.from_numpy(np.diff(m.indptr) <= 32).to(DEV)
        assert torch.equal(got[short], ref[short])
    # adjoint form
    act = torch.rand(n, device=DEV) < 0.07
    bm = _bitmap(act)
    S = torch.randn(n, d, device=DEV)
    ss = slab.SlabTable(n, ns, w, DEV).from_rows(S)          # garbage outside the bitmap on purpose
    y.data.fill_(float("nan"))
    slab.hop(plan, ss, y, gs=gs, src_mask=bm, add=ss, add_mask=bm, scale=0.25)
    Sm = S.double() * act[:, None]
    assert (y.dense().double() - (A64 @ Sm + Sm) * 0.25).abs().max().item() < 1e-5


def _bipartite(U, I, seed, hot=5):
    """A symmetric bipartite graph [[0, R], [R^T, 0]]: ragged user rows (some empty, a few hot), Zipf-ish items, fp32 weights."""
    rng = np.random.RandomState(seed)
    deg = rng.poisson(40, U)
    deg[rng.rand(U) < 0.03] = 0
    deg[rng.choice(U, hot, replace=False)] = 900 + rng.randint(0, 90, hot)
    r = np.repeat(np.arange(U), deg)
    c = np.minimum((I * rng.rand(len(r)) ** 1.7).astype(np.int64), I - 1)
    R = sp.csr_matrix((rng.rand(len(r)).astype(np.float32) + 0.1, (r, c)), shape=(U, I))
    R.sum_duplicates()
    m = sp.bmat([[None, R], [R.T, None]], format="csr").astype(np.float32)
    m.sort_indices()
    return m


@pytest.mark.parametrize("d,w,U,I,window", [(128, 32, 3000, 20000, 1024), (64, 32, 700, 9000, 512), (256, 32, 2100, 6000, 4096),
                                             (512, 32, 300, 5000, 700), (32, 16, 900, 7000, 333), (16, 16, 40000, 3000, 256)])
def test_window_sweep_hop_equals_the_tile_hop(d, w, U, I, window, monkeypatch):
    """elimrec_slab_sweep_hop (csrc/sweep.hip) + the tile hop over the item rows = the tile hop over all rows: the item rows bit
    for bit (same kernel, same tiles), the user rows to fp32 round-off of another fixed summation order (a row's neighbours in
    column order, one fmaf chain) -- which is checked exactly against that order evaluated in fp64-free torch arithmetic on a few
    rows --, with the adjoint's add / add_mask / scale epilogue, at 1 / 2 / 4 / 8 / 16 slabs (8, 4, 2, 1 row parts per slab, several
    slabs per XCD role), row blocks beyond one pass (U = 40 000 at 16-float pieces), windows that do not divide the range, and
    bitwise equal from launch to launch."""
    from elimrec_amd import slab
    monkeypatch.setenv("ELIMREC_SWEEP_WINDOW", str(window))
    n = U + I
    m = _bipartite(U, I, d + w)
    ns = d // w
    gs = slab.choose_groups(ns)
    ipw = 64 // ((ns // gs) * (w // 4))
    plan = slab.SellPlan(m, DEV, threshold=64, side_split=U, tiered=True, ipw=ipw)
    torch.manual_seed(U)
    X = torch.randn(n, d, device=DEV)
    xs = slab.SlabTable(n, ns, w, DEV).from_rows(X)
    ref = xs.like()
    slab.hop(plan, xs, ref, gs=gs)
    plan.sweep = slab.SweepPlan(plan, m, U, DEV, threshold=64, ipw=ipw)
    got = xs.like()
    got.data.fill_(float("nan"))
    slab.hop(plan, xs, got, gs=gs)
    a, b = ref.dense(), got.dense()
    assert torch.equal(a[U:], b[U:])
    assert (a[:U] - b[:U]).abs().max().item() < 2e-5 * max(1.0, a.abs().max().item())
    g = plan.sweep.geometry(ns, w)
    assert g["parts"] == 8 // min(ns, 8) and g["rows"] <= 1247 * (32 // w) + 1 and (U <= 1247 * 32 * g["parts"] or g["passes"] > 1)
    # the stated order, exactly: fmaf over the neighbours in column order
    rows = [0, U // 3, U - 1] + np.argsort(-np.diff(m.indptr[:U + 1]))[:2].tolist()
    for r in rows:
        cols, vals = m.indices[m.indptr[r]:m.indptr[r + 1]], m.data[m.indptr[r]:m.indptr[r + 1]]
        acc = np.zeros(d, np.float32)
        Xh = X[torch.from_numpy(cols.astype(np.int64)).to(DEV)].cpu().numpy() if len(cols) else np.zeros((0, d), np.float32)
        for v, x in zip(vals, Xh):
            acc = (np.float64(v) * x.astype(np.float64) + acc.astype(np.float64)).astype(np.float32)      # fmaf: one rounding
        assert np.array_equal(b[r].cpu().numpy(), acc), r
    # the adjoint's epilogue, and a second launch of the same call
    act = torch.rand(n, device=DEV) < 0.3
    bm = _bitmap(act)
    S = torch.randn(n, d, device=DEV)
    ss = slab.SlabTable(n, ns, w, DEV).from_rows(S)
    want = (torch.from_numpy(m.astype(np.float64).toarray()).to(DEV) @ X.double() + S.double() * act[:, None]) * 0.25 if n <= 30000 else None
    y1, y2 = xs.like(), xs.like()
    slab.hop(plan, xs, y1, gs=gs, add=ss, add_mask=bm, scale=0.25)
    slab.hop(plan, xs, y2, gs=gs, add=ss, add_mask=bm, scale=0.25)
    assert torch.equal(y1.data, y2.data)
    if want is not None:
        assert (y1.dense().double() - want).abs().max().item() < 1e-4
    assert torch.equal(y1.dense()[:U], ((b[:U] + S[:U] * act[:U, None]) * 0.25))


def _plans_equal(host, devp):
    assert (host.n_rows, host.n_src, host.nnz, host.n_seg, host.n_long, host.n_w1, host.n_w4, host.n_tiles, host.tile_groups,
            host.sell_entries, host.sell_seg_entries) == (devp.n_rows, devp.n_src, devp.nnz, devp.n_seg, devp.n_long, devp.n_w1, devp.n_w4,
                                                          devp.n_tiles, devp.tile_groups, devp.sell_entries, devp.sell_seg_entries)
    for k in ("long_rows", "long_seg_ptr", "long_index", "tile_off", "tile_len", "tile_dst", "tile_long", "tile_col", "tile_val", "rowptr",
              "csr_col", "csr_val"):
        a, b = host.t[k], devp.t[k]
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), k
    for f, _ in host.desc._fields_:
        if not f.startswith("d_"):
            assert getattr(host.desc, f) == getattr(devp.desc, f), f


@pytest.mark.parametrize("kind,n,U,ipw,T,split,rows_from", [("random", 2500, 900, 8, 64, 900, 0), ("random", 2500, 900, 8, 32, None, 0),
                                                            ("random", 3000, 1, 16, 32, 1, 0), ("random", 700, 300, 32, 32, 300, 0),
                                                            ("bipartite", 9700, 700, 8, 64, 700, 0), ("bipartite", 9700, 700, 8, 64, None, 700),
                                                            ("bipartite", 23000, 3000, 4, 64, 3000, 0), ("hot", 4000, 2000, 8, 64, 2000, 0)])
def test_device_plan_build_equals_the_host_plan(kind, n, U, ipw, T, split, rows_from):
    """elimrec_plan_rows / _tiles / _scatter (csrc/plan.hip: rocPRIM sorts and scans + one thread per row / segment / tile group) on a
    device CSR give the arrays, counts and descriptor of slab.SellPlan(tiered=True) -- the host's numpy build -- bit for bit: ragged
    rows in every tier (short, a wave, a workgroup, segments), empty rows, several lane-group counts, with and without the side
    order, a plan of the rows from rows_from on (the window sweep's item-row plan); and a hop on the device-built plan equals a hop
    on the host-built one bitwise."""
    from elimrec_amd import slab
    if kind == "random":
        m = _random_graph(n, n + ipw, hot=9, hot_deg=300 * ipw)
    elif kind == "hot":
        m = _random_graph(n, 7, hot=4, hot_deg=20000)
    else:
        m = _bipartite(U, n - U, n)
    host = slab.SellPlan(m, DEV, threshold=T, side_split=split, tiered=True, ipw=ipw, rows_from=rows_from)
    rp, col, val = _t(m.indptr.astype(np.int64)), _t(m.indices.astype(np.int32)), _t(m.data.astype(np.float32))
    devp = slab.SellPlan.on_device(rp, col, val, m.shape[1], threshold=T, side_split=split, ipw=ipw, rows_from=rows_from)
    _plans_equal(host, devp)
    if rows_from == 0:
        d = 256 // ipw if 256 // ipw <= 64 else 64
        ns, w = slab.choose_slabs(d)
        if 64 // ((ns // slab.choose_groups(ns)) * (w // 4)) == ipw:
            X = slab.SlabTable(n, ns, w, DEV).from_rows(torch.randn(n, d, device=DEV))
            ya, yb = X.like(), X.like()
            slab.hop(host, X, ya)
            slab.hop(devp, X, yb)
            assert torch.equal(ya.data, yb.data)


def test_device_plan_build_at_the_tiktok_and_c4_shapes():
    """The same equality on the graphs of BASELINE.json configs[1] and configs[3] (18 M non-zeros, |I| = 1.2 M), the device CSR
    coming from csrc/adj.hip (elimrec_build_adj) as it would for a graph that never exists on the host; the configs[3] plan is
    built in well under half a second (< 0.5 s), the host form takes seconds."""
    import time
    from elimrec_amd import SyntheticDataset, slab
    from elimrec_amd.model import create_adj_mat
    for (U, I, E, ipw) in ((36656, 76085, 720829, 8), (36656, 1217360, 16 * 720829, 8)):
        ds = SyntheticDataset(U, I, E, feat_dims=(4, 4, 4), seed=0)
        adj = create_adj_mat(*ds.get_train_interactions(), U, I, "pre").tocsr()
        adj.sort_indices()
        t0 = time.perf_counter()
        host = slab.SellPlan(adj, DEV, threshold=64, side_split=U, tiered=True, ipw=ipw)
        torch.cuda.synchronize()
        t_host = time.perf_counter() - t0
        rp, col, val = _t(adj.indptr.astype(np.int64)), _t(adj.indices.astype(np.int32)), _t(adj.data.astype(np.float32))
        slab.SellPlan.on_device(rp, col, val, adj.shape[1], threshold=64, side_split=U, ipw=ipw)      # (code objects, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        devp = slab.SellPlan.on_device(rp, col, val, adj.shape[1], threshold=64, side_split=U, ipw=ipw)
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t0
        _plans_equal(host, devp)
        print("plan of %d non-zeros: host %.2f s, device %.3f s" % (adj.nnz, t_host, t_dev))
        assert t_dev < 0.5


@pytest.mark.parametrize("name", ["ml3", "gcmc"])
def test_trainer_on_a_device_built_plan_matches_the_reference_fixture(name):
    """--plan_build=device: the engine's wave-tile plans (the transposed one too for the gcmc adjacency, which is not symmetric) come
    from the device CSR through csrc/plan.hip; the trainer reproduces the reference's golden losses and parameters, and the plan is
    the host-built one bit for bit."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, slab
    g = load_golden(name)
    model, _ = build_model_from_fixture(g, DEV, extra_argv=["--plan_build=device"])
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt)
    host = slab.SellPlan(model._scipy_adj(), DEV, threshold=eng.plan.threshold, side_split=model.num_users, tiered=True, ipw=eng.plan.tile_groups)
    _plans_equal(host, eng.plan)
    assert (eng.planT is eng.plan) == bool(model._adj_symmetric)
    steps = int(g["steps"])
    for t in range(1, steps + 1):
        loss = tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))
        assert abs(float(loss) - float(g["step%d/loss" % t])) < 1e-5, t
    eng.sync_to_model()
    sd = model.state_dict()
    for k, v in sub(g, "after%d" % steps).items():
        assert np.abs(sd[k].cpu().numpy() - v).max() < 2e-5, k


@pytest.mark.parametrize("mode", ["train", "frozen"])
def test_tiktok_word_bag_fixture_on_the_engine(mode):
    """The data set "tiktok" (models/EliMRec.py:371-378; fixture `tiktok`, captured from the reference with a scatter-mean stub): the
    model builds t_feat from the loaded word_embedding.weight and the items' word lists exactly as the reference does (1e-7, not
    normalised), the trainer reproduces the reference's three losses and every parameter after Adam -- INCLUDING
    word_embedding.weight, which the reference keeps updating through its retained graph (main.py:100) although nothing reads it
    again: the engine computes that gradient too (default; first-step gradient 1e-4 row-wise, the rows of the words in use after
    three Adam steps 2e-5). `--word_embedding=frozen` leaves it a frozen
    checkpoint key -- the same losses, parameters and scores otherwise."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, EliMRec, FusedAdam
    from helpers import FixtureDataset, fixture_argv
    g = load_golden("tiktok")
    model = EliMRec(make_config(fixture_argv(g) + ["--word_embedding=%s" % mode]), FixtureDataset(g))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sub(g, "init").items() if "@" not in k}, strict=True)
    assert (model.t_feat.numpy() - g["t_feat"]).__abs__().max() < 1e-7            # built from the LOADED word embeddings
    with torch.no_grad():
        for m in ("v", "a"):
            getattr(model, m + "_feat").copy_(torch.from_numpy(g[m + "_feat"]))
    model = model.to(DEV)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt)
    assert eng.word_train == (mode == "train")
    steps = int(g["steps"])
    rows = g["word_rows"]
    for t in range(1, steps + 1):
        loss = tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))
        assert abs(float(loss) - float(g["step%d/loss" % t])) < 1e-5, t
        if t == 1 and mode == "train":
            assert_grad_close(eng._grads["word_embedding.weight"].cpu().numpy()[rows], g["grad1/word_embedding.weight@rows"], "word_embedding.weight")
    sd = model.state_dict()
    for k, v in sub(g, "after%d" % steps).items():
        if "@" in k:
            continue
        if k == "word_embedding.weight":
            mine = sd[k].cpu().numpy()
            if mode == "frozen":
                assert np.array_equal(mine, g["init/" + k])
                assert np.abs(v[rows] - g["init/" + k][rows]).max() < 1.05 * steps * float(g["lr"])   # the deviation's size
            else:
                assert np.abs(mine[rows] - v[rows]).max() < 2e-5
                # (the fixture keeps the rows of the words in use; the others are loaded as zeros here and stay zero -- weight
                # decay of zero -- where the reference's own rows decay by <= lr per step: `@unused_max_move`)
                unused = np.setdiff1d(np.arange(mine.shape[0]), rows)
                assert np.abs(mine[unused]).max() == 0.0
            continue
        assert np.abs(sd[k].cpu().numpy() - v).max() < 2e-5, k
    # predict() on the tables of the last forward, as for the other fixtures
    model.predict_type, model.fusion_mode = "TIE", "rubi"
    users = g["eval_users"].tolist()
    assert np.abs(model.predict(users).numpy() - g["predict/rubi/TIE"]).max() < 1e-5


@pytest.mark.parametrize("fused", ["1", "0"])
def test_trainer_with_the_window_sweep_matches_the_reference_fixtures(monkeypatch, fused):
    """ELIMREC_SWEEP=1 forces the large-table form of the hops on the small fixtures: whole hops = tile hop over the item rows +
    window sweep over the user rows. fused = 1 (the default): the last hop's two launches have the Adam step as their epilogue
    (elimrec_slab_sweep_hop_adam for the user rows; the optimizer spans of the projections ride with the item rows); fused = 0: Adam
    in a launch of its own. (The weight gradients' reduce is a launch of its own in this form either way.) Three trainer steps reproduce the reference's golden losses and parameters, as the default form does."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    monkeypatch.setenv("ELIMREC_SWEEP", "1")
    monkeypatch.setenv("ELIMREC_SWEEP_WINDOW", "64")
    if fused == "0":
        monkeypatch.setenv("ELIMREC_FUSE_ADAM", "0")
    for name in ("ml3", "kwai", "gcmc"):
        g = load_golden(name)
        model, cfg = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        eng = ColumnShardEngine(model)
        tr = ColumnShardTrainer(eng, opt)
        assert eng.sweep and eng.plan.sweep is not None and eng._fuse_adam() == (fused == "1") and not eng._fuse_reduce()
        steps = int(g["steps"])
        for t in range(1, steps + 1):
            loss = tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))
            assert abs(float(loss) - float(g["step%d/loss" % t])) < 1e-5, (name, t)
        eng.sync_to_model()
        sd = model.state_dict()
        for k, v in sub(g, "after%d" % steps).items():
            assert np.abs(sd[k].cpu().numpy() - v).max() < 2e-5, (name, k)


def test_split_rows_of_the_batch_only_changes_no_bit(monkeypatch):
    """ELIMREC_LONG_WANTED=1 forces what the swept form does by default -- hop L's split rows evaluated for the batch's rows only,
    behind the join with the planner -- on a synthetic graph with rows of hundreds of neighbours: losses, parameters and Adam
    moments after ten steps (the native program engaged) are bitwise those of the default order (every split row, ahead of the join)."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, EliMRec, FusedAdam, SyntheticDataset, set_seed
    cfg = make_config(["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0"])
    ds = SyntheticDataset(300, 2500, 40000, feat_dims=(16, 8, 12), seed=3)
    gen = torch.Generator().manual_seed(5)
    B = 256
    batches = [(torch.randint(0, 300, (B,), generator=gen).to(DEV), torch.randint(0, 2500, (B,), generator=gen).to(DEV),
                torch.randint(0, 2500, (B,), generator=gen).to(DEV)) for _ in range(10)]
    got = {}
    for wanted in ("0", "1"):
        monkeypatch.setenv("ELIMREC_LONG_WANTED", wanted)
        set_seed(11)
        model = EliMRec(cfg, ds).to(DEV)
        opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        eng = ColumnShardEngine(model)
        tr = ColumnShardTrainer(eng, opt)
        assert eng._long_wanted_only() == (wanted == "1") and eng.plan.n_long > 100       # (every user row: ~130 neighbours each)
        losses = [float(tr.step(*b)) for b in batches]
        assert tr._native_state()["native_steps"] > 0 and tr._native_state()["failed"] is None
        eng.sync_to_model()
        st = eng.optimizer_state()
        got[wanted] = (losses, {k: v.clone() for k, v in model.state_dict().items()}, st["exp_avg"].clone(), st["exp_avg_sq"].clone())
    a, b = got["0"], got["1"]
    assert a[0] == b[0]
    assert all(torch.equal(a[1][k], b[1][k]) for k in a[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


@pytest.mark.parametrize("d,w,L", [(64, 32, 3), (16, 16, 2), (8, 8, 4), (64, 8, 1), (32, 32, 3)])
def test_slab_rows_layer_means_and_inline_last_hop(d, w, L):
    """elimrec_slab_rows: the layer means at listed rows with hop L evaluated inline (split rows from the seg_only
    launch) are bitwise what the full tables give; several row lists with device counts (the multi-rank form)."""
    from elimrec_amd import slab
    n, U = 2500, 900
    m = _random_graph(n, 5 * d + L)
    plan = slab.SellPlan(m, DEV, threshold=32, side_split=U)
    ns = d // w
    torch.manual_seed(L)
    tabs = [slab.SlabTable(n, ns, w, DEV).from_rows(torch.randn(n, d, device=DEV))]
    for k in range(L):
        tabs.append(tabs[0].like())
        slab.hop(plan, tabs[k], tabs[k + 1])
    D = [t.dense() for t in tabs]
    s = D[0] + D[1]
    for k in range(2, L + 1):
        s = s + D[k]
    inv = 1.0 / (L + 1)
    nar_u = D[0][:U].clone()
    nar_i = D[1][U:].clone()
    for k in range(2, L + 1):
        if k % 2 == 0:
            nar_u = nar_u + D[k][:U]
        else:
            nar_i = nar_i + D[k][U:]
    mean, nar = s * inv, torch.cat([nar_u, nar_i]) * inv
    full0, fulln = torch.empty(n, d, device=DEV), torch.empty(n, d, device=DEV)
    slab.rows(plan, ns, w, L, U, [t.data for t in tabs], None, None, None, n, 1, full0, fulln, False)
    assert torch.equal(full0, mean) and torch.equal(fulln, nar)
    # two row lists, last hop inline
    R = 640
    lists, counts = [], []
    for seed, cnt in ((1, 500), (2, 77)):
        g = torch.Generator(device="cpu").manual_seed(seed)
        ids = torch.sort(torch.randperm(n, generator=g)[:cnt])[0].int()
        lists.append(torch.cat([ids, torch.full((R - cnt,), -(1 << 30), dtype=torch.int32)]))
        counts.append(cnt)
    rows = torch.stack(lists).to(DEV)
    cnts = torch.tensor(counts, dtype=torch.int32, device=DEV)
    long_tab = torch.empty(ns * max(plan.n_long, 1) * w, device=DEV)
    slab.hop(plan, tabs[L - 1], long_tab, seg_only=True)
    packed = torch.full((2 * R, 2 * d), float("nan"), device=DEV)
    slab.rows(plan, ns, w, L, U, [t.data for t in tabs[:L]] + [None], long_tab, rows, cnts, R, 2, packed[:, :d], packed[:, d:], False)
    for li, cnt in enumerate(counts):
        r = rows[li, :cnt].long()
        assert torch.equal(packed[li * R:li * R + cnt, :d], mean[r])
        assert torch.equal(packed[li * R:li * R + cnt, d:], nar[r])
        assert torch.isnan(packed[li * R + cnt:(li + 1) * R]).all()          # padded slots are not written
    # the split rows of the WANTED rows only (seg_only with a row bitmap -- the batch's rows -- on the wave-tile plan the engine uses):
    # the listed rows' compact rows hold the bits of the launch over every split row, those nobody listed keep what the buffer held
    gs = slab.choose_groups(ns)
    plan_t = slab.SellPlan(m, DEV, threshold=32, side_split=U, tiered=True, ipw=64 // ((ns // gs) * (w // 4)))
    if plan_t.n_long:
        lr = plan_t.t["long_rows"][:plan_t.n_long]
        listed = torch.zeros(n, dtype=torch.bool, device=DEV)
        for li, cnt in enumerate(counts):
            listed[rows[li, :cnt].long()] = True
        listed[lr[::2].long()] = True                                            # (... and every second split row, listed or not)
        wanted = torch.zeros((n + 31) // 32, dtype=torch.int32, device=DEV)
        slab.rows_bitmap(torch.nonzero(listed).flatten().int().view(1, -1).contiguous(), n, wanted)
        long_all = torch.full((ns * plan_t.n_long * w,), 7.0, device=DEV)
        long_w = torch.full_like(long_all, 7.0)
        slab.hop(plan_t, tabs[L - 1], long_all, gs=gs, seg_only=True)
        slab.hop(plan_t, tabs[L - 1], long_w, gs=gs, seg_only=True, add_mask=wanted)
        la, lw = long_all.view(ns, plan_t.n_long, w), long_w.view(ns, plan_t.n_long, w)
        is_listed = listed[plan_t.t["long_rows"].long()[:plan_t.n_long]]
        assert bool(is_listed.any()) and (plan_t.n_long < 2 or bool((~is_listed).any())) and not bool((la == 7.0).all(-1).all(0).any())
        assert torch.equal(lw[:, is_listed], la[:, is_listed])
        assert (lw[:, ~is_listed] == 7.0).all()


@pytest.mark.parametrize("W,R,U,I,d,w", [(5, 300, 700, 1300, 32, 32), (2, 64, 40, 90, 64, 8), (8, 1000, 3000, 5000, 8, 8),
                                         (1, 50, 10, 200, 64, 32)])
def test_slab_merge_rows_vs_torch(W, R, U, I, d, w):
    from elimrec_amd import slab
    N = U + I
    torch.manual_seed(W * R)
    keys, rows = [], torch.randn(W * R, 2 * d, device=DEV)
    H, G = torch.zeros(N, d, device=DEV, dtype=torch.float64), torch.zeros(N, d, device=DEV, dtype=torch.float64)
    for r in range(W):
        cnt = int(torch.randint(R // 2, R, (1,)))
        ids = torch.sort(torch.randperm(N)[:cnt])[0].int().to(DEV)
        keys.append(torch.cat([ids, torch.full((R - cnt,), -(1 << 30), dtype=torch.int32, device=DEV)]))
        H.index_add_(0, ids.long(), rows[r * R:r * R + cnt, :d].double())
        G.index_add_(0, ids.long(), rows[r * R:r * R + cnt, d:].double())
    keys = torch.cat(keys)
    sa, sb = slab.SlabTable(N, d // w, w, DEV), slab.SlabTable(N, d // w, w, DEV)
    sa.data.zero_(); sb.data.zero_()
    mask = torch.full(((N + 31) // 32 + 2,), -1, dtype=torch.int32, device=DEV)
    slab.merge_rows(rows, keys, W, U, I, sa, sb, mask)
    assert (sa.dense().double() - torch.cat([H[:U], G[U:]])).abs().max().item() < 1e-5
    assert (sb.dense().double() - torch.cat([G[:U], H[U:]])).abs().max().item() < 1e-5
    active = torch.zeros(N, dtype=torch.bool, device=DEV)
    active[keys[keys >= 0].long()] = True
    assert torch.equal(mask[:(N + 31) // 32], _bitmap(active)[:(N + 31) // 32])
    again_a = sa.dense().clone()
    slab.merge_rows(rows, keys, W, U, I, sa, sb, mask)
    assert torch.equal(sa.dense(), again_a)                                    # fixed summation order


def test_adam_step_out_equals_in_place_kernel():
    from elimrec_amd import ops, slab
    torch.manual_seed(0)
    n = 100003
    p, g, m, v = (torch.randn(n, device=DEV) for _ in range(4))
    v.abs_()
    p2, m2, v2, out = p.clone(), m.clone(), v.clone(), torch.empty_like(p)
    ops.adam_step(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-4, 3)
    slab.adam_step_out(p2, out, g, m2, v2, 1e-3, 0.9, 0.999, 1e-8, 1e-4, 3)
    assert torch.equal(out, p) and torch.equal(m2, m) and torch.equal(v2, v)


# ----------------------------------------------------------------------------- the trainer, world 1, golden fixtures
@pytest.mark.parametrize("name", ["ml3", "kwai", "gcmc", "normal"])
def test_column_shard_trainer_matches_reference_fixture(name):
    """ColumnShardTrainer at world 1 on the reference's golden vectors: losses 1e-5, parameters after Adam 2e-5,
    predict() after training (tables of the last forward, materialised from the slab-major layer tables) 1e-5.
    The fifth fixture, `ablate`, is adj_type=norm: D^-1 (A + I) has a diagonal, so a hop mixes the user-borne and the
    item-borne part of every table on both sides and the parity split that lets ONE [N x d] table carry the id table and
    the part all feature tables share does not exist -- that adjacency runs on the replicated row-major trainer
    (elimrec_amd/dist.py), against the same fixture in tests/test_hip_parity.py::test_training_steps_match_reference."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden(name)
    model, _ = build_model_from_fixture(g, DEV)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt)
    steps = int(g["steps"])
    for t in range(1, steps + 1):
        loss = tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))
        assert abs(float(loss) - float(g["step%d/loss" % t])) < 1e-5, t
        if t in (1, steps):
            eng.sync_to_model()
            sd = model.state_dict()
            for k, v in sub(g, "after%d" % t).items():
                assert np.abs(sd[k].cpu().numpy() - v).max() < 2e-5, (t, k)
    assert rel_err(model.all_users.cpu(), g["cache/all_users"]) < 1e-4
    model.fusion_mode, model.predict_type = "rubi", "TIE"
    got = model.predict(g["eval_users"].tolist()).numpy()
    assert np.abs(got - g["predict/rubi/TIE"]).max() < 1e-5


def test_column_shard_trainer_equals_row_major_trainer():
    """Same model, same batches: the slab-major engine (folded tables, batch rows) and the row-major trainer on the UNFOLDED
    form (dist.py; --propagation=bipartite --head_rows=all: every table through the graph, every projection over all rows --
    the reference's own amount of work) agree to the suite's tolerances, and the slab engine is bitwise reproducible."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    from elimrec_amd.dist import DataParallelTrainer
    g = load_golden("gcmc")
    runs = []
    for kind in ("rows", "slab", "slab"):
        model, _ = build_model_from_fixture(g, DEV, extra_argv=["--propagation=bipartite", "--head_rows=all"] if kind == "rows" else [])
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        if kind == "rows":
            assert not model._lazy and not model._folded
            tr = DataParallelTrainer(model, opt)
        else:
            eng = ColumnShardEngine(model)
            tr = ColumnShardTrainer(eng, opt)
        losses = [float(tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))) for t in (1, 2, 3)]
        runs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    assert runs[1][0] == runs[2][0]
    for k in runs[1][1]:
        assert torch.equal(runs[1][1][k], runs[2][1][k]), k
        assert (runs[0][1][k] - runs[1][1][k]).abs().max().item() < 2e-5, k
    assert np.allclose(runs[0][0], runs[1][0], atol=1e-5)


# ----------------------------------------------------------------------------- W ranks emulated on one GPU
def _emulated_step(engines, batches, deferred=False):
    """What ColumnShardTrainer.step does on W ranks, with the collectives done by hand in one process. deferred: the trainer's
    default multi-rank order -- the weight gradients are finished behind the adjoint hops' tiles, all-reduced after them, and
    the projection weights' optimizer spans run in cs_update."""
    W = len(engines)
    class _NoWait(object):
        def wait(self):
            return True
    for e in engines:
        e.defer_wgrads = bool(deferred)
    acts = torch.stack([e.cs_plan(*b).clone() for e, b in zip(engines, batches)])               # all_gather
    sends = [e.cs_forward(acts) for e in engines]
    sends = [None if s is None else s.clone() for s in sends]
    if W > 1 and engines[0].lookup:      # row-sharded constants: owners pack, the all_to_all by hand, requesters unpack
        rb = engines[0].lookup_row_bytes
        counts = engines[0].cs_lookup_counts(acts).cpu().numpy()            # [requester][owner]
        packed = [e.cs_lookup_pack(acts).clone() for e in engines]
        for q, e in enumerate(engines):
            chunks = []
            for o in range(W):
                off = int(counts[:q, o].sum()) * rb
                chunks.append(packed[o][off:off + int(counts[q, o]) * rb])
            e.cs_lookup_unpack(torch.cat(chunks))
    scale = torch.full((1,), 1.0 / W, device=DEV)
    losses, sends2, wgs = [], [], []
    for q, e in enumerate(engines):
        recv = None if W == 1 else torch.stack([sends[p][q] for p in range(W)])                 # all_to_all
        losses.append(e.cs_head(recv).clone())
        s2, wg = e.cs_backward_local(scale)
        sends2.append(s2.clone()); wgs.append(wg)
    if deferred and W > 1:
        assert all(e.wgrads_deferred() for e in engines)
        for q, e in enumerate(engines):
            e.cs_backward_hops(torch.stack([sends2[p][q] for p in range(W)]), acts, None, lambda: _NoWait())      # all_to_all; hops finish wg
        total = torch.stack([w.clone() for w in wgs]).sum(0)                                    # the late all_reduce
        for q, e in enumerate(engines):
            wgs[q].copy_(total)
            e.cs_update()
        return torch.stack(losses).mean()
    total = torch.stack([w.clone() for w in wgs]).sum(0)                                        # all_reduce
    for q, e in enumerate(engines):
        wgs[q].copy_(total)
        e.cs_backward_hops(torch.stack([sends2[p][q] for p in range(W)]), acts)                 # all_to_all
        e.cs_update()
    return torch.stack(losses).mean()


@pytest.mark.parametrize("deferred", [False, True], ids=["early-allreduce", "deferred-weight-gradients"])
@pytest.mark.parametrize("W", [2, 4, 8])
def test_column_shard_ranks_emulated_on_one_gpu(W, deferred):
    """W column-shard ranks (each owns recdim/W columns and 1/W of the triplets) equal ONE rank on the whole batch:
    loss, embeddings and projection weights after two steps; every rank holds the same projection weights. Both orders of
    the multi-rank backward: weight gradients in a launch of their own, all-reduced under the adjoint hops; and the trainer's
    default -- both phases behind the adjoint hops' tiles (4- and 2-lane row pieces at W = 4 / 8), the all-reduce after them."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden("ml3")
    B = (len(g["step1/users"]) // W) * W

    def make(world, rank):
        model, _ = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        eng = ColumnShardEngine(model)
        eng.cs_setup(world, rank, opt)
        return model, eng

    one_model, one = make(1, 0)
    ranks = [make(W, q) for q in range(W)]
    for t in (1, 2):
        u, p, n = (_t(g["step%d/%s" % (t, k)])[:B] for k in ("users", "pos", "neg"))
        l1 = _emulated_step([one], [(u, p, n)])
        h = B // W
        lw = _emulated_step([e for _, e in ranks], [(u[q * h:(q + 1) * h], p[q * h:(q + 1) * h], n[q * h:(q + 1) * h]) for q in range(W)],
                            deferred=deferred)
        assert abs(float(l1) - float(lw)) < 1e-6
    full = one.master[one.cur].dense()
    shards = torch.cat([e.master[e.cur].dense() for _, e in ranks], dim=1)
    assert (full - shards).abs().max().item() < 2e-5        # Adam's lr*g/(|g|+eps) amplifies round-off where |g| ~ eps
    sd1 = one_model.state_dict()
    for q, (m, _) in enumerate(ranks):
        for k, v in m.state_dict().items():
            if not k.startswith(("embedding_user.", "embedding_item.")):
                assert (v - sd1[k]).abs().max().item() < 2e-5, (q, k)
                assert torch.equal(v, ranks[0][0].state_dict()[k]), (q, k)


# ----------------------------------------------------------------------------- BASELINE.json configs[3] (C4)
def _shape_step_vs_oracle(U, I, E, dims, recdim, B, world=1):
    from elimrec_amd import ColumnShardEngine, EliMRec, FusedAdam, SyntheticDataset, set_seed
    from oracle import elimrec_oracle as eo
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = make_config(["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=%d" % recdim, "--verbose=0"])
    ds = SyntheticDataset(U, I, E, feat_dims=dims, seed=1)
    set_seed(7)
    model = EliMRec(cfg, ds)
    init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    gen = torch.Generator().manual_seed(3)
    train = ds.train_matrix.tocoo()
    pick = torch.randint(0, train.nnz, (B,), generator=gen).numpy()
    u = torch.from_numpy(train.row[pick].astype(np.int64))
    p = torch.from_numpy(train.col[pick].astype(np.int64))
    n = torch.randint(0, I, (B,), generator=gen)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    engines = []
    for q in range(world):
        eng = ColumnShardEngine(model)
        eng.cs_setup(world, q, opt)
        engines.append(eng)
    # forward + backward of one rank-0 step without the update: gradients against the oracle
    eng = engines[0]
    eng.keep_grad = True          # the fused last-hop + Adam epilogue also stores the gradient table this test reads
    acts = eng.cs_plan(u.to(DEV), p.to(DEV), n.to(DEV)).view(1, -1)
    assert world == 1
    eng.cs_forward(acts)
    loss = eng.cs_head(None)
    s2, _ = eng.cs_backward_local(torch.ones(1, device=DEV))
    eng.cs_backward_hops(s2, acts)
    tu, ti = ds.get_train_interactions()
    adj = eo.build_adj(tu, ti, U, I, cfg["adj_type"])
    feats = {m: eo.OracleEliMRec.normalize_features(getattr(ds, m + "_feat")) for m in ("v", "a", "t")}
    om = eo.OracleEliMRec(U, I, recdim, cfg["layer_num"], adj, feats, init, cfg["alpha"], dataset_name="synthetic")
    ol = om.bpr_loss(u, p, n)
    ol.backward()
    assert abs(float(loss) - float(ol.detach())) < 1e-5
    want = om.grads()
    gE = eng.grad.dense().cpu()
    assert_grad_close(gE[:U], want["embedding_user.weight"], "embedding_user.weight")
    assert_grad_close(gE[U:], want["embedding_item.weight"], "embedding_item.weight")
    for k, v in eng._grads.items():
        assert_grad_close(v.cpu(), want[k], k)
    eng._test_batch, eng._test_oracle, eng._test_init = (u, p, n), om, init
    eng._test_oracle_loss = float(ol.detach())
    return model, eng


def _predict_and_topk_vs_oracle(model, om, users, K=10):
    """predict() of a block of users on the tables the last forward cached, and the device top-K, against the oracle's
    predict(): scores 1e-5; the top-K lists ARE the stable (score desc, id asc) ranking of the device scores and differ
    from the oracle's ranking only where two oracle scores are closer than 2e-6."""
    for ptype in ("TIE", "TE"):
        model.predict_type = om.predict_type = ptype
        got = model.predict(users).numpy()
        want = om.predict(users).numpy()
        assert np.abs(got - want).max() < 1e-5, ptype
        idx, val = model.predict_device(users, top_k=K)
        idx = idx.cpu().numpy()
        order = np.argsort(-got, axis=1, kind="stable")[:, :K]
        assert np.array_equal(idx, order), ptype
        ref_order = np.argsort(-want, axis=1, kind="stable")[:, :K]
        for r in np.nonzero((order != ref_order).any(1))[0]:
            assert np.abs(want[r][order[r]] - want[r][ref_order[r]]).max() < 2e-6, (ptype, r)


def test_small_shape_step_vs_oracle_through_the_slab_engine():
    _shape_step_vs_oracle(700, 1900, 9000, (24, 8, 12), 64, 257)


def test_c4_shape_step_vs_oracle():
    """BASELINE.json configs[3] on one GPU: Tiktok shape x16 items (|I| = 1 217 360), recdim 128, B = 2048 -- one
    step of the slab-major engine (the table every rank of an 8-GPU job holds a 16-column slice of) vs the oracle."""
    model, eng = _shape_step_vs_oracle(36656, 1217360, 16 * 720829, (128, 128, 128), 128, 2048)
    # ... and predict() / top-10 of 64 users over the 1.2 M-item catalogue at recdim 128 (the 16-user-per-wave scorer, chunked
    # selection) from the tables that forward cached, against the oracle
    _predict_and_topk_vs_oracle(model, eng._test_oracle, list(range(0, 36656, 570))[:64])


def test_sweep_gate_keeps_the_tile_hop_when_the_slabs_do_not_tile_the_xcds(monkeypatch):
    """recdim 96 = 3 slabs of 32 floats: the window sweep hands every XCD one slab role and refuses 3 (or 6) slabs, so the
    engine must not take it even when it is forced on (ELIMREC_SWEEP=1) or the slice outgrows the caches -- one step vs the
    oracle on the tile hop; capacity.plan counts no sweep plan for such a slice either."""
    from elimrec_amd import capacity, slab
    monkeypatch.setenv("ELIMREC_SWEEP", "1")
    model, eng = _shape_step_vs_oracle(500, 1300, 9000, (16, 8, 12), 96, 300)
    assert (eng.ns, eng.w) == (3, 32) and not eng.sweep and eng.plan.sweep is None
    assert not slab.sweep_tiles_xcds(3) and not slab.sweep_tiles_xcds(6) and slab.sweep_tiles_xcds(4) and slab.sweep_tiles_xcds(16)
    with_sweep = capacity.plan(500, 1300, 9000, 64, (16, 8, 12), world=1, batch=300)
    without = capacity.plan(500, 1300, 9000, 96, (16, 8, 12), world=1, batch=300)
    assert any("window-sweep" in k for k in with_sweep["components"]) and not any("window-sweep" in k for k in without["components"])


def test_c5_shape_scaled_step_and_eval_vs_oracle():
    """BASELINE.json configs[4] scaled to one GPU and an oracle that finishes: |I| = 2 000 000, |U| = 20 000, 16 M
    interactions (users of degree ~800, items of degree ~8: the C5 ratio), recdim 256, three 256-d feature tables.
      * fp32: one step of the slab engine vs the oracle (loss 1e-5, every gradient 1e-4 max-norm and row by row);
        predict() / top-10 for 64 users over the 2 M-item catalogue vs the oracle;
      * --feature_dtype=f16 (the storage configs[4] names): the same step with the folded constants stored in fp16 and
        widened at the lookup -- stated tolerance of the mode against the fp32 run: loss 2e-3 abs, gradients 2e-2 max-norm
        (the exact statement of what it computes is test_16bit_feature_storage_equals_fp32_engine_on_rounded_constants);
    (Memory: tests/test_capacity_gpu.py checks capacity.plan() against the device allocator on the lean form of this shape.)"""
    from elimrec_amd import ColumnShardEngine, FusedAdam
    U, I, E, dims, d, B = 20000, 2000000, 16000000, (256, 256, 256), 256, 2048
    model, eng = _shape_step_vs_oracle(U, I, E, dims, d, B)
    om = eng._test_oracle
    print("device memory held after set-up + one step: %.2f GiB" % (torch.cuda.memory_allocated() / 2 ** 30))
    _predict_and_topk_vs_oracle(model, om, list(range(0, U, 311))[:64])
    # fp16 storage of the constants on the same model and batch
    u, p, n = eng._test_batch
    g32 = {k: v.clone() for k, v in eng._grads.items()}
    gE32 = eng.grad.dense().clone()
    # (the first engine's last adjoint hop carried the projection weights' Adam spans: back to the initial parameters)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in eng._test_init.items()})
    opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    e16 = ColumnShardEngine(model, feature_dtype="f16")
    e16.cs_setup(1, 0, opt)
    e16.keep_grad = True
    acts = e16.cs_plan(u.to(DEV), p.to(DEV), n.to(DEV)).view(1, -1)
    e16.cs_forward(acts)
    loss16 = float(e16.cs_head(None))
    s2, _ = e16.cs_backward_local(torch.ones(1, device=DEV))
    e16.cs_backward_hops(s2, acts)
    loss32 = eng._test_oracle_loss                  # (the oracle's loss of the same batch: _shape_step_vs_oracle)
    assert 0 < abs(loss16 - loss32) < 2e-3, (loss16, loss32)
    # the mode's stated tolerance (2e-2) in max-norm AND row by row, as the fp32 path is held to 1e-4 (helpers.assert_grad_close)
    assert_grad_close(e16.grad.dense().cpu(), gE32.cpu(), "embedding gradient, fp16 constants", rel=2e-2)
    for k, v in e16._grads.items():
        assert_grad_close(v.cpu(), g32[k].cpu(), k, rel=2e-2)
    assert e16.fshard.table.dtype == torch.float16 and e16.fshard.nbytes() < 0.51 * (U + I) * (sum(dims) + 4) * 4


def test_fused_head_forward_equals_batched_gemms(monkeypatch):
    """csrc/head.hip (feature blocks + fused Linear + single-modal heads of the active rows in one launch) against the
    two batched-GEMM launches it replaces, on a recdim-64 model: OutAct, YAct, loss and the gradient rows to fp32
    round-off; the whole step then matches the oracle like the unfused one."""
    from elimrec_amd import ColumnShardEngine, EliMRec, FusedAdam, SyntheticDataset, set_seed
    cfg = make_config(["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0"])
    ds = SyntheticDataset(700, 1900, 9000, feat_dims=(128, 24, 64), seed=1)
    got = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("ELIMREC_FUSED_HEAD", fused)
        set_seed(7)
        model = EliMRec(cfg, ds).to(DEV)
        eng = ColumnShardEngine(model)
        eng.cs_setup(1, 0, FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"]))
        assert eng._fused_head_ok() == (fused == "1")
        gen = torch.Generator().manual_seed(3)
        B = 300
        u = torch.randint(0, 700, (B,), generator=gen).to(DEV)
        p = torch.randint(0, 1900, (B,), generator=gen).to(DEV)
        n = torch.randint(0, 1900, (B,), generator=gen).to(DEV)
        acts = eng.cs_plan(u, p, n).view(1, -1)
        eng.cs_forward(acts)
        loss = eng.cs_head(None)
        ws = model._ws
        na = int(ws["seg_info"][0])
        got[fused] = (float(loss), ws["OutAct"][:na].clone(), ws["YAct"][:na].clone(), ws["grad_rows"].clone())
    a, b = got["0"], got["1"]
    assert abs(a[0] - b[0]) < 1e-6
    for x, y in zip(a[1:], b[1:]):
        assert rel_err(y.cpu(), x.cpu()) < 2e-6


@pytest.mark.parametrize("d,w,gs,ipw", [(64, 32, 2, 8), (8, 8, 1, 32), (64, 64, 1, 4),
                                        (128, 32, 4, 8), (16, 16, 1, 16)])
def test_tiered_one_launch_hop(d, w, gs, ipw):
    """The default hop: rows above the lane-group threshold go to one wave or one workgroup each, only the longest are
    segmented and combined in-launch. Against an fp64 product (1e-5), plain / masked +
    addend / seg_only (every row above the threshold, compact), and bitwise reproducible."""
    from elimrec_amd import slab
    n = 4000
    m = _random_graph(n, 7 * d + w, hot=9, hot_deg=900)
    rng = np.random.RandomState(5)
    extra = sp.csr_matrix((np.ones(3000, np.float32) * 0.01, (np.zeros(3000, int) + 17, rng.choice(n, 3000, replace=False))), shape=(n, n))
    mid_rows = np.repeat(np.arange(40, 70), 120)                                  # 30 rows of ~120 neighbours: a wave each
    mid = sp.csr_matrix((np.full(len(mid_rows), 0.02, np.float32), (mid_rows, rng.randint(0, n, len(mid_rows)))), shape=(n, n))
    m = (m + extra + mid).tocsr()                # row 17: ~3000 neighbours -> segments + tickets at every ipw
    m.sum_duplicates()
    m.sort_indices()
    plan = slab.SellPlan(m, DEV, threshold=32, side_split=1500, tiered=True, ipw=ipw)
    assert plan.tiered and plan.n_w1 + plan.n_w4 > 0 and (plan.n_seg > 0 or ipw >= 16)
    ns = d // w
    torch.manual_seed(1)
    X = torch.randn(n, d, device=DEV)
    x = slab.SlabTable(n, ns, w, DEV).from_rows(X)
    Xr = x.dense().double()
    A64 = torch.from_numpy(m.astype(np.float64).toarray()).to(DEV)
    want = A64 @ Xr
    # fp32 sums of up to 3000 terms: the error bound scales with sum |a||x| of the row (4 ulp-ish of it), not with 1
    scale = A64.abs() @ Xr.abs()
    tol = lambda ref, sc=scale: 4e-6 * sc + 1e-6
    y = x.like()
    y.data.fill_(float("nan"))
    slab.hop(plan, x, y, gs=gs)
    assert ((y.dense().double() - want).abs() <= tol(want)).all()
    y2 = x.like()
    slab.hop(plan, x, y2, gs=gs)
    assert torch.equal(y.data, y2.data)
    act = torch.rand(n, device=DEV) < 0.1
    bm = _bitmap(act)
    S = torch.randn(n, d, device=DEV)
    src = slab.SlabTable(n, ns, w, DEV).from_rows(S)
    slab.hop(plan, src, y, gs=gs, src_mask=bm, add=src, add_mask=bm, scale=0.5)
    Sm = S.double() * act[:, None]
    want2 = (A64 @ Sm + Sm) * 0.5
    assert ((y.dense().double() - want2).abs() <= tol(want2, A64.abs() @ Sm.abs() + Sm.abs())).all()
    long_tab = torch.full((ns * plan.n_long * w,), float("nan"), device=DEV)
    slab.hop(plan, x, long_tab, gs=gs, seg_only=True)
    lt = long_tab.view(ns, plan.n_long, w).permute(1, 0, 2).reshape(plan.n_long, d).double()
    rows = plan.t["long_rows"][:plan.n_long].long()
    assert ((lt - want[rows]).abs() <= 4e-6 * scale[rows] + 1e-6).all()


def test_loss_tensors_of_an_epoch_stay_valid():
    """main.py stacks the loss tensors of a whole epoch before one device->host copy: the tensors the trainer returned
    hundreds of steps ago must still hold their step's loss (they are slots of a ring, shard.LOSS_RING long)."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden("ml3")
    batches = [tuple(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")) for t in (1, 2, 3)]
    runs = []
    for keep in (True, False):
        model, _ = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
        assert tr.engine.loss_ring_len >= 4096
        out = [tr.step(*batches[i % 3]) for i in range(300)] if keep else [float(tr.step(*batches[i % 3])) for i in range(300)]
        runs.append(torch.stack(out).cpu().tolist() if keep else out)
    assert runs[0] == runs[1]


@pytest.mark.parametrize("masked", [True, False])
def test_weight_gradients_behind_a_hops_tiles(masked):
    """elimrec_slab_hop_bwd_w: the partial launch (phase 0) and the slab reduce (phase 1) of a weight-gradient batch as
    extra workgroups of two hop launches, and the merge / deferred-reduce forms of the batched call -- bitwise the
    gradients of elimrec_linear_bwd_w_batched, bitwise the tables of elimrec_slab_hop; gradients 1e-4 rel vs fp64."""
    from elimrec_amd import ops, slab
    n, d, w, gs = 3000, 64, 32, 2
    m = _random_graph(n, 11, hot=5, hot_deg=700)
    plan = slab.SellPlan(m, DEV, threshold=64, side_split=1000, tiered=True, ipw=8)
    torch.manual_seed(3)
    x = slab.SlabTable(n, d // w, w, DEV).from_rows(torch.randn(n, d, device=DEV))
    bm = _bitmap(torch.rand(n, device=DEV) < 0.2) if masked else None
    R = 1500
    A1, B1 = torch.randn(R, 64, device=DEV), torch.randn(R, 256, device=DEV)
    A2, B2 = torch.randn(R, 64, device=DEV), torch.randn(n, 128, device=DEV)
    idx = torch.randint(0, n, (R,), device=DEV, dtype=torch.int32)
    rng = torch.tensor([100, 1400], dtype=torch.int32, device=DEV)

    def problems(tag):
        o = {k: torch.full(s, float("nan"), device=DEV) for k, s in (("w1", (64, 256)), ("b1", (64,)), ("w2", (64, 128)), ("b2", (64,)))}
        return o, [dict(A=A1, B=B1, out=o["w1"], colsum=o["b1"]), dict(A=A2, B=B2, out=o["w2"], row_index=idx, rng=rng, colsum=o["b2"])]

    wsp = torch.empty(ops.linear_bwd_w_batched_workspace([(R, 64, 256), (R, 64, 128)]), dtype=torch.uint8, device=DEV)
    ref, pr = problems("ref")
    ops.linear_bwd_w_batched(pr, wsp)
    rel = lambda got, want: ((got.double() - want).norm() / want.norm()).item()
    assert rel(ref["w1"], A1.double().T @ B1.double()) < 1e-5
    sel = slice(100, 1400)
    assert rel(ref["w2"], A2[sel].double().T @ B2[idx[sel].long()].double()) < 1e-5
    assert rel(ref["b1"], A1.double().sum(0)) < 1e-5
    y0, y1 = x.like(), x.like()
    slab.hop(plan, x, y0, gs=gs, src_mask=bm)
    slab.hop(plan, y0, y1, gs=gs)
    # both phases behind hops
    got, pr = problems("tail")
    wsp2 = torch.empty_like(wsp)
    h = ops.linear_bwd_w_batched(pr, wsp2, defer_all=True)
    z0, z1 = x.like(), x.like()
    slab.hop(plan, x, z0, gs=gs, src_mask=bm, bwd_w=h, bwd_w_phase=0)
    slab.hop(plan, z0, z1, gs=gs, bwd_w=h, bwd_w_phase=1)
    assert torch.equal(z0.data, y0.data) and torch.equal(z1.data, y1.data)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
    # partial launch of its own, reduce behind a hop / as a launch
    for tail in (True, False):
        got, pr = problems("defer")
        h = ops.linear_bwd_w_batched(pr, wsp2, defer_reduce=True)
        if tail:
            slab.hop(plan, x, z0, gs=gs, src_mask=bm, bwd_w=h, bwd_w_phase=1)
            assert torch.equal(z0.data, y0.data)
        else:
            ops.linear_bwd_w_reduce(h)
        for k in ref:
            assert torch.equal(ref[k], got[k]), (k, tail)
    # the adjoint-source merge as extra workgroups of the partial launch
    U, I, M = 1000, 2000, 2
    keys = torch.sort(torch.randperm(n)[:700])[0].int().to(DEV)
    keys = torch.cat([keys, torch.full((68,), -(1 << 30), dtype=torch.int32, device=DEV)])
    rows = torch.randn(768, M * d, device=DEV)
    sa, sb, ta, tb = (slab.SlabTable(n, d // w, w, DEV) for _ in range(4))
    for t in (sa, sb, ta, tb):
        t.data.zero_()
    mk1 = torch.full(((n + 31) // 32 + 2,), -1, dtype=torch.int32, device=DEV)
    mk2 = mk1.clone()
    slab.merge_rows(rows, keys, 1, U, I, sa, sb, mk1, M=M)
    got, pr = problems("merge")
    ops.linear_bwd_w_batched(pr, wsp2, merge=dict(rows=rows, keys=keys, world=1, U=U, I=I, srcA=ta, srcB=tb, mask=mk2, M=M))
    assert torch.equal(sa.data, ta.data) and torch.equal(sb.data, tb.data) and torch.equal(mk1, mk2)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k


@pytest.mark.parametrize("name", ["ml3", "normal", "kwai"])
def test_step_variants_are_bitwise_equal(monkeypatch, name):
    """The launch-saving forms of the step change WHERE work runs, not what is computed: Adam as the last hop's epilogue
    (+ the projection weights' spans as extra workgroups), the planner / source-bit pass / weight packing on a second
    stream, the adjoint sources written by the head backward (or merged inside the weight-gradient launch), the weight
    gradients' slab reduce inside the adjoint's first hop launch, the loss summed by an extra workgroup of the Adam hop (or inside
    the BPR-head launch), the head's feature
    blocks in a launch of their own on the second stream beside the forward hops, the active rows' layer means evaluated by the
    head's main-stream launch instead of a rows launch -- each switched off gives bitwise the same losses, parameters and
    Adam moments after three steps. The two-launch SELL-64 hop sums long rows in another order: equal to round-off.
    Fixtures: ml3 (recdim 32: the generic head kernels), normal (recdim 64, three modalities: the fused 16-row head, the sources
    written by the head backward) and kwai (recdim 64, one modality)."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden(name)

    def run(env):
        for k in ("ELIMREC_FUSE_ADAM", "ELIMREC_AUX_STREAM", "ELIMREC_SLAB_TIERED", "ELIMREC_FUSED_HEAD", "ELIMREC_FUSE_MERGE", "ELIMREC_MERGE_FIRST",
                  "ELIMREC_FUSE_REDUCE", "ELIMREC_HEAD_SOURCES", "ELIMREC_FUSE_BWDW", "ELIMREC_HEAD_SPLIT", "ELIMREC_ROWS_IN_HEAD", "ELIMREC_LOSS_LATE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model, _ = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        eng = ColumnShardEngine(model)
        tr = ColumnShardTrainer(eng, opt)
        losses = [tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg"))).clone() for t in (1, 2, 3)]
        if not env:
            assert eng._sources_in_head() == (int(g["recdim"]) == 64)      # recdim 64: the fused forms are what the default step runs
        eng.sync_to_model()
        st = eng.optimizer_state()
        return (torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                st["exp_avg"], st["exp_avg_sq"])

    base = run({})
    for env in ({"ELIMREC_LOSS_LATE": "0"}, {"ELIMREC_FUSE_ADAM": "0"}, {"ELIMREC_AUX_STREAM": "0"}, {"ELIMREC_FUSE_ADAM": "0", "ELIMREC_AUX_STREAM": "0"},
                {"ELIMREC_HEAD_SOURCES": "0"}, {"ELIMREC_HEAD_SOURCES": "0", "ELIMREC_FUSE_MERGE": "0"}, {"ELIMREC_FUSE_REDUCE": "0"}, {"ELIMREC_FUSE_BWDW": "0"},
                {"ELIMREC_HEAD_SPLIT": "0"}, {"ELIMREC_ROWS_IN_HEAD": "0"},
                {"ELIMREC_HEAD_SOURCES": "0", "ELIMREC_FUSE_MERGE": "0", "ELIMREC_FUSE_REDUCE": "0", "ELIMREC_AUX_STREAM": "0"}):
        other = run(env)
        assert torch.equal(base[0], other[0]), env
        for k in base[1]:
            assert torch.equal(base[1][k], other[1][k]), (env, k)
        assert torch.equal(base[2], other[2]) and torch.equal(base[3], other[3]), env
    other = run({"ELIMREC_FUSED_HEAD": "0"})           # two batched GEMM launches + separate loss sum: same arithmetic?
    assert (base[0] - other[0]).abs().max() < 1e-6
    other = run({"ELIMREC_SLAB_TIERED": "0"})
    assert (base[0] - other[0]).abs().max() < 1e-6
    for k in base[1]:
        assert (base[1][k] - other[1][k]).abs().max() < 2e-5, k


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_shapes_column_shard_vs_row_major_trainer(seed):
    """Random small graphs, batch sizes that are not multiples of any tile height, 2-4 layers, the three bipartite
    adjacencies, recdim 64 (fused 16-row head, wave-tile hops, Adam epilogue, second stream) and 32 (batched-GEMM head):
    two steps of the column-shard trainer against the row-major trainer on the unfolded form (dist.py,
    --propagation=bipartite --head_rows=all: other hop, head and optimizer kernels, all tables through the graph)
    -- losses 1e-5, parameters 2e-5."""
    import os
    from helpers import ROOT
    from elimrec_amd import (ColumnShardEngine, ColumnShardTrainer, Configurator, EliMRec, FusedAdam, Logger,
                             PairwiseSamplerV2, SyntheticDataset, set_seed)
    from elimrec_amd.dist import DataParallelTrainer
    rs = np.random.RandomState(100 + seed)
    U, I = int(rs.randint(40, 900)), int(rs.randint(60, 2500))
    E = int(rs.randint(max(U, I) * 3, max(U, I) * 10))
    dims = tuple(int(4 * rs.randint(2, 36)) for _ in range(3))
    recdim = 64 if seed % 3 else 32
    L = int(rs.randint(2, 5))
    B = int(rs.randint(5, 700))
    adj = str(rs.choice(["pre", "gcmc", "plain"]))
    Logger.logger = Logger(show_in_console=False)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        argv = ["x", "--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=%d" % recdim,
                "--layer_num=%d" % L, "--adj_type=%s" % adj, "--verbose=0"]
        cfgs = {kind: Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                                   argv=argv + (["--propagation=bipartite", "--head_rows=all"] if kind == "rows" else []))
                for kind in ("rows", "slab")}
        cfg = cfgs["slab"]
        ds = SyntheticDataset(U, I, E, feat_dims=dims, seed=seed)
        u, p, n = PairwiseSamplerV2(ds, batch_size=B, device=DEV, seed=seed).sample_epoch()
        out = []
        for kind in ("rows", "slab"):
            set_seed(7)
            model = EliMRec(cfgs[kind], ds).to(DEV)
            opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
            if kind == "rows":
                tr, eng = DataParallelTrainer(model, opt), None
            else:
                eng = ColumnShardEngine(model)
                tr = ColumnShardTrainer(eng, opt)
            losses = [float(tr.step(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B])) for i in range(2)]
            out.append((losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    finally:
        os.chdir(cwd)
    (l0, p0), (l1, p1) = out
    tol = 1e-4 if adj == "plain" else 1e-5            # un-normalised adjacency: the embeddings grow by ~degree per hop
    assert max(abs(a - b) for a, b in zip(l0, l1)) < tol, (l0, l1, U, I, B, L, adj, recdim)
    for k in p0:
        assert (p0[k] - p1[k]).abs().max() < 2e-5 * (10 if adj == "plain" else 1), (k, U, I, B, L, adj, recdim)


# ----------------------------------------------------------------------------- two PROCESSES, one GPU
def _two_proc_worker(rank, world, port, out_dir, feature_shard="replicated", name="ml3", extra=()):
    import os, sys
    import torch.distributed as dist
    from helpers import ROOT
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)        # RCCL refuses two ranks on one device
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden(name)
    model, _ = build_model_from_fixture(g, DEV, extra_argv=extra)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model, feature_shard=feature_shard)
    tr = ColumnShardTrainer(eng, opt, world_size=world, rank=rank)
    losses, batches = [], []
    for t in (1, 2):
        u, p, n = (_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg"))
        h = len(u) // world
        sl = slice(rank * h, (rank + 1) * h)
        batches.append((u[sl].clone(), p[sl].clone(), n[sl].clone()))
    tr.plan_lookup(batches[:1])                     # step 1 planned ahead, step 2 reads its split sizes from the device
    for b in batches:
        losses.append(float(tr.global_loss(tr.step(*b))))
    if feature_shard == "row":
        assert tr.lookup and tr.lookup_syncs == 1 and tr.xgmi_bytes["all_to_all_lookup"] > 0
        assert eng.fshard.table.shape[0] < model.num_users + model.num_items
        assert eng.fold_mode == "sharded" and model._ws["fold"] is None      # no rank ever computed or held a full S_m
    if "--feature_load=block" in extra:
        # the raw features were READ block by block: this rank asked for its own item block of every table whose width splits
        # over the ranks (the distributed fold's column slices must be multiples of 4 floats) -- and for nothing else
        I = model.num_items
        mine = (I * rank // world, I * (rank + 1) // world)
        for m_ in model._mods:
            fb = getattr(model, m_ + "_feat")
            splits = fb.shape[1] % world == 0 and (fb.shape[1] // world) % 4 == 0
            assert fb.blocks == [mine if splits else (0, I)], (m_, fb.blocks)
    # validation on the tables of the last forward: user-sharded with replicated tables, ITEM-sharded with row shards
    model.fusion_mode, model.predict_type = "rubi", "TIE"
    res, _ = model.test()
    users = g["eval_users"].tolist()
    pred = model.predict(users).numpy()
    if feature_shard == "row":
        assert model._eval_shard is not None and model._eval_shard_Y.shape[0] < model.num_users + model.num_items
        with pytest.raises(RuntimeError):
            model.all_items
    eng.sync_to_model()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.array(losses), eval_result=res, predict=pred,
             **{k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    if feature_shard == "row":
        # the reference's tie order over ITEM-SHARDED tables: every item of my shard gets a twin (equal rows score equally under
        # every user), so every user's merged K + 1 best hold exact ties; those rows' all-gathered score rows are ranked by the
        # device replay of evaluate.h:26-33 -- the lists must be the oracle's ranking of the rows predict() returns
        from elimrec_amd import ops
        from oracle import eval_oracle as evo
        sh, Y, U_ = model._eval_shard, model._eval_shard_Y, model.num_users
        nloc = Y.shape[0] - U_
        Y[U_ + 1:U_ + nloc:2] = Y[U_:U_ + nloc - 1:2].clone()
        ops.row_sqnorms(Y, model.latent_dim, 1 + model.S, sh.backend.sqn)
        evalr = model.test_evaluator.evaluator
        evalr._dev_cache.clear()
        ev_users = g["evalbatch/users"].tolist()
        K = evalr.max_top
        assert evalr.tie_order == "reference"
        _, idx, _ = evalr.evaluate_batch(model, ev_users, return_topk=True)
        assert evalr.tie_rows_replayed >= len(ev_users) // 2
        tp_, ti_ = evalr._batch_csr(ev_users, evalr.user_pos_train, DEV, unique=False)
        sc = torch.empty(len(ev_users), model.num_items, device=DEV)
        model.predict_device(ev_users, scores=sc, train_ptr=tp_, train_items=ti_)
        tp0, ti0 = evo.truth_to_csr([[0]] * len(ev_users))
        _, want = evo.evaluate_matrix(sc.cpu().numpy(), tp0, ti0, [1], K)
        assert np.array_equal(idx.cpu().numpy(), want)
    dist.destroy_process_group()


@pytest.mark.parametrize("feature_shard,name,extra", [("replicated", "ml3", ()), ("row", "ml3", ()),
                                                      ("row", "ablate", ("--propagation=folded",)),
                                                      ("row", "ml3", ("--lean_tables=1", "--feature_load=block"))],
                         ids=["replicated", "row", "row-wide-form", "row-lean-block-loader"])
def test_two_processes_on_one_gpu_equal_one_process(tmp_path, feature_shard, name, extra):
    """The real engine and the real trainer in two PROCESSES (both on cuda:0; gloo group, collectives staged through the host
    because RCCL refuses duplicate devices): every collective of a column-sharded step executes between processes. Both ranks
    end with the same full model, equal to one process on the concatenated batch. feature_shard = row: each process holds
    half the rows of the folded constants and the step's variable-size all_to_all brings the other half's active rows.
    row-wide-form: the `ablate` fixture (adjacency with a diagonal, mean fusion) in the wide form, two column slices of both
    halves of the wide tables. row-lean-block-loader: lean tables with the raw features read block by block (--feature_load=block):
    every process reads its own item block of a feature table and nothing else of it."""
    import torch.multiprocessing as mp
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    world = 2
    port = 33500 + (os.getpid() % 2000) + (3 if feature_shard == "row" else 0)
    mp.spawn(_two_proc_worker, args=(world, port + (7 if extra else 0), str(tmp_path), feature_shard, name, extra), nprocs=world, join=True)
    rs = [dict(np.load(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    for k in rs[0]:
        assert np.array_equal(rs[0][k], rs[1][k]), k
    g = load_golden(name)
    model, _ = build_model_from_fixture(g, DEV, extra_argv=extra)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt)
    losses = []
    for t in (1, 2):
        u, p, n = (_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg"))
        mlen = (len(u) // world) * world
        losses.append(float(tr.step(u[:mlen], p[:mlen], n[:mlen])))
    model.fusion_mode, model.predict_type = "rubi", "TIE"
    res, _ = model.test()
    pred = model.predict(g["eval_users"].tolist()).numpy()
    eng.sync_to_model()
    assert np.allclose(rs[0]["losses"], losses, atol=1e-5)
    for k, v in model.state_dict().items():
        assert np.abs(rs[0][k] - v.detach().cpu().numpy()).max() < 2e-5, k
    # evaluation of the two-process job (item-sharded when the constants are row-sharded) == one process
    assert np.abs(rs[0]["predict"] - pred).max() < 2e-6
    assert np.abs(rs[0]["eval_result"] - res).max() < 1e-7, (rs[0]["eval_result"], res)


# ----------------------------------------------------------------------------- one-rank RCCL group, multi-rank code path
def _rccl_one_rank_worker(rank, port, out_dir, feature_shard="replicated"):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["ELIMREC_SHARD_MULTI"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden("ml3")
    model, _ = build_model_from_fixture(g, DEV)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model, feature_shard=feature_shard)
    tr = ColumnShardTrainer(eng, opt, world_size=1, rank=0)
    assert tr.multi and eng.multi and tr.lookup == (feature_shard == "row")
    losses = []
    for t in (1, 2, 3):
        u, p, n = (_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg"))
        losses.append(float(tr.global_loss(tr.step(u, p, n))))
    eng.sync_to_model()
    np.savez(os.path.join(out_dir, "rccl.npz"), losses=np.array(losses), xgmi=np.array(sorted(tr.xgmi_bytes)),
             **{k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("feature_shard", ["replicated", "row"])
def test_multi_rank_step_over_a_one_rank_rccl_group(tmp_path, feature_shard):
    """ELIMREC_SHARD_MULTI=1: ONE rank runs the multi-rank step -- all_gather of the active ids, both all_to_alls, the
    asynchronous all_reduce of the weight gradients, the rank-ordered merge, the separate optimizer launch -- over a real
    RCCL (backend "nccl") process group on the GPU: every collective call of the step as the 8-GPU job issues it (tensor
    shapes, dtypes, contiguity, stream hand-over, async handles). Three steps equal the one-rank fast path to round-off."""
    import torch.multiprocessing as mp
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    port = 35500 + (os.getpid() % 2000) + (3 if feature_shard == "row" else 0)
    mp.spawn(_rccl_one_rank_worker, args=(port, str(tmp_path), feature_shard), nprocs=1, join=True)
    got = dict(np.load(tmp_path / "rccl.npz"))
    assert set(got.pop("xgmi").tolist()) == {"all_gather", "all_to_all_fwd", "all_to_all_bwd", "all_reduce"} | (
        {"all_to_all_lookup"} if feature_shard == "row" else set())        # ... + the uint8 all_to_all with split sizes
    g = load_golden("ml3")
    model, _ = build_model_from_fixture(g, DEV)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt)
    assert not tr.multi
    losses = [float(tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))) for t in (1, 2, 3)]
    eng.sync_to_model()
    assert np.abs(got.pop("losses") - np.array(losses)).max() < 1e-5
    for k, v in model.state_dict().items():
        assert np.abs(got[k] - v.detach().cpu().numpy()).max() < 2e-5, k


# ----------------------------------------------------------------------------- row-sharded constant tables (lookup)
@pytest.mark.parametrize("W,dtype", [(1, "f32"), (2, "f32"), (5, "f16"), (8, "bf16"), (8, "f32"), (3, "f16")])
def test_lookup_counts_pack_unpack_vs_torch(W, dtype):
    """csrc/lookup.hip on W emulated ranks: counts == a torch count of every list's ids per owner; pack -> (all_to_all by
    hand) -> unpack gives, for every requester, exactly the owner tables' rows (rounded to the storage dtype) of ITS
    active nodes in ITS order, and c to fp32 (16-bit storage: hi + lo, 2^-16 relative at worst for bf16)."""
    from elimrec_amd.lookup import DTYPES, FeatureShard, RowOwnerMap
    U, I, R = 700, 1900, 300
    dims = (24, 8, 12)
    g = torch.Generator().manual_seed(W)
    tabs = [torch.randn(U + I, D, generator=g).to(DEV) for D in dims]
    c = torch.rand(U + I, generator=g).to(DEV)
    ub = None if W != 5 else [0, 10, 10, 300, 650, 700]          # an empty user block, uneven blocks
    owners = RowOwnerMap(U, I, W, ub=ub)
    shards = [FeatureShard(owners, o, tabs, c, dtype=dtype) for o in range(W)]
    lists = []
    for r in range(W):
        n_act = int(torch.randint(1, R + 1, (1,), generator=g)) if r else R       # list 0 is full, the others ragged
        ids = torch.randperm(U + I, generator=g)[:n_act].sort().values
        lists.append(torch.cat([ids, torch.full((R - n_act,), -(1 << 30), dtype=torch.int64)]).to(torch.int32))
    acts = torch.stack(lists).to(DEV)
    counts = shards[0].counts(acts).cpu().numpy()
    owner_of = np.zeros(U + I, np.int64)
    for o in range(W):
        owner_of[owners.nodes(o)] = o
    for r in range(W):
        valid = lists[r][lists[r] >= 0].numpy()
        assert np.array_equal(counts[r], np.bincount(owner_of[valid], minlength=W)), r
    rb = shards[0].row_bytes
    packed = []
    for o in range(W):
        buf = torch.zeros(W * R * rb, dtype=torch.uint8, device=DEV)
        off = torch.zeros(W + 1, dtype=torch.int32, device=DEV)
        shards[o].pack(acts, buf, off)
        assert np.array_equal(off.cpu().numpy(), np.concatenate([[0], np.cumsum(counts[:, o])]))
        packed.append(buf)
    tdt = DTYPES[dtype][1]
    full = torch.cat(tabs, dim=1)
    for q in range(W):
        recv = torch.cat([packed[o][int(counts[:q, o].sum()) * rb:(int(counts[:q, o].sum()) + int(counts[q, o])) * rb] for o in range(W)])
        S = torch.full((R, sum(dims)), 7.0, device=DEV)
        cr = torch.full((R,), 7.0, device=DEV)
        shards[q].unpack(acts[q], recv, S, cr)
        valid = lists[q][lists[q] >= 0].long().to(DEV)
        n = len(valid)
        assert torch.equal(S[:n], full[valid].to(tdt).float())
        assert (S[n:] == 7.0).all() and (cr[n:] == 7.0).all()            # rows behind the valid prefix untouched
        tol = 0.0 if dtype == "f32" else (2.0 ** -20 if dtype == "f16" else 2.0 ** -15)
        assert (cr[:n] - c[valid]).abs().max().item() <= tol
    if W == 1:       # one rank: straight from the local table
        S2, c2 = torch.empty_like(S), torch.empty_like(cr)
        shards[0].unpack(acts[0], None, S2, c2, direct=True)
        assert torch.equal(S2[:n], S[:n]) and torch.equal(c2[:n], cr[:n])


@pytest.mark.parametrize("W", [1, 2, 4])
@pytest.mark.parametrize("name", ["ml3", "kwai"])
def test_row_sharded_constants_equal_replicated_bitwise(W, name):
    """--feature_shard=row on W emulated ranks (each holds 1/W of the rows of S_m / c; every step fetches the rows of its
    active nodes: counts -> pack -> exchange -> unpack) against the replicated tables on the same W ranks: the looked-up
    rows are the same fp32 values consumed in the same order, so losses, embeddings and weights agree BITWISE; and both
    match the reference fixture's losses (kwai: recdim 64, the fused head; ml3: recdim 32, the batched GEMMs)."""
    from elimrec_amd import ColumnShardEngine, FusedAdam
    g = load_golden(name)
    B = (len(g["step1/users"]) // W) * W

    def make(rank, mode):
        model, _ = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        eng = ColumnShardEngine(model, feature_shard=mode)
        eng.cs_setup(W, rank, opt)
        return model, eng

    out = {}
    for mode in ("replicated", "row"):
        ranks = [make(q, mode) for q in range(W)]
        assert all(e.lookup == (mode == "row") for _, e in ranks)
        if mode == "row" and W > 1:
            rows = sum(e.fshard.table.shape[0] for _, e in ranks)
            assert rows == int(g["num_users"]) + int(g["num_items"])          # a partition of the rows
        losses = []
        for t in (1, 2, 3):
            u, p, n = (_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg"))
            h = min(B, len(u)) // W          # (the fixture's third batch is the epoch's ragged tail)
            losses.append(float(_emulated_step([e for _, e in ranks], [(u[q * h:(q + 1) * h], p[q * h:(q + 1) * h], n[q * h:(q + 1) * h])
                                                                        for q in range(W)])))
        out[mode] = (losses, torch.cat([e.master[e.cur].dense() for _, e in ranks], dim=1),
                     {k: v.clone() for k, v in ranks[0][0].state_dict().items() if not k.startswith("embedding_")})
    assert out["row"][0] == out["replicated"][0]
    assert torch.equal(out["row"][1], out["replicated"][1])
    for k, v in out["replicated"][2].items():
        assert torch.equal(out["row"][2][k], v), k
    if W == 1:
        for t in (1, 2, 3):
            assert abs(out["row"][0][t - 1] - float(g["step%d/loss" % t])) < 1e-5


@pytest.mark.parametrize("name", ["kwai", "ml3"])
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_16bit_feature_storage_equals_fp32_engine_on_rounded_constants(dtype, name):
    """--feature_dtype=f16|bf16 (BASELINE.json configs[4]: "fp16" features): the constants are STORED in 16 bits and widened
    when a step looks its rows up; everything else is the fp32 path. So the run equals, to the last bits of c's hi + lo
    split (tolerance 2e-6), the fp32 engine on a model whose S_m were rounded to that dtype -- the rounding oracle of
    this mode -- and differs from the unrounded fp32 run by the storage error (stated tolerance: loss 2e-3)."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden(name)            # kwai: recdim 64, the fused head; ml3: recdim 32, the batched GEMMs
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    res = {}
    for mode in ("stored", "rounded", "plain"):
        model, _ = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        eng = ColumnShardEngine(model, feature_dtype=dtype if mode == "stored" else "f32")
        tr = ColumnShardTrainer(eng, opt)
        if mode == "rounded":
            fold = model._ws["fold"]
            for k in model._mods:
                fold[k].copy_(fold[k].to(tdt).float())
        losses = [float(tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))) for t in (1, 2, 3)]
        res[mode] = (losses, eng.master[eng.cur].dense().clone())
    assert np.abs(np.array(res["stored"][0]) - np.array(res["rounded"][0])).max() < 2e-6
    assert (res["stored"][1] - res["rounded"][1]).abs().max().item() < 2e-5
    assert res["stored"][0] != res["plain"][0]                                   # the storage really is 16-bit
    assert np.abs(np.array(res["stored"][0]) - np.array(res["plain"][0])).max() < 2e-3


@pytest.mark.parametrize("form", ["split+rows", "split", "one-launch"])
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_16bit_constants_read_by_the_head_equal_the_widening_pass_bitwise(dtype, form, monkeypatch):
    """One rank, 16-bit constants: by default the fused head reads the packed 16-bit rows where they lie and widens in registers
    (elimrec_head_fwd_fused_src16; the step keeps the fp32 step's shape: feature blocks beside the hops, rows evaluated in the
    head's launch); ELIMREC_DIRECT16=0 keeps the earlier form -- a widening pass over the batch's rows (elimrec_lookup_unpack)
    in front of a head that reads its fp32 output. The same arithmetic on the same values: 14 steps (the one-call program takes
    over midway), every loss, the embedding tables, every other parameter and both Adam moments bit for bit."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden("kwai")
    bs = [tuple(_t(g["step%d/%s" % (1 + k % 2, key)]) for key in ("users", "pos", "neg")) for k in range(14)]
    res = {}
    # the head's three launch forms under the direct reads: feature blocks on the second stream + rows in the head's launch
    # (default), feature blocks on the second stream + a rows launch, everything in one head launch
    if form != "split+rows":
        monkeypatch.setenv("ELIMREC_ROWS_IN_HEAD", "0")
    if form == "one-launch":
        monkeypatch.setenv("ELIMREC_HEAD_SPLIT", "0")
    for direct in ("1", "0"):
        monkeypatch.setenv("ELIMREC_DIRECT16", direct)
        model, _ = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        eng = ColumnShardEngine(model, feature_dtype=dtype)
        tr = ColumnShardTrainer(eng, opt)
        losses = [tr.step(*b) for b in bs]
        assert eng._direct16 == (direct == "1") and eng._fused_head_ok()
        st = eng.optimizer_state()
        res[direct] = ([float(x) for x in torch.stack(losses).cpu()], eng.master[eng.cur].dense().clone(),
                       {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("embedding_")},
                       st["exp_avg"].clone(), st["exp_avg_sq"].clone(), tr._native_state()["native_steps"])
    assert res["1"][0] == res["0"][0]
    assert torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][3], res["0"][3]) and torch.equal(res["1"][4], res["0"][4])
    for k, v in res["0"][2].items():
        assert torch.equal(res["1"][2][k], v), k
    assert res["1"][5] > 0                                           # the direct form runs as a one-call program too


@pytest.mark.parametrize("name", ["ml3", "kwai", "gcmc"])
def test_hop_kernel_fold_of_the_constants_equals_the_model_fold(name, monkeypatch):
    """The distributed fold's arithmetic (ColumnShardEngine._horner_mean: t <- X0 + A t with the slab hop kernels, widths
    padded to the plan's lane-group geometry) on one rank against EliMRec._fold_constants (the bipartite propagation
    kernels): S_m and c agree to 1e-6 -- two summation orders of the same mean_k A^k [0 ; F_m]."""
    from elimrec_amd import ColumnShardEngine, FusedAdam
    g = load_golden(name)
    model, _ = build_model_from_fixture(g, DEV)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    ref = ColumnShardEngine(model, feature_shard="row")
    ref.cs_setup(1, 0, opt)
    assert ref.fold_mode == "model"
    want = ref.fshard.table.clone()
    monkeypatch.setenv("ELIMREC_FOLD", "sharded")
    model2, _ = build_model_from_fixture(g, DEV)
    eng = ColumnShardEngine(model2, feature_shard="row")
    eng.cs_setup(1, 0, FusedAdam(model2.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"])))
    assert eng.fold_mode == "sharded" and model2._ws["fold"] is None
    got = eng.fshard.table
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 1e-6 * max(1.0, want.abs().max().item())
    # and a step on the hop-folded constants still matches the reference fixture
    u, p, n = (_t(g["step1/%s" % k]) for k in ("users", "pos", "neg"))
    acts = eng.cs_plan(u, p, n).view(1, -1)
    eng.cs_forward(acts)
    assert abs(float(eng.cs_head(None)) - float(g["step1/loss"])) < 1e-5


@pytest.mark.parametrize("name", ["kwai", "ml3", "gcmc"])
def test_native_step_program_equals_python_issued_steps(name, monkeypatch):
    """The step as ONE host call (csrc/program.hip: the traced launches, stream hand-overs and patched per-step arguments,
    issued from C) against the same steps issued launch by launch from Python: 40 steps on changing batches (every step a
    different set of index tensors), losses and every parameter bit for bit; predict() on the tables of the last native step;
    a batch of another size in between takes the ordinary path and the programs keep working after it."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden(name)
    base = [tuple(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")) for t in (1, 2)]
    n0 = min(len(b[0]) for b in base)
    gen = torch.Generator().manual_seed(0)
    batches = []
    for s in range(40):
        b = base[s % 2]
        perm = torch.randperm(n0, generator=gen).to(DEV)
        size = n0 - 5 if s == 25 else n0
        batches.append(tuple(x[:n0][perm][:size].clone() for x in b))
    out = {}
    for native in ("0", "1"):
        monkeypatch.setenv("ELIMREC_NATIVE_STEP", native)
        model, _ = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        eng = ColumnShardEngine(model)
        tr = ColumnShardTrainer(eng, opt)
        losses = torch.stack([tr.step(*b) for b in batches]).cpu().numpy()
        st = tr._native_state()
        if native == "1":
            assert st["failed"] is None, st["failed"]
            assert st["native_steps"] >= 20, st
        else:
            assert st["native_steps"] == 0
        model.fusion_mode, model.predict_type = "rubi", "TIE"
        pred = model.predict(g["eval_users"].tolist())
        eng.sync_to_model()
        out[native] = (losses, pred, {k: v.detach().clone() for k, v in model.state_dict().items()},
                       opt.export_state(model.named_parameters()))
    assert np.array_equal(out["0"][0], out["1"][0])
    assert torch.equal(out["0"][1], out["1"][1])
    for k, v in out["0"][2].items():
        assert torch.equal(out["1"][2][k], v), k
    for k, st0 in out["0"][3].items():
        assert out["1"][3][k]["step"] == st0["step"] and torch.equal(out["1"][3][k]["exp_avg"], st0["exp_avg"]), k


def _native_multi_worker(rank, port, out_dir, feature_shard, native):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["ELIMREC_SHARD_MULTI"] = "1"
    os.environ["ELIMREC_NATIVE_STEP"] = os.environ["ELIMREC_NATIVE_COMM"] = native
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden("kwai")
    model, _ = build_model_from_fixture(g, DEV)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model, feature_shard=feature_shard)
    tr = ColumnShardTrainer(eng, opt, world_size=1, rank=0)
    base = [tuple(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")) for t in (1, 2)]
    n0 = min(len(b[0]) for b in base)
    gen = torch.Generator().manual_seed(0)
    batches = []
    for s in range(30):
        perm = torch.randperm(n0, generator=gen).to(DEV)
        batches.append(tuple(x[:n0][perm].clone() for x in base[s % 2]))
    tr.plan_lookup(batches)
    losses = torch.stack([tr.step(*b) for b in batches]).cpu().numpy()
    st = tr._native_state()
    assert (tr._native_comm() is not None) == (native == "1")
    if native == "1":
        assert st["failed"] is None, st["failed"]
        assert st["native_steps"] >= 20, st
        names = [tr._native["programs"][k][0]._keep for k in tr._native["programs"]]
        assert names
    eng.sync_to_model()
    np.savez(os.path.join(out_dir, "native%s.npz" % native), losses=losses, lookup_syncs=np.array(tr.lookup_syncs),
             **{k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("feature_shard", ["replicated", "row"])
def test_native_multi_rank_program_with_rccl_calls(tmp_path, feature_shard):
    """The MULTI-rank step as one host call: the exchanges are the library's own RCCL calls (elimrec_comm_all_gather /
    all_to_all / all_to_all_v / all_reduce_f32 on a communicator created from a broadcast unique id), enqueued on our
    streams and listed in the step's program beside the kernels. One rank over a real RCCL communicator: 30 steps from the
    program equal, bit for bit, the same steps issued from Python with torch.distributed's collectives."""
    import torch.multiprocessing as mp
    port = 37500 + (os.getpid() % 2000) + (3 if feature_shard == "row" else 0)
    for native in ("0", "1"):
        mp.spawn(_native_multi_worker, args=(port + int(native), str(tmp_path), feature_shard, native), nprocs=1, join=True)
    a, b = dict(np.load(tmp_path / "native0.npz")), dict(np.load(tmp_path / "native1.npz"))
    assert int(a["lookup_syncs"]) == 0 and int(b["lookup_syncs"]) == 0
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def _long_wanted_multi_worker(rank, port, out_dir, wanted):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["ELIMREC_SHARD_MULTI"] = "1"
    os.environ["ELIMREC_LONG_WANTED"] = wanted
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, EliMRec, FusedAdam, SyntheticDataset, set_seed
    cfg = make_config(["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0"])
    ds = SyntheticDataset(300, 2500, 40000, feat_dims=(16, 8, 12), seed=3)
    gen = torch.Generator().manual_seed(5)
    B = 256
    batches = [(torch.randint(0, 300, (B,), generator=gen).to(DEV), torch.randint(0, 2500, (B,), generator=gen).to(DEV),
                torch.randint(0, 2500, (B,), generator=gen).to(DEV)) for _ in range(14)]
    set_seed(11)
    model = EliMRec(cfg, ds).to(DEV)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt, world_size=1, rank=0)
    assert tr.multi and eng._long_wanted_only() == (wanted == "1") and eng.plan.n_long > 100 and eng._split_share > 0.3
    losses = torch.stack([tr.step(*b) for b in batches]).cpu().numpy()
    st = tr._native_state()
    assert st["failed"] is None and st["native_steps"] > 0, st
    eng.sync_to_model()
    np.savez(os.path.join(out_dir, "wanted%s.npz" % wanted), losses=losses,
             **{k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    dist.destroy_process_group()


def test_multi_rank_step_with_the_split_rows_of_the_batches_only(tmp_path):
    """Several ranks (here: one rank's multi-rank step over a real RCCL communicator, the step's program engaged): hop L's split
    rows for the rows of every rank's batch only -- the bitmap of the gathered ids, made on the second stream, is the wanted-rows
    bitmap, and the main stream waits for it instead of running the split rows ahead of the id exchange. Bitwise the steps of the
    default order on a graph whose split rows hold most of the non-zeros (where the engine chooses this form by itself)."""
    import torch.multiprocessing as mp
    port = 38300 + (os.getpid() % 1000)
    for wanted in ("0", "1"):
        mp.spawn(_long_wanted_multi_worker, args=(port + int(wanted), str(tmp_path), wanted), nprocs=1, join=True)
    a, b = dict(np.load(tmp_path / "wanted0.npz")), dict(np.load(tmp_path / "wanted1.npz"))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("W,R,dl", [(1, 50, 64), (2, 300, 32), (8, 1000, 8), (5, 7, 4)])
def test_peer_cols_to_rows_and_rows_bitmap_vs_torch(W, R, dl):
    """The two exchange helpers of the multi-rank step: received column slices [W x R x 2 x dl] -> two row views (what the
    torch permuted copies did), and the row bitmap of W sorted id lists == the words elimrec_slab_merge_rows writes for the
    same lists (an all-padding list and duplicates across lists included)."""
    from elimrec_amd import ops, slab
    g = torch.Generator().manual_seed(W * 1000 + R)
    recv = torch.randn(W, R, 2 * dl, generator=g).to(DEV)
    pair = torch.full((R, 2, W * dl + 4), 7.0, device=DEV)                     # strided destinations, padding untouched
    ops.peer_cols_to_rows(recv, pair[:, 0, :W * dl], pair[:, 1, :W * dl])
    r = recv.view(W, R, 2, dl)
    assert torch.equal(pair[:, 0, :W * dl], r[:, :, 0].permute(1, 0, 2).reshape(R, W * dl))
    assert torch.equal(pair[:, 1, :W * dl], r[:, :, 1].permute(1, 0, 2).reshape(R, W * dl))
    assert (pair[:, :, W * dl:] == 7.0).all()
    U, I = 700, 1900
    N = U + I
    lists = []
    for q in range(W):
        n_act = 0 if (q == 1 and W > 2) else int(torch.randint(1, min(R, N) + 1, (1,), generator=g))
        ids = torch.randperm(N, generator=g)[:n_act].sort().values
        lists.append(torch.cat([ids, torch.full((R - n_act,), -(1 << 30), dtype=torch.int64)]).to(torch.int32))
    acts = torch.stack(lists).to(DEV)
    mask = torch.full(((N + 31) // 32 + 2,), -1, dtype=torch.int32, device=DEV)
    slab.rows_bitmap(acts, N, mask)
    want = np.zeros(N, bool)
    for l in lists:
        want[l[l >= 0].numpy()] = True
    words = mask.cpu().numpy().view(np.uint32)[:(N + 31) // 32]
    got = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).reshape(-1)[:N]
    assert np.array_equal(got, want)
    # ... and the same words as the merge kernel leaves for these lists
    ns, w = slab.choose_slabs(dl)
    srcA, srcB = slab.SlabTable(N, ns, w, DEV), slab.SlabTable(N, ns, w, DEV)
    mask2 = torch.zeros_like(mask)
    slab.merge_rows(torch.randn(W * R, 2 * dl, generator=g).to(DEV), acts.reshape(-1), W, U, I, srcA, srcB, mask2)
    assert torch.equal(mask2[:(N + 31) // 32], mask[:(N + 31) // 32])


@pytest.mark.parametrize("W", [1, 2, 4, 8])
def test_fused_head_reads_the_peers_pieces_in_place(W):
    """elimrec_head_fwd_fused_peers: the fused head on the forward exchange's received buffer [W x R x (out0 dl | narrow dl)]
    leaves the bits of elimrec_peer_cols_to_rows followed by elimrec_head_fwd_fused on its two row views -- OutAct (block 0
    included, which the launch writes itself) and YAct, phases 0 and 2; a buffer whose pieces are not 4-float multiples or
    do not make 64 columns is refused."""
    from elimrec_amd import ColumnShardEngine, EliMRec, FusedAdam, SyntheticDataset, _lib, ops, set_seed
    cfg = make_config(["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0"])
    ds = SyntheticDataset(700, 1900, 9000, feat_dims=(128, 24, 64), seed=1)
    set_seed(7)
    model = EliMRec(cfg, ds).to(DEV)
    eng = ColumnShardEngine(model)
    eng.cs_setup(1, 0, FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"]))
    assert eng._fused_head_ok()
    gen = torch.Generator().manual_seed(3)
    B = 300
    u = torch.randint(0, 700, (B,), generator=gen).to(DEV)
    p = torch.randint(0, 1900, (B,), generator=gen).to(DEV)
    n = torch.randint(0, 1900, (B,), generator=gen).to(DEV)
    eng._rows_in_head_on = False                    # the rows launch leaves out0 / narrow of the active rows for this test to cut up
    acts = eng.cs_plan(u, p, n).view(1, -1)
    eng.cs_forward(acts)
    torch.cuda.synchronize()
    ws = model._ws
    R, d, dl = model._plan_n, 64, 64 // W
    out0, nar = ws["OutAct"][:R, :d].clone(), eng.nar_act[:R].clone()
    recv = torch.empty(W, R, 2 * dl, device=DEV)
    for q in range(W):
        recv[q, :, :dl] = out0[:, q * dl:(q + 1) * dl]
        recv[q, :, dl:] = nar[:, q * dl:(q + 1) * dl]
    got = {}
    for form in ("rows", "peers"):
        for phase in (0, 2):
            ws["OutAct"][:R].fill_(-3.0)
            ws["YAct"][:R].fill_(-3.0)
            if form == "rows":
                pair = torch.zeros(R, 2, d, device=DEV)
                ops.peer_cols_to_rows(recv, pair[:, 0, :], pair[:, 1, :])
                eng._out0_src, eng._nar_src, eng._peer_src = pair[:, 0, :], pair[:, 1, :], None
            else:
                eng._out0_src, eng._nar_src, eng._peer_src = None, None, recv
            eng._head_fused_call(ws, R, phase)
            torch.cuda.synchronize()
            got[form, phase] = (ws["OutAct"][:R].clone(), ws["YAct"][:R].clone())
    na = int(ws["seg_info"][0])
    for phase in (0, 2):
        for a, b in zip(got["rows", phase], got["peers", phase]):
            assert torch.equal(a[:na], b[:na])
        assert torch.equal(got["peers", phase][0][:na, :d], out0[:na])
    eng._out0_src = eng._nar_src = eng._peer_src = None
    if W == 1:
        bad = torch.zeros(3, R, 2 * 20, device=DEV)          # 3 x 20 columns are not a row of 64
        with pytest.raises(RuntimeError, match="must make the 64 columns"):
            eng._peer_src = bad
            try:
                eng._head_fused_call(ws, R, 2)
            finally:
                eng._peer_src = None


# ----------------------------------------------------------------------------- adjacencies with a diagonal: the wide form
def test_wide_form_trains_the_norm_adjacency_fixture_on_the_column_shard_engine():
    """The fifth fixture (`ablate`: adj_type = norm, D^-1 (A + I); mean fusion; modality ablation 'va') on ColumnShardTrainer
    with --propagation=folded: the graph carries the E_u-borne and the E_i-borne part side by side in wide tables
    (csrc/wide.hip) -- losses 1e-5, parameters after Adam 2e-5, predict() after training 1e-5 against the reference's golden
    vectors, as the other four fixtures on this engine."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden("ablate")
    model, _ = build_model_from_fixture(g, DEV, extra_argv=["--propagation=folded"])
    assert model._wide and model._lazy and not model._bipartite
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt)
    assert eng.wide and eng.lookup
    steps = int(g["steps"])
    for t in range(1, steps + 1):
        loss = tr.step(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))
        assert abs(float(loss) - float(g["step%d/loss" % t])) < 1e-5, t
        if t in (1, steps):
            eng.sync_to_model()
            sd = model.state_dict()
            for k, v in sub(g, "after%d" % t).items():
                assert np.abs(sd[k].cpu().numpy() - v).max() < 2e-5, (t, k)
    model.fusion_mode, model.predict_type = "rubi", "TIE"
    got = model.predict(g["eval_users"].tolist()).numpy()
    assert np.abs(got - g["predict/rubi/TIE"]).max() < 1e-5


@pytest.mark.parametrize("adj_type,L,recdim,W", [("norm", 3, 64, 1), ("mean", 2, 32, 1), ("norm", 4, 64, 2), ("mean", 3, 32, 4)])
def test_wide_form_step_vs_oracle(adj_type, L, recdim, W):
    """One step of the wide form against the oracle on a seeded shape, adj_type norm and mean + I, layer counts 2-4, the fused
    (recdim 64) and the generic head, one rank and W emulated ranks with row-sharded constants: loss 1e-5, the embedding
    gradient and every weight gradient 1e-4 max-norm and row by row."""
    from elimrec_amd import ColumnShardEngine, EliMRec, FusedAdam, SyntheticDataset, set_seed
    from oracle import elimrec_oracle as eo
    U, I, E, dims, B = 700, 1900, 9000, (24, 8, 12), 256
    argv = ["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=%d" % recdim, "--verbose=0",
            "--adj_type=%s" % adj_type, "--layer_num=%d" % L, "--propagation=folded"]
    ds = SyntheticDataset(U, I, E, feat_dims=dims, seed=1)
    gen = torch.Generator().manual_seed(3)
    train = ds.train_matrix.tocoo()
    pick = torch.randint(0, train.nnz, (B,), generator=gen).numpy()
    u = torch.from_numpy(train.row[pick].astype(np.int64)); p = torch.from_numpy(train.col[pick].astype(np.int64))
    n = torch.randint(0, I, (B,), generator=gen)
    engines, init = [], None
    for q in range(W):
        cfg = make_config(argv + (["--feature_shard=row"] if W > 1 else []))
        set_seed(7)
        model = EliMRec(cfg, ds)
        init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
        model = model.to(DEV)
        eng = ColumnShardEngine(model)
        eng.cs_setup(W, q, FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"]))
        eng.keep_grad = True
        assert eng.wide
        engines.append(eng)
    h = B // W
    batches = [(u[q * h:(q + 1) * h].to(DEV), p[q * h:(q + 1) * h].to(DEV), n[q * h:(q + 1) * h].to(DEV)) for q in range(W)]
    # the emulated step without the update: forward, head, backward -- gradients against the oracle on the whole batch
    acts = torch.stack([e.cs_plan(*b).clone() for e, b in zip(engines, batches)])
    sends = [e.cs_forward(acts) for e in engines]
    sends = [None if s is None else s.clone() for s in sends]
    if W > 1:
        rb = engines[0].lookup_row_bytes
        counts = engines[0].cs_lookup_counts(acts).cpu().numpy()
        packed = [e.cs_lookup_pack(acts).clone() for e in engines]
        for q, e in enumerate(engines):
            e.cs_lookup_unpack(torch.cat([packed[o][int(counts[:q, o].sum()) * rb:(int(counts[:q, o].sum()) + int(counts[q, o])) * rb] for o in range(W)]))
    scale = torch.full((1,), 1.0 / W, device=DEV)
    losses, sends2, wgs = [], [], []
    for q, e in enumerate(engines):
        recv = None if W == 1 else torch.stack([sends[o][q] for o in range(W)])
        losses.append(e.cs_head(recv).clone())
        s2, wg = e.cs_backward_local(scale)
        sends2.append(s2.clone()); wgs.append(wg.clone())
    for q, e in enumerate(engines):
        e.cs_backward_hops(torch.stack([sends2[o][q] for o in range(W)]) if W > 1 else sends2[0], acts)
    tu, ti = ds.get_train_interactions()
    cfg = make_config(argv)
    adj = eo.build_adj(tu, ti, U, I, adj_type)
    feats = {m: eo.OracleEliMRec.normalize_features(getattr(ds, m + "_feat")) for m in ("v", "a", "t")}
    om = eo.OracleEliMRec(U, I, recdim, L, adj, feats, init, cfg["alpha"], dataset_name="synthetic")
    hB = h * W
    ol = om.bpr_loss(u[:hB], p[:hB], n[:hB])
    ol.backward()
    assert abs(float(torch.stack(losses).mean()) - float(ol.detach())) < 1e-5
    want = om.grads()
    gE = torch.cat([e.grad.dense() for e in engines], dim=1).cpu()
    assert_grad_close(gE[:U], want["embedding_user.weight"], "embedding_user.weight")
    assert_grad_close(gE[U:], want["embedding_item.weight"], "embedding_item.weight")
    wg = torch.stack(wgs).sum(0)                      # the all_reduce of the projection-weight gradients
    ws0 = engines[0].model._ws
    for k, (off, numel) in ws0["param_off"].items():
        if k.startswith(("embedding_user.", "embedding_item.")) or k not in want:
            continue
        mine = wg[off - ws0["tail_off"]:off - ws0["tail_off"] + numel].view_as(want[k]).cpu()
        assert_grad_close(mine, want[k], k)


def test_wide_form_native_step_program_equals_python_issued_steps(monkeypatch):
    """The wide form's step (wide_rows, the plain merge, L adjoint hops, wide_grad) as a native program: 30 steps on changing
    batches, bit for bit the steps issued from Python (concat fusion; the `ablate` fixture's mean fusion keeps the ordinary
    path, its head runs torch ops)."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, EliMRec, FusedAdam, SyntheticDataset, set_seed
    U, I, E, dims, B = 700, 1900, 9000, (24, 8, 12), 256
    argv = ["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0",
            "--adj_type=norm", "--layer_num=3", "--propagation=folded"]
    ds = SyntheticDataset(U, I, E, feat_dims=dims, seed=1)
    gen = torch.Generator().manual_seed(5)
    batches = [(torch.randint(0, U, (B,), generator=gen).to(DEV), torch.randint(0, I, (B,), generator=gen).to(DEV),
                torch.randint(0, I, (B,), generator=gen).to(DEV)) for _ in range(30)]
    out = {}
    for native in ("0", "1"):
        monkeypatch.setenv("ELIMREC_NATIVE_STEP", native)
        cfg = make_config(argv)
        set_seed(7)
        model = EliMRec(cfg, ds).to(DEV)
        opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        eng = ColumnShardEngine(model)
        tr = ColumnShardTrainer(eng, opt)
        assert eng.wide
        losses = torch.stack([tr.step(*b) for b in batches]).cpu().numpy()
        st = tr._native_state()
        assert st["failed"] is None, st["failed"]
        assert (st["native_steps"] >= 20) == (native == "1"), st
        eng.sync_to_model()
        out[native] = (losses, {k: v.detach().clone() for k, v in model.state_dict().items()})
    assert np.array_equal(out["0"][0], out["1"][0])
    for k, v in out["0"][1].items():
        assert torch.equal(out["1"][1][k], v), k


def test_native_program_periodic_self_check_and_hyper_parameter_changes(monkeypatch):
    """ELIMREC_PROGRAM_CHECK=N: every N-th step of a native program is issued the ordinary way under a tracer and compared with
    the program call by call -- same bits as an unchecked run, checks counted, nothing flagged. A learning rate changed
    mid-run (a scheduler) is NOT frozen into the program: the program is keyed by the optimizer's hyper-parameters, and the
    run equals one that issues every step from Python."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    g = load_golden("ml3")
    # (three copies of one batch in turn: a program is built from two steps whose index tensors are different objects)
    batches = [tuple(_t(g["step1/%s" % k]) for k in ("users", "pos", "neg")) for _ in range(3)]
    res = {}
    for mode in ("native", "checked", "python"):
        monkeypatch.setenv("ELIMREC_PROGRAM_CHECK", "3" if mode == "checked" else "0")
        monkeypatch.setenv("ELIMREC_NATIVE_STEP", "0" if mode == "python" else "1")
        model, _ = build_model_from_fixture(g, DEV)
        opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
        losses = []
        for k in range(24):
            if k == 14:
                opt.param_groups[0]["lr"] = 0.5 * float(g["lr"])
            losses.append(tr.step(*batches[k % 3]))
        st = tr._native_state()
        res[mode] = ([float(x) for x in losses], {k: v.cpu().clone() for k, v in model.state_dict().items()})
        if mode == "python":
            assert st["native_steps"] == 0
        else:
            assert st["native_steps"] > 8 and st["failed"] is None, st["failed"]
            assert (st["checks"] >= 2) == (mode == "checked")
    for mode in ("checked", "python"):
        assert res[mode][0] == res["native"][0], mode
        for k, v in res["native"][1].items():
            assert torch.equal(res[mode][1][k], v), (mode, k)


def test_large_batch_step_paths_vs_oracle_and_each_other(monkeypatch):
    """B = 4096 (12 288 slots: beyond the one-workgroup planner): the device-wide bitmap planner, the adjoint's source bits behind the
    forward's join, and -- for batches announced with prestage() -- the planner of step t + 1 under step t's backward on the
    alternating buffer sets. The first step's loss against the oracle (1e-5); eight steps -- losses, parameters, Adam moments --
    bitwise equal with and without prestage and with the radix-sort planner (ELIMREC_PLAN_SORTED=1)."""
    from oracle import elimrec_oracle as eo
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, EliMRec, FusedAdam, SyntheticDataset, set_seed
    cfg = make_config(["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0"])
    U, I, B = 2000, 6000, 4096
    ds = SyntheticDataset(U, I, 60000, feat_dims=(16, 8, 12), seed=2)
    gen = torch.Generator().manual_seed(1)
    batches = [(torch.randint(0, U, (B,), generator=gen).to(DEV), torch.randint(0, I, (B,), generator=gen).to(DEV),
                torch.randint(0, I, (B,), generator=gen).to(DEV)) for _ in range(8)]
    got = {}
    for form in ("prestaged", "plain", "sorted"):
        monkeypatch.setenv("ELIMREC_PLAN_SORTED", "1" if form == "sorted" else "0")
        set_seed(5)
        model = EliMRec(cfg, ds)
        init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
        model = model.to(DEV)
        opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        eng = ColumnShardEngine(model)
        tr = ColumnShardTrainer(eng, opt)
        if form == "prestaged":
            tr.prestage(batches)
        losses = [float(tr.step(*b)) for b in batches]
        eng.sync_to_model()
        st = eng.optimizer_state()
        got[form] = (losses, {k: v.clone() for k, v in model.state_dict().items()}, st["exp_avg"].clone(), st["exp_avg_sq"].clone())
        if form == "prestaged":
            tu, ti = ds.get_train_interactions()
            adj = eo.build_adj(tu, ti, U, I, cfg["adj_type"])
            feats = {m: eo.OracleEliMRec.normalize_features(getattr(ds, m + "_feat")) for m in ("v", "a", "t")}
            om = eo.OracleEliMRec(U, I, 64, cfg["layer_num"], adj, feats, init, cfg["alpha"], dataset_name="synthetic")
            ol = om.bpr_loss(*(t.cpu() for t in batches[0]))
            assert abs(losses[0] - float(ol.detach())) < 1e-5
    a = got["prestaged"]
    for other in ("plain", "sorted"):
        b = got[other]
        assert a[0] == b[0], other
        assert all(torch.equal(a[1][k], b[1][k]) for k in a[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), other
