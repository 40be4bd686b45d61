"""The rest of the reference's model API on the GPU: forward() / getEmbedding() against the golden tables, the
base-class losses (models/BasicModel.py:59-113) differentiated through the HIP table build against the oracle's
autograd, util/mlp.py's MLP against torch, torch.ops registration. Needs a GPU: `-m gpu`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import build_model_from_fixture, feats_of, load_golden, rel_err, sub

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_forward_and_get_embedding_match_reference(fixture_name):
    """models/EliMRec.py:274-307 against the tables the reference computed in its first forward (fwd1/*)."""
    g = load_golden(fixture_name)
    model, _ = build_model_from_fixture(g, DEV)
    au, ai = g["fwd1/all_users"], g["fwd1/all_items"]
    u, p, n = (g["step1/%s" % k] for k in ("users", "pos", "neg"))
    got = model.forward(_t(u), _t(p))
    assert not got.requires_grad
    want = (au[u] * ai[p]).sum(1)
    assert np.abs(got.cpu().numpy() - want).max() < 1e-5 * max(1.0, np.abs(want).max())
    with torch.no_grad():
        ue, pe, ne, u0, p0, n0 = model.getEmbedding(_t(u), _t(p), _t(n))
    for mine, ref in ((ue, au[u]), (pe, ai[p]), (ne, ai[n]), (u0, g["init/embedding_user.weight"][u]),
                      (p0, g["init/embedding_item.weight"][p]), (n0, g["init/embedding_item.weight"][n])):
        assert rel_err(mine.cpu(), ref) < 1e-4
    out = model.getEmbedding(_t(u), _t(p), None)
    assert out[2] is None and out[5] is None


def _oracle(g):
    from oracle import elimrec_oracle as eo
    adj = eo.build_adj(g["train_u"], g["train_i"], int(g["num_users"]), int(g["num_items"]), str(g["adj_type"]))
    return eo.OracleEliMRec(int(g["num_users"]), int(g["num_items"]), int(g["recdim"]), int(g["layer_num"]), adj, feats_of(g),
                            sub(g, "init"), float(g["alpha"]), dataset_name=str(g["dataset_name"]),
                            modality=str(g["modality"]), mm_fusion_mode=str(g["mm_fusion_mode"]))


@pytest.mark.parametrize("loss_name", ["bpr_loss", "infonce", "fast_loss"])
@pytest.mark.parametrize("name", ["ml3", "ablate"])
def test_base_class_losses_through_the_table_autograd_bridge(name, loss_name):
    """BasicModel's generic losses on EliMRec's tables: value and every parameter gradient against the same formulas
    differentiated by torch through the oracle's compute() (1e-4 rel)."""
    from elimrec_amd import BasicModel
    g = load_golden(name)
    model, cfg = build_model_from_fixture(g, DEV)
    om = _oracle(g)
    u, p, n = (torch.from_numpy(g["step1/%s" % k]).long() for k in ("users", "pos", "neg"))
    loss = getattr(BasicModel, loss_name)(model, u.to(DEV), p.to(DEV), n.to(DEV))
    loss.backward()
    au, ai = om.compute()
    if loss_name == "bpr_loss":
        want = torch.mean(F.softplus((au[u] * ai[n]).sum(1) - (au[u] * ai[p]).sum(1)))
    elif loss_name == "infonce":
        logits = torch.mm(F.normalize(au[u], dim=1), F.normalize(ai[p], dim=1).T) / cfg["temp"]
        want = F.cross_entropy(logits, torch.arange(len(u)))
    else:
        ue, pe = F.normalize(au[u], dim=1), F.normalize(ai[p], dim=1)
        nu, ni = F.normalize(au, dim=1), F.normalize(ai, dim=1)
        ps = (ue * pe).sum(1)
        a = float(g["alpha"])
        want = torch.sum((a - 1) * ps ** 2 - 2 * a * ps) + torch.trace((nu.T @ nu) @ (ni.T @ ni))
    want.backward()
    assert abs(float(loss.detach()) - float(want.detach())) < 1e-4 * max(1.0, abs(float(want.detach())))
    ref = om.grads()
    mine = {k: q.grad for k, q in model.named_parameters() if q.grad is not None}
    scale = max(float(v.abs().max()) for v in ref.values())
    for k, v in ref.items():
        if float(v.abs().max()) < 1e-6 * scale:      # mathematically zero (e.g. the item bias under pos - neg): round-off only
            assert k not in mine or float(mine[k].abs().max()) < 1e-6 * scale, k
            continue
        assert k in mine, k
        assert rel_err(mine[k].cpu(), v) < 1e-4, k


@pytest.mark.parametrize("dims,act", [((128, [64, 64]), "relu"), ((100, [36, 10, 6]), "relu"), ((24, [16]), "relu"),
                                      ((32, [64, 8]), "tanh")])
def test_mlp_matches_torch(dims, act):
    """util/mlp.py:6-38: same parameters, same input -> same output and gradients as the torch module (1e-5 / 1e-4)."""
    from elimrec_amd import MLP
    torch.manual_seed(0)
    mine = MLP(dims[0], dims[1], activation=act).to(DEV)
    mine.init_weight("xavier")
    x = torch.randn(37, dims[0], device=DEV, requires_grad=True)
    y = mine(x)
    xr = x.detach().clone().requires_grad_(True)
    h = xr
    for i, lin in enumerate(mine.linears):
        h = F.linear(h, lin.weight.detach().clone().requires_grad_(False), lin.bias.detach())
        if i < len(mine.linears) - 1:
            h = F.__dict__[act](h)
    assert (y - h).abs().max().item() < 1e-5
    w = torch.randn_like(y)
    (y * w).sum().backward()
    # torch reference with its own parameter copies
    ref = torch.nn.ModuleList([torch.nn.Linear(l.in_features, l.out_features) for l in mine.linears]).to(DEV)
    for a, b in zip(ref, mine.linears):
        a.load_state_dict(b.state_dict())
    h = xr
    for i, lin in enumerate(ref):
        h = lin(h)
        if i < len(ref) - 1:
            h = F.__dict__[act](h)
    (h * w).sum().backward()
    assert rel_err(x.grad.cpu(), xr.grad.cpu()) < 1e-4
    for a, b in zip(ref, mine.linears):
        assert rel_err(b.weight.grad.cpu(), a.weight.grad.cpu()) < 1e-4
        assert rel_err(b.bias.grad.cpu(), a.bias.grad.cpu()) < 1e-4
    assert len(list(mine.parameters())) == 2 * len(dims[1])


def _net(tmp_path, extra, num_epoch=4):
    import importlib
    import os
    import sys
    from helpers import ROOT
    sys.path.insert(0, ROOT)
    main = importlib.import_module("main")
    from elimrec_amd import Configurator, set_seed
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        args = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                            argv=["main.py", "--data.input.dataset=synthetic", "--alpha=0.5", "--synthetic_shape=[300,500,6000]",
                                  "--synthetic_dims=[16,8,12]", "--recdim=32", "--batch_size=512", "--num_epoch=%d" % num_epoch,
                                  "--test_step=1", "--verbose=0", "--path=%s" % str(tmp_path / "ck")] + list(extra))
        set_seed(args["seed"])
        return main.Net(args), main
    finally:
        os.chdir(cwd)


def test_resume_restores_parameters_and_optimizer_state(tmp_path):
    """N4: train 2 epochs, save; a fresh process-equivalent Net with --resume continues with the SAME next step as
    the uninterrupted run (parameters, Adam moments of the projection weights and of the column-sharded embeddings,
    step counts, epoch counter): bitwise equal parameters after one more identical batch."""
    import os
    from helpers import ROOT
    a, _ = _net(tmp_path, ["--loss=bpr_loss"], num_epoch=2)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        a.run()
        path = a.recommender.getFileName()
        a.save_checkpoint(path, 1, dict(best_recall={"TE": 0.0, "TIE": 0.0}, best_epoch={"TE": 0, "TIE": 0},
                                        best_valid_line="", test_lines={"TE": "", "TIE": ""}))
        assert os.path.exists(path) and os.path.exists(path + ".resume")
        b, _ = _net(tmp_path, ["--loss=bpr_loss", "--resume=%s" % path], num_epoch=2)
    finally:
        os.chdir(cwd)
    assert b.start_epoch == 2 and b.engine is not None and b.engine.step_count == a.engine.step_count
    torch.manual_seed(5)
    u = torch.randint(0, 300, (256,), device=DEV)
    p = torch.randint(0, 500, (256,), device=DEV)
    n = torch.randint(0, 500, (256,), device=DEV)
    la, lb = a.trainer.step(u, p, n), b.trainer.step(u, p, n)
    assert float(la) == float(lb)
    a.engine.sync_to_model(); b.engine.sync_to_model()
    sa, sb = a.recommender.state_dict(), b.recommender.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_driver_honours_config_loss(tmp_path):
    """main.py:98 dispatches on config.loss: --loss=infonce trains through the generic path (loss decreases), an
    unknown method name fails at construction."""
    import os
    from helpers import ROOT
    net, _ = _net(tmp_path, ["--loss=infonce"], num_epoch=1)
    assert net.trainer is None
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        from elimrec_amd import PairwiseSamplerV2
        batches = list(PairwiseSamplerV2(net.dataset, batch_size=512, device=DEV, seed=1))
        first = float(net.generic_step(*batches[0]))
        for b in batches[1:6]:
            net.generic_step(*b)
        again = float(net.generic_step(*batches[0]))
        with pytest.raises(AttributeError):
            _net(tmp_path, ["--loss=no_such_loss"])
    finally:
        os.chdir(cwd)
    assert again < first


def test_out_of_range_index_is_flagged_not_dereferenced():
    """a bad user / item id must not read or write out of bounds; check_indices() raises like the reference's
    gather would."""
    g = load_golden("ml3")
    model, _ = build_model_from_fixture(g, DEV)
    u, p, n = (_t(g["step1/%s" % k]).clone() for k in ("users", "pos", "neg"))
    n[3] = int(g["num_items"]) + 5
    u[1] = -2
    loss = model.bpr_loss(u, p, n)
    assert torch.isfinite(loss)
    with pytest.raises(IndexError):
        model.check_indices()
    model.check_indices()          # the flag is cleared


def test_torch_ops_match_the_ctypes_binding():
    """torch.ops.elimrec.* (TORCH_LIBRARY over the C ABI) against the same entry points through ctypes / torch."""
    from elimrec_amd import ops, torch_ops
    import scipy.sparse as sp
    t = torch_ops.load()
    rng = np.random.RandomState(0)
    n, C, L = 800, 64, 3
    m = sp.random(n, n, density=0.01, random_state=rng, format="csr", dtype=np.float32)
    m.sort_indices()
    X = torch.randn(n, C, device=DEV)
    rp, col, val = (_t(m.indptr.astype(np.int32)), _t(m.indices.astype(np.int32)), _t(m.data.astype(np.float32)))
    got = t.propagate(rp, col, val, X, L)
    A = torch.from_numpy(m.toarray()).to(DEV).double()
    xs = [X.double()]
    for _ in range(L):
        xs.append(A @ xs[-1])
    assert (got.double() - torch.stack(xs).mean(0)).abs().max().item() < 1e-5
    # Linear forward / weight gradient
    a, w, b = torch.randn(300, 128, device=DEV), torch.randn(64, 128, device=DEV), torch.randn(64, device=DEV)
    assert (t.linear_fwd(a, w, b) - F.linear(a, w, b)).abs().max().item() < 1e-4
    gw, gb = t.linear_bwd_w(a, torch.randn(300, 32, device=DEV).mul_(0).add_(1.0))
    assert (gb - a.sum(0)).abs().max().item() < 1e-3 and gw.shape == (128, 32)
    # Adam in place == the ctypes call
    p, g, mm, vv = (torch.randn(1000, device=DEV) for _ in range(4))
    vv.abs_()
    p2, m2, v2 = p.clone(), mm.clone(), vv.clone()
    out = t.adam_step_(p, g, mm, vv, 1e-3, 0.9, 0.999, 1e-8, 1e-4, 2)
    ops.adam_step(p2, g, m2, v2, 1e-3, 0.9, 0.999, 1e-8, 1e-4, 2)
    assert out.data_ptr() == p.data_ptr() and torch.equal(p, p2) and torch.equal(mm, m2) and torch.equal(vv, v2)
    # scoring + top-K + metrics on a fixture
    g_ = load_golden("ml3")
    model, _ = build_model_from_fixture(g_, DEV)
    model.compute()
    users = torch.arange(20, device=DEV)
    model.predict_type = "TIE"
    idx_ref, val_ref = model.predict_device(users, top_k=10)
    idx, vals = t.score_topk(model._ws["Y"], model.num_users, model.num_items, users, model.latent_dim, model.S, model._head_mask(),
                             0, 2, None, None, 10)
    assert torch.equal(idx, idx_ref) and torch.equal(vals, val_ref)
    tp = torch.arange(0, 21, device=DEV, dtype=torch.int64)
    ti = torch.arange(20, device=DEV, dtype=torch.int32)
    want = torch.empty(20, 30, device=DEV)
    ops.rank_metrics(idx_ref, tp, ti, [1, 2, 4], want)
    assert torch.equal(t.rank_metrics(idx, tp, ti, [1, 2, 4]), want)
    # loss rows of the BPR head
    u, pp, nn_ = (_t(g_["step1/%s" % k]) for k in ("users", "pos", "neg"))
    bw = model._block_weights()
    loss_rows, grad_rows, keys = t.bpr_head_fwd(model._ws["Y"], model.num_users, model.num_items, u, pp, nn_, model.latent_dim, bw)
    assert keys.numel() == 3 * u.numel() and grad_rows.shape == (3 * u.numel(), model.Cy)
    assert abs(float(loss_rows.sum()) - float(g_["step1/loss"])) < 1e-5
    with pytest.raises(RuntimeError):
        t.propagate(rp[:-3], col, val, X, L)
    # transpose flag: the adjoint of a matrix that is not self-adjoint (SparseAddmmBackward multiplies by A^T)
    got_t = t.propagate(rp, col, val, X, L, True)
    xs = [X.double()]
    for _ in range(L):
        xs.append(A.t() @ xs[-1])
    assert (got_t.double() - torch.stack(xs).mean(0)).abs().max().item() < 1e-5
    # segment_reduce (= the row-gradient reduction behind bpr_head_fwd: "bpr_head_bwd") == index_add in key order
    active, reduced, seg = t.segment_reduce(grad_rows, keys, model.num_users)
    n_act = int(seg[0])
    uniq = torch.unique(keys.long())
    assert n_act == len(uniq) and torch.equal(active[:n_act].long(), uniq) and int(seg[1]) == int((uniq < model.num_users).sum())
    want = torch.zeros(model.num_users + model.num_items, model.Cy, device=DEV, dtype=torch.float64).index_add_(0, keys.long(), grad_rows.double())
    assert (reduced[:n_act].double() - want[uniq]).abs().max().item() < 1e-6


def test_torch_ops_of_the_sharded_paths_and_the_head_backward():
    """torch.ops.elimrec for the entry points SURVEY.md 8(b) lists beyond the first nine (missing 5): `bpr_head_bwd`
    (the saved row gradients reduced per node into a dense dY == index_add), the item-sharded scorer (`score_shard_row_sums` +
    `score_topk_shard` per shard + `topk_merge` == `score_topk` over the whole catalogue) and the row-sharded constants' lookup
    (`lookup_counts` / `lookup_pack` / `lookup_unpack` over two owners == indexing the full tables)."""
    from elimrec_amd import torch_ops
    from elimrec_amd.lookup import FeatureShard, RowOwnerMap
    t = torch_ops.load()
    g_ = load_golden("ml3")
    model, _ = build_model_from_fixture(g_, DEV)
    model.compute()
    U, I, d, Y = model.num_users, model.num_items, model.latent_dim, model._ws["Y"]
    u, pp, nn_ = (_t(g_["step1/%s" % k]) for k in ("users", "pos", "neg"))
    loss_rows, grad_rows, keys = t.bpr_head_fwd(Y, U, I, u, pp, nn_, d, model._block_weights())
    gout = torch.full((), 0.5, device=DEV)
    dY = t.bpr_head_bwd(grad_rows, keys, gout, U + I)
    want = torch.zeros(U + I, model.Cy, device=DEV, dtype=torch.float64).index_add_(0, keys.long(), grad_rows.double() * 0.5)
    assert dY.shape == (U + I, model.Cy) and (dY.double() - want).abs().max().item() < 1e-6
    # item shards: W = 2
    users = torch.arange(24, device=DEV)
    idx_ref, val_ref = t.score_topk(Y, U, I, users, d, model.S, model._head_mask(), 0, 2, None, None, 10)
    cut = [0, I // 3, I]
    shards = [torch.cat([Y[:U], Y[U + cut[r]:U + cut[r + 1]]]).contiguous() for r in range(2)]
    total = sum(t.score_shard_row_sums(shards[r], U, cut[r + 1] - cut[r], users, d, model.S, model._head_mask(), 0, I) for r in range(2))
    parts = [t.score_topk_shard(shards[r], U, cut[r + 1] - cut[r], users, d, model.S, model._head_mask(), 0, 2, None, None, 10, total, I, cut[r])
             for r in range(2)]
    idx, val = t.topk_merge(torch.cat([p[1] for p in parts], 1), torch.cat([p[0] for p in parts], 1), 10)
    assert torch.equal(idx, idx_ref) and (val - val_ref).abs().max().item() < 1e-6
    # row-sharded constants: two owners, requester 0
    torch.manual_seed(0)
    N = U + I
    tabs = [torch.randn(N, 12, device=DEV), torch.randn(N, 20, device=DEV)]
    c = torch.rand(N, device=DEV)
    own = RowOwnerMap(U, I, 2)
    fs = [FeatureShard(own, r, tabs, c) for r in range(2)]
    R = 64
    acts = torch.full((2, R), -(1 << 30), dtype=torch.int32, device=DEV)
    for r in range(2):
        ids = torch.unique(torch.randint(0, N, (R - 9,), generator=torch.Generator().manual_seed(r))).to(DEV).int()
        acts[r, :ids.numel()] = ids
    ub, ib = own.ub.tolist(), own.ib.tolist()
    counts = t.lookup_counts(acts, U, I, ub, ib)
    assert int(counts[0].sum()) == int((acts[0] >= 0).sum())
    chunks = []
    for o in range(2):
        send, off = t.lookup_pack(acts, U, I, ub, ib, o, fs[o].table, fs[o].row_bytes)
        lo, hi = int(off[0]), int(off[1])
        assert hi - lo == int(counts[0, o])
        chunks.append(send[lo:hi])
    S, cc = t.lookup_unpack(acts[0], U, I, ub, ib, 0, torch.cat(chunks).contiguous(), fs[0].row_bytes, 0, 32, False)
    n0 = int((acts[0] >= 0).sum())
    a0 = acts[0, :n0].long()
    assert torch.equal(S[:n0], torch.cat([tabs[0][a0], tabs[1][a0]], 1)) and torch.equal(cc[:n0], c[a0])


@pytest.mark.parametrize("adj_type", ["pre", "plain", "gcmc", "norm", "mean"])
def test_device_adjacency_is_bit_identical_to_scipy(adj_type):
    """N3: csrc/adj.hip against model.create_adj_mat (itself bit-identical to the reference's matrix, test_host_logic):
    same row pointers, same sorted columns, same fp32 values bit for bit -- fixtures and a skewed random graph with
    isolated nodes."""
    from elimrec_amd.adjacency import build_adj_device
    from elimrec_amd.model import create_adj_mat
    cases = []
    g = load_golden("ml3")
    cases.append((g["train_u"], g["train_i"], int(g["num_users"]), int(g["num_items"])))
    rng = np.random.RandomState(3)
    U, I = 3000, 5000
    pop = 1.0 / np.arange(1, I + 1) ** 0.9
    pairs = np.unique(np.stack([rng.randint(0, U - 50, 60000), rng.choice(I - 80, 60000, p=pop[:I - 80] / pop[:I - 80].sum())], 1), axis=0)
    cases.append((pairs[:, 0], pairs[:, 1], U, I))             # the last 50 users / 80 items are isolated
    for tu, ti, nu, ni in cases:
        want = create_adj_mat(tu, ti, nu, ni, adj_type)
        rp, cl, vl = build_adj_device(tu, ti, nu, ni, adj_type, DEV)
        assert np.array_equal(rp.cpu().numpy(), want.indptr)
        assert np.array_equal(cl.cpu().numpy(), want.indices)
        assert np.array_equal(vl.cpu().numpy().view(np.uint32), want.data.astype(np.float32).view(np.uint32))
    with pytest.raises(ValueError):
        build_adj_device([1, 1], [2, 2], 5, 5, "pre", DEV)


def test_model_built_on_the_device_adjacency_trains_like_the_host_one():
    g = load_golden("ml3")
    from helpers import FixtureDataset, fixture_argv, make_config
    from elimrec_amd import EliMRec
    runs = []
    for how in ("host", "device"):
        cfg = make_config(fixture_argv(g) + ["--adj_build=%s" % how])
        model = EliMRec(cfg, FixtureDataset(g))
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sub(g, "init").items()})
        for m in ("v", "a", "t"):
            getattr(model, m + "_feat").copy_(torch.from_numpy(g[m + "_feat"]))
        model = model.to(DEV)
        loss = model.bpr_loss(*(_t(g["step1/%s" % k]) for k in ("users", "pos", "neg")))
        runs.append(float(loss))
    assert runs[0] == runs[1] and abs(runs[0] - float(g["step1/loss"])) < 1e-5
