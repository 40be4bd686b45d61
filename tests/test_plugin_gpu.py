"""The reference's own loop body (main.py:98-101: bpr_loss -> zero_grad -> backward(retain_graph=True) -> opt.step) on the
column-shard engine (elimrec_amd/plugin.py): bitwise the engine's own step, whatever the caller reads in between. `-m gpu`."""
import numpy as np
import pytest
import torch

from helpers import assert_grad_close, build_model_from_fixture, feats_of, load_golden, sub

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _batches(g, n):
    """n batches of the fixture: its equal-sized ones in turn, its short last one once in the middle (an epoch's tail)."""
    steps = int(g["steps"])
    order = [1 + k % (steps - 1) for k in range(n)]
    if n >= steps:
        order[n // 2] = steps
    return [tuple(_t(g["step%d/%s" % (t, key)]) for key in ("users", "pos", "neg")) for t in order]


def _opt(model, g, cls=None):
    from elimrec_amd import FusedAdam
    return (cls or FusedAdam)(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))


def _state(model, eng):
    out = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    st = eng.optimizer_state()
    out["@m1"], out["@m2"] = st["exp_avg"], st["exp_avg_sq"]
    return out


def _trainer_run(g, n, extra=()):
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer
    model, _ = build_model_from_fixture(g, DEV, extra_argv=extra)
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, _opt(model, g))
    losses = [tr.step(*b) for b in _batches(g, n)]
    losses = [float(x) for x in torch.stack(losses).cpu()]
    return losses, _state(model, eng), tr


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("name", ["ml3", "kwai", "gcmc", "normal"])
def test_plugin_loop_equals_trainer_step_bitwise(name):
    """main.py:98-101 as written, 12 steps (the native one-call program takes over at the sixth): every loss, every parameter,
    both Adam moments bit for bit those of ColumnShardTrainer.step -- and every step went down the one-enqueue path."""
    g = load_golden(name)
    n = 17
    want_losses, want, tr = _trainer_run(g, n)
    assert tr._native_state()["native_steps"] > 0, tr._native_state()["failed"]
    model, _ = build_model_from_fixture(g, DEV)
    opt = _opt(model, g)
    losses = []
    for u, p, neg in _batches(g, n):
        loss = model.bpr_loss(u, p, neg)
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        losses.append(loss)
    ctl = model.plugin
    assert ctl.fast_steps == n and ctl.slow_steps == 0
    assert ctl.trainer._native_state()["native_steps"] > 0, ctl.trainer._native_state()["failed"]
    assert [loss.cpu().item() for loss in losses] == want_losses
    _same(_state(model, ctl.engine), want)


@pytest.mark.parametrize("name", ["ml3", "kwai"])
def test_plugin_loop_with_line_102_reads_the_published_loss(name):
    """main.py:98-102 verbatim -- `loss.cpu().item()` after EVERY step. From the second step on the launch that sums the loss
    publishes it into coherent host memory and the read waits for that launch alone: every step still goes down the one-enqueue
    path, and every loss, parameter and Adam moment has the bits of ColumnShardTrainer.step (the published word is the loss
    tensor's value; the step's arithmetic does not change with the launch that sums the loss rows)."""
    g = load_golden(name)
    n = 21
    want_losses, want, _ = _trainer_run(g, n)
    model, _ = build_model_from_fixture(g, DEV)
    opt = _opt(model, g)
    got, held = [], []
    for k, (u, p, neg) in enumerate(_batches(g, n)):
        loss = model.bpr_loss(u, p, neg)
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        got.append(loss.cpu().item() if k % 2 == 0 else loss.item())          # main.py:102, both spellings
        held.append(loss)
    ctl = model.plugin
    assert got == want_losses
    assert ctl.fast_steps == n and ctl.slow_steps == 0
    fused = ctl.engine._fused_head_ok()                                       # (the fused head's BPR launch is the one that publishes)
    assert ctl.published_steps == (n - 1 if fused else 0)                     # every step after the first noticed read
    assert [float(x) for x in torch.stack([h.detach() for h in held[-8:]]).cpu()] == want_losses[-8:]    # the device tensors too
    _same(_state(model, ctl.engine), want)
    # a caller that stops reading: back to the late sum, same bits
    for u, p, neg in _batches(g, 3):
        loss = model.bpr_loss(u, p, neg)
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
    assert ctl.published_steps == (n if fused else 0)                         # (the step after the last read still published)
    if not fused:
        return
    # a value asked for after the host ring (64 launches) has wrapped past it comes from the device tensor: the same bits
    from elimrec_amd import ops
    model2, _ = build_model_from_fixture(g, DEV)
    opt2 = _opt(model2, g)
    held2 = []
    for k, (u, p, neg) in enumerate(_batches(g, n)):
        loss = model2.bpr_loss(u, p, neg)
        if k == 1:
            model2.plugin.engine._loss_pub = ops.LossPublisher(2)             # a two-word ring
        opt2.zero_grad()
        loss.backward(retain_graph=True)
        opt2.step()
        assert loss.item() == want_losses[k]
        held2.append(loss)
    assert model2.plugin.engine.loss_publisher().wait(held2[3].__dict__["_elimrec_pub"]) is None      # wrapped
    assert [h.item() for h in held2] == want_losses


def test_plugin_loop_with_per_step_loss_item_and_reads_in_between():
    """The reference's line 102 (`loss.cpu().item()` every step) and every other look at an intermediate result -- the loss
    before backward, .grad after it, the cached tables, predict() -- leave the bits alone: the halves run launch by launch."""
    g = load_golden("ml3")
    n = 9
    want_losses, want, _ = _trainer_run(g, n)
    model, _ = build_model_from_fixture(g, DEV)
    opt = _opt(model, g)
    got = []
    ref = sub(g, "grad1")
    for k, (u, p, neg) in enumerate(_batches(g, n)):
        loss = model.bpr_loss(u, p, neg)
        if k % 3 == 1:
            assert loss.item() == want_losses[k]          # read BEFORE backward: forward half runs
        opt.zero_grad()
        loss.backward(retain_graph=True)
        if k == 0:
            mine = {name: prm.grad for name, prm in model.named_parameters() if prm.grad is not None}
            assert set(mine) == set(ref)
            for name, gr in ref.items():
                assert_grad_close(mine[name].cpu(), gr, name)
        if k == 4:
            assert model.all_users.shape == (model.num_users, model.latent_dim)
            model.predict(g["eval_users"].tolist()[:4])
        opt.step()
        got.append(loss.cpu().item())                                 # main.py:102
    assert got == want_losses
    ctl = model.plugin
    assert ctl.slow_steps >= 4 and ctl.fast_steps >= 4
    _same(_state(model, ctl.engine), want)


def test_plugin_loop_reference_fixture_and_stale_predict():
    """The golden run through the loop body: losses, parameters after the steps, predict() on the stale tables."""
    g = load_golden("ml3")
    model, _ = build_model_from_fixture(g, DEV)
    opt = _opt(model, g)
    steps = int(g["steps"])
    for t in range(1, steps + 1):
        loss = model.bpr_loss(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        assert abs(loss.cpu().item() - float(g["step%d/loss" % t])) < 1e-5
    sd = model.state_dict()
    for k, v in sub(g, "after%d" % steps).items():
        assert np.abs(sd[k].cpu().numpy() - v).max() < 2e-5, k
    model.fusion_mode, model.predict_type = "rubi", "TIE"
    got = model.predict(g["eval_users"].tolist()).numpy()
    assert np.abs(got - g["predict/rubi/TIE"]).max() < 1e-5


def test_plugin_loop_with_torch_adam():
    """optim.Adam(model.parameters()) as the reference constructs it (main.py:49): gradients are materialised for it, the
    embedding tables it updates are re-loaded into the engine -- the golden parameters after three steps."""
    g = load_golden("ml3")
    model, _ = build_model_from_fixture(g, DEV)
    opt = _opt(model, g, torch.optim.Adam)
    steps = int(g["steps"])
    for t in range(1, steps + 1):
        loss = model.bpr_loss(*(_t(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg")))
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        assert abs(loss.cpu().item() - float(g["step%d/loss" % t])) < 1e-5, t
    sd = model.state_dict()
    for k, v in sub(g, "after%d" % steps).items():
        assert np.abs(sd[k].cpu().numpy() - v).max() < 2e-5, k


def test_plugin_backward_with_gradient_and_edited_grads():
    """backward(gradient=2) goes through autograd and doubles every gradient; gradients edited before the step are the ones
    the update uses (here: zeroed -> only weight decay moves the parameters, as with torch)."""
    g = load_golden("ml3")
    model, _ = build_model_from_fixture(g, DEV)
    opt = _opt(model, g)
    u, p, neg = _batches(g, 1)[0]
    loss = model.bpr_loss(u, p, neg)
    opt.zero_grad()
    loss.backward(gradient=torch.tensor(2.0, device=DEV))
    ref = sub(g, "grad1")
    for name, prm in model.named_parameters():
        if name in ref:
            assert_grad_close(prm.grad.cpu(), 2.0 * ref[name], name)
    opt.zero_grad()
    model2, _ = build_model_from_fixture(g, DEV)
    opt2 = _opt(model2, g)
    loss = model2.bpr_loss(u, p, neg)
    loss.backward()
    for prm in model2.parameters():
        if prm.grad is not None:
            prm.grad.zero_()
    before = {k: v.cpu().clone() for k, v in model2.state_dict().items()}
    opt2.step()
    after = model2.state_dict()
    for k in ref:
        twin = before[k].clone().requires_grad_(True)
        twin.grad = torch.zeros_like(twin)
        torch.optim.Adam([twin], lr=float(g["lr"]), weight_decay=float(g["weight_decay"])).step()
        assert (after[k].cpu() - twin.detach()).abs().max().item() < 2e-7, k


def test_plugin_stale_backward_raises_and_second_loss_settles_the_first():
    g = load_golden("ml3")
    model, _ = build_model_from_fixture(g, DEV)
    opt = _opt(model, g)
    b = _batches(g, 2)
    first = model.bpr_loss(*b[0])
    second = model.bpr_loss(*b[1])                  # the first loss becomes real here (its tensor is still held)
    assert abs(first.item() - float(g["step1/loss"])) < 1e-5
    with pytest.raises(RuntimeError, match="stale"):
        first.backward()
    second.backward()
    opt.step()
    assert np.isfinite(second.item())


def test_plugin_two_backwards_before_one_step_accumulate_like_torch():
    """bpr_loss(b1).backward(); bpr_loss(b2).backward(); opt.step(): torch semantics -- .grad holds g1 + g2 and the update
    consumes the sum (the fused step used to run on b2 alone and drop g1). Checked against a twin model that
    materialises both gradients separately and steps torch's own Adam on their sum; a regulariser's gradient written by
    autograd's AccumulateGrad (never through the Python .grad setter) is zeroed by zero_grad() and not carried over."""
    g = load_golden("ml3")
    b = _batches(g, 2)
    # the twin: each batch's gradients alone (read through .grad), summed by hand
    twin, _ = build_model_from_fixture(g, DEV)
    topt = _opt(twin, g, torch.optim.Adam)
    parts = []
    for u, p, neg in b:
        topt.zero_grad()
        loss = twin.bpr_loss(u, p, neg)
        loss.backward()
        parts.append({k: prm.grad.detach().clone() for k, prm in twin.named_parameters() if prm.grad is not None})
    topt.zero_grad()
    for k, prm in twin.named_parameters():
        if k in parts[0]:
            prm.grad = parts[0][k] + parts[1][k]
    topt.step()
    want = {k: v.detach().cpu() for k, v in twin.state_dict().items()}
    model, _ = build_model_from_fixture(g, DEV)
    opt = _opt(model, g)
    opt.zero_grad()
    model.bpr_loss(*b[0]).backward()
    model.bpr_loss(*b[1]).backward()                     # settles the first: g1 lands in .grad
    opt.step()
    got = model.state_dict()
    for k in want:
        assert (got[k].cpu() - want[k]).abs().max().item() < 3e-7, k
    # AccumulateGrad behind the setter's back
    opt.zero_grad()
    assert not model.plugin.grads_set and not model.plugin.any_raw_grad()
    reg = (model.embedding_user_after_GCN.weight ** 2).sum()
    reg.backward()
    assert not model.plugin.grads_set and model.plugin.any_raw_grad()
    opt.zero_grad()
    assert not model.plugin.any_raw_grad()


def test_loss_read_without_a_step_then_prestaged_steps_stay_bitwise():
    """A pass that does not flip the batch-buffer sets (a loss read and nothing else: forward_only) followed by steps on
    PRESTAGED batches: the next planner must not run ahead into the set the unfinished pass still reads. Every loss
    and the final state equal a run without the extra read, bit for bit, over many alternations."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer
    g = load_golden("kwai")
    n = 24
    bs = _batches(g, n)

    def run(with_reads):
        model, _ = build_model_from_fixture(g, DEV)
        eng = ColumnShardEngine(model)
        tr = ColumnShardTrainer(eng, _opt(model, g))
        tr.prestage(bs)
        losses = []
        for k, b in enumerate(bs):
            if with_reads and k % 2 == 1:
                ctx = tr.forward_only(*bs[(k + 3) % n])              # another batch's forward: planned, never stepped
                assert ctx is not None and not eng.ahead_safe()
            losses.append(tr.step(*b))
        return [float(x) for x in torch.stack(losses).cpu()], _state(model, eng)
    base_l, base = run(False)
    read_l, read = run(True)
    assert read_l == base_l
    _same(read, base)


@pytest.mark.parametrize("extra", [["--lean_tables=1"], ["--feature_dtype=bf16"]])
def test_plugin_loop_under_lean_tables_and_16bit_features(extra):
    g = load_golden("ml3")
    want_losses, want, _ = _trainer_run(g, 4, extra=extra)
    model, _ = build_model_from_fixture(g, DEV, extra_argv=extra)
    opt = _opt(model, g)
    got = []
    for u, p, neg in _batches(g, 4):
        loss = model.bpr_loss(u, p, neg)
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        got.append(loss)
    assert [x.item() for x in got] == want_losses
    _same(_state(model, model.plugin.engine), want)


def test_plugin_loop_on_an_adjacency_with_a_diagonal():
    """--propagation=folded on adj_type=norm (the wide form): no raise, the golden losses."""
    g = load_golden("ablate")
    extra = ["--propagation=folded"]
    model, _ = build_model_from_fixture(g, DEV, extra_argv=extra)
    opt = _opt(model, g)
    got = []
    batches = [tuple(_t(g["step%d/%s" % (t, key)]) for key in ("users", "pos", "neg")) for t in (1, 2, 3)]
    for u, p, neg in batches:
        loss = model.bpr_loss(u, p, neg)
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        got.append(loss.cpu().item())
    for t in range(3):
        assert abs(got[t] - float(g["step%d/loss" % (t + 1)])) < 1e-5
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer
    twin, _ = build_model_from_fixture(g, DEV, extra_argv=extra)
    eng = ColumnShardEngine(twin)
    tr = ColumnShardTrainer(eng, _opt(twin, g))
    assert [tr.step(*b).item() for b in batches] == got
    _same(_state(model, model.plugin.engine), _state(twin, eng))


@pytest.mark.parametrize("name", ["ml3", "kwai"])
def test_plugin_loop_on_tiny_and_degenerate_batches_vs_oracle(name):
    """The reference's loop body on batches the fixtures do not hold: one triplet, two, five, a batch whose triplets are all the
    same, a batch whose positive equals its negative (loss = log 2 exactly in the cosine form) -- loss against the oracle's
    (1e-5), every gradient of the first such step (1e-4 row-wise), and the loop keeps running through all of them."""
    from oracle import elimrec_oracle as eo
    from elimrec_amd import FusedAdam
    g = load_golden(name)
    model, cfg = build_model_from_fixture(g, DEV)
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    adj = eo.build_adj(g["train_u"], g["train_i"], int(g["num_users"]), int(g["num_items"]), str(g["adj_type"]))
    om = eo.OracleEliMRec(int(g["num_users"]), int(g["num_items"]), int(g["recdim"]), int(g["layer_num"]), adj, feats_of(g),
                          sub(g, "init"), float(g["alpha"]), dataset_name=str(g["dataset_name"]),
                          modality=str(g["modality"]), mm_fusion_mode=str(g["mm_fusion_mode"]))
    U, I = int(g["num_users"]), int(g["num_items"])
    gen = torch.Generator().manual_seed(9)
    cases = []
    for B in (1, 2, 5):
        cases.append((torch.randint(0, U, (B,), generator=gen), torch.randint(0, I, (B,), generator=gen), torch.randint(0, I, (B,), generator=gen)))
    cases.append((torch.full((7,), 3), torch.full((7,), 5), torch.full((7,), 11)))                 # seven times the same triplet
    same = torch.randint(0, I, (4,), generator=gen)
    cases.append((torch.randint(0, U, (4,), generator=gen), same, same.clone()))                   # positive == negative
    # the first case: loss and gradients of the untouched initial parameters against the oracle
    u, p, n = cases[0]
    want = om.bpr_loss(u, p, n)
    want.backward()
    loss = model.bpr_loss(u.to(DEV), p.to(DEV), n.to(DEV))
    opt.zero_grad()
    loss.backward(retain_graph=True)
    assert abs(float(loss) - float(want.detach())) < 1e-5
    ref = om.grads()
    mine = {k: q.grad for k, q in model.named_parameters() if q.grad is not None}
    assert set(mine) == set(ref)
    for k, gr in ref.items():
        assert_grad_close(mine[k].cpu(), gr, k)
    opt.step()
    # the others on the evolving parameters: finite losses, the loop does not stall; the degenerate one has the closed-form loss
    for u, p, n in cases[1:]:
        loss = model.bpr_loss(u.to(DEV), p.to(DEV), n.to(DEV))
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        assert np.isfinite(float(loss))
    # (pos == neg: every cosine difference is 0, so each of the 1 + alpha * S terms is log 2)
    om2_terms = float(loss) / np.log(2.0)
    assert abs(om2_terms - round(om2_terms)) < 1e-4 or abs(om2_terms - (1 + float(g["alpha"]) * model.S)) < 1e-4, om2_terms
    model.state_dict()
    for q in model.parameters():
        assert torch.isfinite(q.detach()).all()
