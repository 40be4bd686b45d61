"""Lean tables (--lean_tables=1, BASELINE.json configs[4]): nothing of N rows on the device outside the column-sharded engine,
and elimrec_amd.capacity.plan() against the device allocator. Needs a real MI355X: `-m gpu`."""
import numpy as np
import pytest
import torch

from helpers import make_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(U, I, E, dims, recdim, lean, seed=7):
    from elimrec_amd import EliMRec, SyntheticDataset, set_seed
    argv = ["--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=%d" % recdim, "--verbose=0",
            "--feature_shard=row"] + (["--lean_tables=1"] if lean else []) + (["--feature_load=block"] if lean == "block" else [])
    cfg = make_config(argv)
    ds = SyntheticDataset(U, I, E, feat_dims=dims, seed=1)
    set_seed(seed)
    return cfg, ds, EliMRec(cfg, ds).to(DEV)


def _batches(ds, B, n, seed=3):
    g = torch.Generator().manual_seed(seed)
    train = ds.train_matrix.tocoo()
    out = []
    for _ in range(n):
        pick = torch.randint(0, train.nnz, (B,), generator=g).numpy()
        out.append((torch.from_numpy(train.row[pick].astype(np.int64)).to(DEV), torch.from_numpy(train.col[pick].astype(np.int64)).to(DEV),
                    torch.randint(0, ds.num_items, (B,), generator=g).to(DEV)))
    return out


@pytest.mark.parametrize("recdim", [64, 32])
def test_lean_tables_train_and_evaluate_like_regular_tables(recdim, monkeypatch):
    """The same seeded model with and without --lean_tables=1 (host-resident embedding parameters and raw features, no
    device CSR copies, no [N x C] workspace tables; constants from the distributed fold in both): three trainer steps give
    the same losses bit for bit, sync_to_model() the same parameters, evaluate() -- item-shard scorer on one rank against the
    whole-table scorer -- the same metrics, predict() the same scores."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    monkeypatch.setenv("ELIMREC_FOLD", "sharded")
    U, I, E, dims, B = 700, 1900, 9000, (24, 8, 12), 257
    res = {}
    for lean in (False, True, "block"):       # "block": lean tables + the raw features read block by block (--feature_load=block)
        cfg, ds, model = _build(U, I, E, dims, recdim, lean)
        assert model._lean == bool(lean)
        if lean == "block":
            assert not torch.is_tensor(model.v_feat) and model.v_feat.shape == (I, dims[0]) and model.v_feat.rows_read == 0
        elif lean:
            assert model.embedding_user.weight.device.type == "cpu" and model.v_feat.device.type == "cpu"
        if lean:
            assert model.s_dense_v.weight.is_cuda and not hasattr(model, "adj_rowptr")
        opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        eng = ColumnShardEngine(model)
        tr = ColumnShardTrainer(eng, opt)
        assert eng.lookup and eng.fold_mode == "sharded"
        losses = [float(tr.step(*b)) for b in _batches(ds, B, 3)]
        model.predict_type = "TIE"
        users = list(range(0, U, 17))[:40]
        pred = model.predict(users).numpy()
        ev, _ = model.evaluate()
        eng.sync_to_model()
        res[lean] = (losses, pred, ev, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        if lean == "block":                   # one rank: its item block is the whole range, read once per modality
            for m_ in "vat":
                assert getattr(model, m_ + "_feat").blocks == [(0, I)]
        if lean:
            ws = model._ws
            assert ws["fold"] is None and ws["Y"] is None and "Out" not in ws and "X0d" not in ws and "layers" not in ws
            loss = model.bpr_loss(*_batches(ds, 8, 1)[0])      # the reference's loop body runs on the engine, lean or not
            loss.backward()
            opt.step()
            assert np.isfinite(loss.item()) and model.plugin.fast_steps == 1
    for lean in (True, "block"):
        assert res[lean][0] == res[False][0]
        for k, v in res[False][3].items():
            assert torch.equal(res[lean][3][k], v), k
        assert np.abs(res[lean][1] - res[False][1]).max() < 2e-7
        assert np.abs(res[lean][2] - res[False][2]).max() < 1e-7


def test_capacity_plan_matches_the_device_allocator_at_the_scaled_c5_shape():
    """BASELINE.json configs[4] scaled to one GPU (|I| = 2 M, |U| = 20 k, 16 M interactions, recdim 256, 3 x 256-d features
    stored fp16, lean tables): what elimrec_amd.capacity.plan() adds up from the allocators' formulas is what the device
    allocator holds -- after set-up + training steps (training_bytes) and with the evaluator's item-sharded tables on top
    (total_bytes) -- to 10 %. The same function sizes the real configs[4] in DESIGN.md."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, capacity
    U, I, E, dims, d, B = 20000, 2000000, 16000000, (256, 256, 256), 256, 2048
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    cfg, ds, model = _build(U, I, E, dims, d, lean=True)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    eng = ColumnShardEngine(model, feature_dtype="f16")
    tr = ColumnShardTrainer(eng, opt)
    losses = [float(tr.step(*b)) for b in _batches(ds, B, 3)]
    assert all(np.isfinite(losses)) and losses[2] < losses[0] + 0.05
    torch.cuda.synchronize()
    held_train = torch.cuda.memory_allocated() - base
    p = capacity.plan(U, I, int(eng.plan.nnz) // 2, d, dims, world=1, layers=model.n_layers, batch=B, symmetric=model._adj_symmetric,
                      feature_dtype="f16", plan_stats=capacity.stats_of(eng.plan), eval_users=256)
    print(capacity.table(p))
    print("held after training steps: %.2f GiB, planned %.2f GiB" % (held_train / 2 ** 30, (p["training_bytes"] - p["components"][[k for k in p["components"] if k.startswith("hop-L")][0]]) / 2 ** 30))
    hopL = [v for k, v in p["components"].items() if k.startswith("hop-L")][0]
    assert abs((p["training_bytes"] - hopL) - held_train) < 0.10 * held_train
    model.predict_type = "TIE"
    idx, val = model.predict_device(list(range(256)), top_k=10)
    torch.cuda.synchronize()
    assert idx.shape == (256, 10) and int(idx.min()) >= 0 and int(idx.max()) < I
    held_all = torch.cuda.memory_allocated() - base
    transient = [v for k, v in p["eval_components"].items() if k.startswith("transient")][0]
    assert abs((p["total_bytes"] - transient) - held_all) < 0.10 * held_all, ((p["total_bytes"] - transient) / 2 ** 30, held_all / 2 ** 30)
    # per-non-zero / per-row size of the plan on this graph: the ratios the estimate without a live plan uses
    st = capacity.stats_of(eng.plan)
    est = capacity.PLAN_BYTES_PER_NNZ * st["nnz"] + capacity.PLAN_BYTES_PER_ROW * (U + I)
    assert abs(est - st["index_bytes"]) < 0.10 * st["index_bytes"], (est, st["index_bytes"])
    # ... and of the window-sweep plan this shape's whole hops run on (a [N x 256] slice is 2 GB: beyond the Infinity Cache)
    assert eng.sweep and abs(capacity.SWEEP_BYTES_PER_NNZ * st["nnz"] - st["sweep_bytes"]) < 0.25 * st["sweep_bytes"], (
        capacity.SWEEP_BYTES_PER_NNZ * st["nnz"], st["sweep_bytes"])
