"""CPU-side tests: config surface, iterator/sampler bookkeeping, model construction, C-ABI exports."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, build_model_from_fixture, load_golden, make_config, sub


def test_configurator_resolution_rules():
    cfg = make_config(["--alpha=0.5", "--loss=bpr_loss", "--batch_size=1024", "--recdim=32", "--newkey=[1,2]"])
    assert cfg["batch_size"] == 1024 and cfg.batch_size == 1024          # CLI overrides a key present in the file
    assert cfg["recdim"] == 32 and cfg["lr"] == 0.001 and cfg["weight_decay"] == 1e-4
    assert cfg["alpha"] == 0.5 and cfg["loss"] == "bpr_loss"              # CLI-only keys resolve through cmd_arg
    assert cfg["newkey"] == [1, 2] and "newkey" in cfg and "nope" not in cfg
    assert cfg["metric"] == ["Precision", "Recall", "NDCG"] and cfg["group_view"] is None
    assert cfg["no_cuda"] is False and cfg["save_flag"] is True and cfg["suffix"] == "" and cfg["logits"] == "cosin"
    with pytest.raises(KeyError):
        cfg["missing_key"]
    with pytest.raises(TypeError):
        cfg[3]
    with pytest.raises(SyntaxError):
        make_config(["alpha=0.5"])
    with pytest.raises(ValueError):
        make_config(["--a=b=c"])
    cfg.device = "cuda:0"                                                 # plain attribute (main.py:36)
    assert cfg.device == "cuda:0"
    assert "recdim=32" in cfg.params_str()


def test_data_iterator_semantics():
    from elimrec_amd import DataIterator
    a, b = list(range(10)), list(range(10, 20))
    batches = list(DataIterator(a, b, batch_size=4, shuffle=False))
    assert [len(x[0]) for x in batches] == [4, 4, 2] and batches[2] == [[8, 9], [18, 19]]
    assert len(DataIterator(a, b, batch_size=4)) == 3 and len(DataIterator(a, b, batch_size=4, drop_last=True)) == 2
    assert list(DataIterator(a, batch_size=5)) == [[0, 1, 2, 3, 4], [5, 6, 7, 8, 9]]     # single column -> flat lists
    np.random.seed(5)
    perm = np.random.permutation(10).tolist()
    np.random.seed(5)
    got = [x for bu, _ in DataIterator(a, b, batch_size=3, shuffle=True) for x in bu]
    assert got == perm                                                    # same RNG draw as the reference
    with pytest.raises(ValueError):
        DataIterator(a, b[:-1])


def test_sampler_bookkeeping_and_errors():
    from elimrec_amd import PairwiseSamplerV2, SyntheticDataset
    ds = SyntheticDataset(50, 80, 600, feat_dims=(8, 8, 8), seed=3)
    s = PairwiseSamplerV2(ds, batch_size=64)
    assert s.num_trainings == ds.train_matrix.nnz
    assert len(s) == (s.num_trainings + 63) // 64
    assert len(PairwiseSamplerV2(ds, batch_size=64, drop_last=True)) == s.num_trainings // 64
    with pytest.raises(ValueError):
        PairwiseSamplerV2(ds, neg_num=0)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            next(iter(PairwiseSamplerV2(ds, batch_size=64, device="cpu")))


def test_synthetic_dataset_contract():
    from elimrec_amd import SyntheticDataset
    ds = SyntheticDataset(300, 500, 5000, feat_dims=(16, 8, 4), seed=0)
    tot = ds.train_matrix + ds.valid_matrix + ds.test_matrix
    assert tot.max() == 1.0                                               # splits are disjoint, no duplicates
    assert (np.asarray(tot.sum(1)).ravel() >= 3).all() and (np.asarray(tot.sum(0)).ravel() >= 1).all()
    assert abs(ds.train_matrix.nnz / tot.nnz - 0.8) < 0.01
    assert ds.v_feat.shape == (500, 16) and ds.t_feat.shape == (500, 4)
    ds2 = SyntheticDataset(300, 500, 5000, feat_dims=(16, 8, 4), seed=0)
    assert (ds.train_matrix != ds2.train_matrix).nnz == 0 and torch.equal(ds.a_feat, ds2.a_feat)


def test_model_construction_matches_reference_surface(fixture_name):
    """Same state_dict key set as the reference, same adjacency, and (same seed, same draw order)
    bit-identical initial parameters."""
    from elimrec_amd import EliMRec, set_seed
    from helpers import FixtureDataset, fixture_argv
    g = load_golden(fixture_name)
    cfg = make_config(fixture_argv(g))
    set_seed(cfg["seed"])
    model = EliMRec(cfg, FixtureDataset(g))
    ref = sub(g, "init")
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref.keys())
    for k, v in ref.items():
        assert np.array_equal(sd[k].numpy(), v), k
    n = model.num_users + model.num_items
    import scipy.sparse as sp
    mine = sp.csr_matrix((model.adj_val.numpy(), model.adj_col.numpy(), model.adj_rowptr.numpy()), shape=(n, n)).tocoo()
    ref_adj = sp.coo_matrix((g["adj_values"], (g["adj_indices"][0], g["adj_indices"][1])), shape=(n, n))
    assert (abs(mine - ref_adj)).max() == 0.0
    assert model._adj_symmetric == (str(g["adj_type"]) in ("pre", "plain"))      # D^-1/2 A D^-1/2 and A itself


def test_hot_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = load_golden("ml3")
    model, _ = build_model_from_fixture(g, "cpu")
    u = torch.from_numpy(g["step1/users"])
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP"):
        model.bpr_loss(u, torch.from_numpy(g["step1/pos"]), torch.from_numpy(g["step1/neg"]))
    with pytest.raises(RuntimeError):
        model.predict([0, 1])
    from elimrec_amd import ops
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.fixed_order_sum(torch.zeros(4), torch.zeros(1))


def test_c_abi_library_exports_every_declared_symbol():
    """include/elimrec_hip.h is the contract: every function it declares must be exported by the
    built library and bound (with a signature) by the ctypes stub."""
    from elimrec_amd import _lib
    text = open(os.path.join(ROOT, "include", "elimrec_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(elimrec_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 18
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, name
    assert declared == set(_lib.SIGNATURES.keys())
    assert lib.elimrec_abi_version() == 1
    # size queries are host-only and must work without a GPU
    assert lib.elimrec_linear_bwd_w_workspace(76085, 64, 128) > 0
    assert lib.elimrec_score_workspace(128, 76085, 10) > 0


def test_checkpoint_name_and_metrics_info():
    g = load_golden("ml3")
    model, cfg = build_model_from_fixture(g, "cpu")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        cfg.alg_arg["path"] = repr(os.path.join(td, "ck"))
        name = model.getFileName()
        assert name.endswith("EliMRec-movielens-bpr_loss-.pth.tar") and os.path.isdir(os.path.join(td, "ck"))
    assert model.valid_evaluator.metrics_info().startswith("metrics:\tPrecision@10")


def test_csv_dataset_loader_remaps_ids_like_the_reference(tmp_path):
    """data/dataset.py:105-185,194-238 semantics: ids remapped by first appearance over
    concat(train, test, valid); features indexed by ORIGINAL item id; duplicates collapse."""
    from elimrec_amd import Dataset
    d = tmp_path / "dataset"
    d.mkdir()
    (d / "toy.train").write_text("10,7\n10,3\n20,7\n30,5\n10,7\n")
    (d / "toy.test").write_text("20,3\n30,9\n")
    (d / "toy.valid").write_text("10,9\n40,5\n")
    feats = np.arange(12 * 4, dtype=np.float32).reshape(12, 4)
    for tag in ("FeatureVideo_normal", "FeatureAudio_avg_normal", "FeatureText_stl_normal"):
        np.save(d / ("toy_%s.npy" % tag), feats + (0 if "Video" in tag else 100 if "Audio" in tag else 200))
    conf = {"data.input.dataset": "toy", "data.input.path": str(d), "data.convert.separator": ",",
            "data.column.format": "UI", "splitter": "given", "with_item_vat": True}
    ds = Dataset(conf)
    assert ds.userids == {10: 0, 20: 1, 30: 2, 40: 3}                  # first appearance: train, then test, then valid
    assert ds.itemids == {7: 0, 3: 1, 5: 2, 9: 3}
    assert (ds.num_users, ds.num_items) == (4, 4)
    assert ds.get_user_train_dict() == {0: [0, 1], 1: [0], 2: [2]}     # the duplicate (10,7) collapses
    assert ds.get_user_test_dict() == {1: [1], 2: [3]} and ds.get_user_valid_dict() == {0: [3], 3: [2]}
    tu, ti = ds.get_train_interactions()
    assert sorted(zip(tu, ti)) == [(0, 0), (0, 1), (1, 0), (2, 2)]
    assert torch.equal(ds.v_feat, torch.from_numpy(feats[[7, 3, 5, 9]]))   # rows of the ORIGINAL ids, in remap order
    assert float(ds.a_feat[0, 0]) == 100 + 7 * 4 and float(ds.t_feat[3, 1]) == 200 + 9 * 4 + 1
    with pytest.raises(NotImplementedError):
        Dataset(dict(conf, splitter="ratio"))
    with pytest.raises(ValueError):
        Dataset(dict(conf, **{"data.column.format": "XYZ"}))
    # --feature_load=block: nothing is read at construction; a block is the rows of its ORIGINAL ids out of the memory-mapped
    # file, row-normalised as the model's buffers are, and only the rows asked for are counted
    lazy = Dataset(dict(conf, feature_load="block"))
    assert not hasattr(lazy, "v_feat")
    fb = lazy.feature_blocks("a")
    assert fb.shape == (4, 4) and fb.rows_read == 0
    want = torch.nn.functional.normalize(ds.a_feat.float(), dim=1)
    assert torch.equal(fb[1:3], want[1:3]) and torch.equal(fb[3:4], want[3:4]) and fb.rows_read == 3 and fb.blocks == [(1, 3), (3, 4)]
    with pytest.raises(TypeError):
        fb[[0, 2]]
    with pytest.raises(RuntimeError):
        fb.to("cpu")
    assert torch.equal(ds.feature_blocks("v")[0:4], torch.nn.functional.normalize(ds.v_feat.float(), dim=1))


def test_csv_dataset_loader_matches_reference_on_the_golden_data(tmp_path):
    """Regenerate the exact files tests/golden/make_golden.py fed to the reference's Dataset (same seeded
    generator) and load them with elimrec_amd.Dataset: interactions, dicts and (normalised) features must
    equal what the reference produced (captured in tests/golden/ml3.npz)."""
    import importlib.util
    from helpers import GOLDEN, csr_dict
    from elimrec_amd import Dataset
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.write_dataset(str(tmp_path), "movielens", np.random.RandomState(11), 70, 110, (40, 24, 20))
    conf = {"data.input.dataset": "movielens", "data.input.path": str(tmp_path / "dataset"),
            "data.convert.separator": ",", "data.column.format": "UI", "splitter": "given", "with_item_vat": True}
    ds = Dataset(conf)
    g = load_golden("ml3")
    assert (ds.num_users, ds.num_items) == (int(g["num_users"]), int(g["num_items"]))
    tu, ti = ds.get_train_interactions()
    assert sorted(zip(tu, ti)) == sorted(zip(g["train_u"].tolist(), g["train_i"].tolist()))
    for split, got in (("train", ds.get_user_train_dict()), ("valid", ds.get_user_valid_dict()),
                       ("test", ds.get_user_test_dict())):
        want = csr_dict(g, split)
        assert {k: sorted(v) for k, v in got.items()} == {k: sorted(v) for k, v in want.items()}, split
    for m in "vat":
        got = torch.nn.functional.normalize(getattr(ds, m + "_feat").float(), dim=1).numpy()
        assert np.array_equal(got, g[m + "_feat"]), m


def test_torch_ops_library_registers_every_op():
    """libelimrec_torch.so loads without a GPU and registers the TORCH_LIBRARY(elimrec) schema; a CPU tensor
    has no kernel behind it -- the dispatcher says so instead of anything falling back."""
    import pytest
    import torch
    from elimrec_amd import torch_ops
    ns = torch_ops.load()
    for name in torch_ops.OPS:
        assert hasattr(ns, name), name
    with pytest.raises(NotImplementedError):
        ns.linear_fwd(torch.zeros(4, 4), torch.zeros(4, 4), None)


def test_plugin_tensor_classes_route_reads_through_the_controller():
    """elimrec_amd/plugin.py without a GPU: the parameter and loss classes behind `bpr_loss -> backward -> optimizer.step`.
    Reading a VALUE of an embedding parameter (directly, nested in a list, through .data) asks the controller to write the
    engine's master copy back first; metadata does not. `.grad` materialises a deferred backward on first read; assigning
    None (zero_grad) cancels it. A pending loss takes `backward()` without arguments as a request and makes itself real for
    every other use."""
    import torch
    from torch import nn
    from elimrec_amd.plugin import EmbeddingParameter, LazyGradParameter, PendingLoss

    class Ctl(object):
        def __init__(self):
            self.master_newer, self.synced, self.deferred, self.materialised, self.cancelled = True, 0, False, 0, 0
            self.requested, self.realised, self.grads_set = 0, 0, False

        def sync_params(self, implicit=False):
            self.synced += 1
            self.master_newer = False

        def grads_deferred(self):
            return self.deferred

        def materialise_grads(self):
            self.materialised += 1
            self.deferred = False

        def cancel_backward(self):
            self.cancelled += 1

        def request_backward(self, handle):
            self.requested += 1
            return True

        def realise_forward(self, handle=None):
            self.realised += 1

    ctl = Ctl()
    emb = nn.Embedding(5, 4)
    p = EmbeddingParameter(emb.weight.data)
    emb.weight = p
    p.__dict__["_elimrec_ctl"] = ctl
    assert isinstance(emb.weight, nn.Parameter) and [n for n, _ in emb.named_parameters()] == ["weight"]
    assert p.shape == (5, 4) and p.data_ptr() and p.numel() == 20 and p.device.type == "cpu" and ctl.synced == 0     # metadata
    _ = p.sum()
    assert ctl.synced == 1 and not ctl.master_newer
    _ = p.sum()
    assert ctl.synced == 1                                  # nothing newer: no second write-back
    ctl.master_newer = True
    _ = torch.cat([p, p])                                   # nested in a list
    assert ctl.synced == 2
    ctl.master_newer = True
    _ = p.data
    assert ctl.synced == 3
    ctl.master_newer = True
    _ = emb.state_dict()["weight"]                          # state_dict detaches the parameter: a read
    assert ctl.synced == 4
    w = LazyGradParameter(torch.ones(3))
    w.__dict__["_elimrec_ctl"] = ctl
    assert w.grad is None and ctl.materialised == 0
    ctl.deferred = True
    assert w.grad is None and ctl.materialised == 1         # a deferred backward becomes real on first read
    w.grad = torch.zeros(3)
    assert ctl.grads_set and w.grad is not None and ctl.cancelled == 0
    w.grad = None                                           # what zero_grad() does
    assert ctl.cancelled == 1 and w.grad is None
    opt = torch.optim.SGD([w, p], lr=0.1)                   # a torch optimizer sees ordinary parameters
    w.grad = torch.ones(3)
    opt.step()
    assert torch.allclose(w.detach(), torch.full((3,), 0.9))
    loss = torch.zeros((), requires_grad=True).clone().as_subclass(PendingLoss)
    loss.__dict__["_elimrec_ctl"] = ctl
    assert loss.shape == () and loss.requires_grad and ctl.realised == 0        # metadata
    assert loss.backward(retain_graph=True) is None and ctl.requested == 1      # a request, not an autograd pass
    assert loss.item() == 0.0 and ctl.realised == 1
    assert float(torch.stack([loss, loss]).sum()) == 0.0 and ctl.realised >= 2  # nested too
    assert type(loss.detach()) is torch.Tensor
