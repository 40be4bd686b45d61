#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference). Nothing here ships to
the GPU box except the .npz files it writes. The reference is copied to a
scratch directory OUTSIDE the repo, its Cython helpers are rebuilt there for
this CPython, the five import shims of SURVEY.md §8(c) are applied, and the
reference's own `Net` / `EliMRec` / `UniEvaluator` / `PairwiseSamplerV2` are
driven on tiny synthetic datasets. Inputs and outputs are dumped as data.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
    python tests/golden/make_golden.py gcmc       # only the named fixture(s)

Fixtures (all fp32, CPU):
  ml3      3-modal generic loader path, concat fusion, adj_type=pre, rubi
  kwai     id+V only (dataset name "kwai"), 2 tables
  ablate   --modality=va, adj_type=norm (non-symmetric), mean fusion
  gcmc     adj_type=gcmc (bipartite but NOT symmetric), 4 layers, --modality=vt
  normal   --predict_type=normal (fused loss only: the single-modal heads get no gradient), adj_type=plain, 2 layers
  metrics  known-answer vectors for the C++ top-K + metric kernels
  sampler  one epoch of PairwiseSamplerV2 on the ml3 data (libc rand stream)
"""
import collections
import collections.abc
import importlib.machinery
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def prepare_copy():
    w = tempfile.mkdtemp(prefix="elimrec_ref_")
    for name in ("data", "evaluator", "models", "util", "conf", "main.py",
                 "NeuRec.properties", "setup.py"):
        src = os.path.join(REF, name)
        dst = os.path.join(w, name)
        if os.path.isdir(src):
            shutil.copytree(src, dst)
        else:
            shutil.copy(src, dst)
    subprocess.check_call(["chmod", "-R", "u+w", w])
    for root, dirs, files in os.walk(w):
        for d in list(dirs):
            if d in ("build", ".ipynb_checkpoints", "__pycache__"):
                shutil.rmtree(os.path.join(root, d))
                dirs.remove(d)
        for f in files:
            if f.endswith((".so", ".pyd", ".pyc")):
                os.remove(os.path.join(root, f))
    subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"],
                          cwd=w, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return w


def install_shims():
    collections.Iterable = collections.abc.Iterable
    for n in ("tensorflow", "torch_scatter"):
        m = types.ModuleType(n)
        m.__spec__ = importlib.machinery.ModuleSpec(n, None)
        sys.modules[n] = m
    sys.modules["torch_scatter"].scatter = scatter_mean_stub      # called on the "tiktok" path only (models/EliMRec.py:377)


def scatter_mean_stub(src, index, reduce="mean", dim=0):
    """torch_scatter.scatter(src, index, reduce='mean', dim=0) for the one call site of the reference (word embeddings of an item's
    words -> their mean): out[i] = sum of the rows of src with index i / their number (torch_scatter clamps the count at 1); rows
    0 .. max(index). torch_scatter itself is not installed in the build container -- this stub IS what the fixture pins."""
    import torch
    assert reduce == "mean" and dim == 0 and src.dim() == 2
    n = int(index.max()) + 1
    tot = torch.zeros(n, src.shape[1], dtype=src.dtype).index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype)).clamp_(min=1)
    return tot / cnt[:, None]


def synth_interactions(rs, U, I, per_user=(4, 9)):
    """Every user gets >=4 items; every item appears at least once."""
    pairs = set()
    for u in range(U):
        k = rs.randint(per_user[0], per_user[1])
        # Zipf-ish item popularity
        p = 1.0 / np.arange(1, I + 1) ** 0.8
        p /= p.sum()
        for i in rs.choice(I, size=k, replace=False, p=p):
            pairs.add((u, int(i)))
    for i in range(I):
        if not any(pi == i for _, pi in pairs):
            pairs.add((int(rs.randint(U)), i))
    pairs = sorted(pairs)
    rs.shuffle(pairs)
    # 70/15/15 split, but every user and every item must be present in the
    # union in first-appearance order that keeps ids == original ids.
    train, valid, test = [], [], []
    by_user = collections.defaultdict(list)
    for u, i in pairs:
        by_user[u].append(i)
    for u in range(U):
        items = by_user[u]
        n = len(items)
        n_te = max(1, n // 6)
        n_va = max(1, n // 6)
        test += [(u, i) for i in items[:n_te]]
        valid += [(u, i) for i in items[n_te:n_te + n_va]]
        train += [(u, i) for i in items[n_te + n_va:]]
    return train, valid, test


def write_dataset(w, name, rs, U, I, dims):
    d = os.path.join(w, "dataset")
    os.makedirs(d, exist_ok=True)
    train, valid, test = synth_interactions(rs, U, I)
    # the reference remaps ids by first appearance over concat(train,test,valid);
    # store ORIGINAL ids as a random relabelling so the remap path is exercised.
    uperm = rs.permutation(U) + 1000
    iperm = rs.permutation(I)
    for split, rows in (("train", train), ("valid", valid), ("test", test)):
        with open(os.path.join(d, "%s.%s" % (name, split)), "w") as f:
            for u, i in rows:
                f.write("%d,%d\n" % (uperm[u], iperm[i]))
    feats = {}
    import torch
    if name == "kwai":
        v = rs.randn(I, dims[0]).astype(np.float32)
        torch.save(torch.from_numpy(v), os.path.join(d, "kwai_feat_v.pt"))
        feats["v"] = v
    elif name == "tiktok":
        # data/dataset.py:165-176: visual / audio tensors indexed by ORIGINAL item id, and the text as a [2 x n] tensor of
        # (original item id, word id) pairs -- every item gets 1..6 words out of the reference's 11 574-word vocabulary
        for key, fn, dm in (("v", "visual", dims[0]), ("a", "audio", dims[1])):
            x = rs.randn(I, dm).astype(np.float32)
            torch.save(torch.from_numpy(x), os.path.join(d, "tiktok_%s_feat.pt" % fn))
            feats[key] = x
        pairs = [(i, int(wd)) for i in range(I) for wd in rs.randint(0, 11574, size=rs.randint(1, 7))]
        rs.shuffle(pairs)
        torch.save(torch.tensor(pairs, dtype=torch.int64).T.contiguous(), os.path.join(d, "tiktok_textual_feat.pt"))
    else:
        for key, fn, dm in (("v", "FeatureVideo_normal", dims[0]),
                            ("a", "FeatureAudio_avg_normal", dims[1]),
                            ("t", "FeatureText_stl_normal", dims[2])):
            x = rs.randn(I, dm).astype(np.float32)
            np.save(os.path.join(d, "%s_%s.npy" % (name, fn)), x)
            feats[key] = x
    return feats


def sample_triplets(rs, train_dict, I, B):
    users = np.array(sorted(train_dict.keys()))
    u = users[rs.randint(len(users), size=B)]
    p = np.empty(B, np.int64)
    n = np.empty(B, np.int64)
    for k, uu in enumerate(u):
        items = train_dict[int(uu)]
        p[k] = items[rs.randint(len(items))]
        s = set(items)
        while True:
            c = int(rs.randint(I))
            if c not in s:
                n[k] = c
                break
    return u.astype(np.int64), p, n


def run_fixture(w, name, dataset, U, I, dims, argv_extra, B, steps, seed):
    import torch
    os.chdir(w)
    rs = np.random.RandomState(seed)
    write_dataset(w, dataset, rs, U, I, dims)
    sys.argv = ["main.py", "--recommender=EliMRec", "--data.input.dataset=%s" % dataset,
                "--alpha=0.5", "--loss=bpr_loss", "--batch_size=%d" % B,
                "--verbose=0", "--save_flag=False"] + argv_extra
    from util.configurator import Configurator
    from util.tool import set_seed
    args = Configurator("./NeuRec.properties", default_section="hyperparameters")
    set_seed(args["seed"])
    import main
    import tqdm
    main.tqdm = tqdm.tqdm
    net = main.Net(args)
    rec = net.recommender
    ds = net.dataset
    out = {}
    out["num_users"] = np.int64(ds.num_users)
    out["num_items"] = np.int64(ds.num_items)
    out["recdim"] = np.int64(args["recdim"])
    out["layer_num"] = np.int64(args["layer_num"])
    out["alpha"] = np.float64(args["alpha"])
    out["lr"] = np.float64(args["lr"])
    out["weight_decay"] = np.float64(args["weight_decay"])
    out["adj_type"] = np.array(str(args["adj_type"]))
    out["dataset_name"] = np.array(dataset)
    out["modality"] = np.array(rec.modality)
    out["mm_fusion_mode"] = np.array(rec.mm_fusion_mode)
    out["train_predict_type"] = np.array(str(rec.predict_type))      # what the training loss is built for (:125-126)
    tu, ti = ds.get_train_interactions()
    out["train_u"] = np.asarray(tu, np.int32)
    out["train_i"] = np.asarray(ti, np.int32)
    for split, dct in (("train", ds.get_user_train_dict()), ("valid", ds.get_user_valid_dict()),
                       ("test", ds.get_user_test_dict())):
        keys = sorted(dct.keys())
        out["%s_dict_users" % split] = np.asarray(keys, np.int32)
        out["%s_dict_ptr" % split] = np.cumsum([0] + [len(dct[k]) for k in keys]).astype(np.int64)
        out["%s_dict_items" % split] = np.asarray([i for k in keys for i in dct[k]], np.int32)
    adj = rec.norm_adj.coalesce()
    out["adj_indices"] = adj.indices().numpy().astype(np.int64)
    out["adj_values"] = adj.values().numpy().astype(np.float32)
    out["v_feat"] = rec.v_feat.numpy()
    if dataset != "kwai":
        out["a_feat"] = rec.a_feat.numpy()
        out["t_feat"] = rec.t_feat.detach().numpy()
    if dataset == "tiktok":
        out["words_tensor"] = rec.words_tensor.numpy().astype(np.int64)      # [2 x n]: (item id after the remap, word id)
    for k, v in rec.state_dict().items():
        out["init/" + k] = v.detach().numpy().copy()

    train_dict = ds.get_user_train_dict()
    trs = np.random.RandomState(seed + 1)
    for t in range(1, steps + 1):
        u, p, n = sample_triplets(trs, train_dict, ds.num_items, B if t < steps else B - 5)
        out["step%d/users" % t] = u
        out["step%d/pos" % t] = p
        out["step%d/neg" % t] = n
        loss = rec.bpr_loss(torch.tensor(u), torch.tensor(p), torch.tensor(n))
        net.opt.zero_grad()
        loss.backward(retain_graph=True)
        out["step%d/loss" % t] = np.float32(loss.item())
        if t == 1:
            for k, prm in rec.named_parameters():
                if prm.grad is not None:
                    out["grad1/" + k] = prm.grad.detach().numpy().copy()
            # intermediates of the first forward (pins the layer-by-layer restatement)
            out["fwd1/all_users"] = rec.all_users.detach().numpy().copy()
            out["fwd1/all_items"] = rec.all_items.detach().numpy().copy()
            out["fwd1/i_emb"] = rec.i_emb.detach().numpy().copy()
            out["fwd1/v_emb"] = rec.v_emb.detach().numpy().copy()
        net.opt.step()
        if t in (1, steps):
            for k, v in rec.state_dict().items():
                out["after%d/%s" % (t, k)] = v.detach().numpy().copy()
    out["steps"] = np.int64(steps)

    # ---- evaluation: tables cached by the LAST training forward (pre-update params)
    out["cache/all_users"] = rec.all_users.detach().numpy().copy()
    out["cache/all_items"] = rec.all_items.detach().numpy().copy()
    for k, v in rec.all_s_embs.items():
        out["cache/" + k] = v.detach().numpy().copy()
    eval_users = sorted(ds.get_user_test_dict().keys())[:37]
    out["eval_users"] = np.asarray(eval_users, np.int64)
    for fmode in ("rubi", "hm", "sum"):
        rec.fusion_mode = fmode
        for ptype in ("TE", "TIE", "normal"):
            rec.predict_type = ptype
            if fmode != "rubi" and ptype == "normal":
                continue
            out["predict/%s/%s" % (fmode, ptype)] = np.asarray(rec.predict(eval_users), np.float32)
    rec.fusion_mode = "rubi" if "s_fusion_mode" not in args else args["s_fusion_mode"]
    for ptype in ("TE", "TIE"):
        rec.predict_type = ptype
        r, buf = rec.evaluate()
        out["evaluate/%s/valid" % ptype] = np.asarray(r, np.float32)
        out["evaluate/%s/valid_str" % ptype] = np.array(buf)
        r, buf = rec.test()
        out["evaluate/%s/test" % ptype] = np.asarray(r, np.float32)
    # per-user metric rows of the first eval batch on the test split (TIE, train items masked)
    rec.predict_type = "TIE"
    ev = rec.test_evaluator.evaluator
    users0 = list(ev.user_pos_test.keys())[:ev.batch_size]
    score = np.array(rec.predict(users0, None), dtype=np.float32)
    for idx, uu in enumerate(users0):
        score[idx][ev.user_pos_train.get(uu, [])] = -np.inf
    out["evalbatch/users"] = np.asarray(users0, np.int64)
    out["evalbatch/masked_scores"] = score.copy()
    res = ev.eval_score_matrix(score, [ev.user_pos_test[uu] for uu in users0], ev.metrics,
                               top_k=ev.max_top, thread_num=ev.num_thread)
    out["evalbatch/per_user_metrics"] = np.asarray(res, np.float32)
    out["evalbatch/metric_ids"] = np.asarray(ev.metrics, np.int32)
    out["evalbatch/top_k"] = np.int64(ev.max_top)
    if dataset == "tiktok":
        # word_embedding.weight is [11 574 x 128] and appears four times: keep the rows of the words that occur (the others feed
        # nothing) and, of the rest, how far coupled weight decay moved them -- helpers.load_golden puts the tables back
        rows = np.unique(out["words_tensor"][1])
        out["word_rows"], out["word_vocab"] = rows.astype(np.int64), np.int64(out["init/word_embedding.weight"].shape[0])
        rest = np.setdiff1d(np.arange(int(out["word_vocab"])), rows)
        for k in [k for k in out if k.endswith("word_embedding.weight")]:
            if k.startswith("after"):
                out[k + "@unused_max_move"] = np.float32(np.abs(out[k][rest] - out["init/word_embedding.weight"][rest]).max())
            if not k.startswith("init/"):
                out[k + "@rows"] = out[k][rows].copy()
        out["init/word_embedding.weight@rows"] = out["init/word_embedding.weight"][rows].copy()
        for k in [k for k in out if k.endswith("word_embedding.weight")]:
            del out[k]
    np.savez_compressed(os.path.join(OUT, "%s.npz" % name), **out)
    print("wrote %s.npz  (%d arrays)  loss=%s" % (name, len(out), [float(out["step%d/loss" % t]) for t in range(1, steps + 1)]))
    return net


def metrics_kat(w):
    os.chdir(w)
    from evaluator.backend.cpp.cpp_evaluator import CPPEvaluator
    ev = CPPEvaluator()
    out = {}
    rs = np.random.RandomState(7)
    cases = []
    # the SURVEY §4 tiny case (ties in row 0)
    s = np.array([[.5, .5, .5, .1, .9, .5], [1, 2, 3, 4, 5, 6]], np.float32)
    cases.append((s, [[4, 1], [0]], 3))
    # random rows with heavy ties (few distinct values), K=10
    s = rs.randint(0, 12, size=(16, 200)).astype(np.float32) / 16.0
    truth = [sorted(set(rs.randint(0, 200, size=rs.randint(1, 15)).tolist())) for _ in range(16)]
    cases.append((s, truth, 10))
    # distinct values, K=20, -inf masked entries
    s = rs.rand(8, 333).astype(np.float32)
    s[:, ::7] = -np.inf
    truth = [sorted(set(rs.randint(0, 333, size=rs.randint(1, 40)).tolist())) for _ in range(8)]
    cases.append((s, truth, 20))
    # K larger than #truth, single truth item
    s = rs.rand(5, 50).astype(np.float32)
    truth = [[int(rs.randint(50))] for _ in range(5)]
    cases.append((s, truth, 50))
    out["n_cases"] = np.int64(len(cases))
    for c, (s, truth, k) in enumerate(cases):
        res = ev.eval_score_matrix(np.ascontiguousarray(s), truth, [1, 2, 3, 4, 5], top_k=k, thread_num=2)
        out["case%d/scores" % c] = s
        out["case%d/truth_ptr" % c] = np.cumsum([0] + [len(t) for t in truth]).astype(np.int64)
        out["case%d/truth_items" % c] = np.asarray([i for t in truth for i in t], np.int32)
        out["case%d/top_k" % c] = np.int64(k)
        out["case%d/result" % c] = np.asarray(res, np.float32)  # [users, 5*k], metric ids 1..5
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("wrote metrics.npz")


def sampler_epoch(w, net):
    """One epoch of the reference sampler on the ml3 train dict. libc rand() is never seeded
    by the reference and nothing before this point draws from it, so the stream starts from
    the C library's default seed exactly as in a reference run."""
    os.chdir(w)
    from data.sampler import PairwiseSamplerV2
    ds = net.dataset
    np.random.seed(123)
    smp = PairwiseSamplerV2(ds, neg_num=1, batch_size=64, shuffle=True)
    us, ps, ns, lens = [], [], [], []
    for bu, bp, bn in smp:
        us += [int(x) for x in bu]
        ps += [int(x) for x in bp]
        ns += [int(x) for x in bn]
        lens.append(len(bu))
    out = {"users": np.asarray(us, np.int64), "pos": np.asarray(ps, np.int64), "neg": np.asarray(ns, np.int64),
           "batch_lens": np.asarray(lens, np.int64), "num_items": np.int64(ds.num_items),
           "len": np.int64(len(smp)), "np_seed": np.int64(123)}
    td = ds.get_user_train_dict()
    keys = sorted(td.keys())
    out["train_dict_users"] = np.asarray(keys, np.int32)
    out["train_dict_ptr"] = np.cumsum([0] + [len(td[k]) for k in keys]).astype(np.int64)
    out["train_dict_items"] = np.asarray([i for k in keys for i in td[k]], np.int32)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)
    print("wrote sampler.npz (%d samples, %d batches)" % (len(us), len(lens)))


def main():
    w = prepare_copy()
    sys.path.insert(0, w)
    install_shims()
    only = set(sys.argv[1:])
    want = lambda name: not only or name in only
    sys.argv = sys.argv[:1]
    try:
        if want("ml3") or want("sampler"):
            net = run_fixture(w, "ml3", "movielens", U=70, I=110, dims=(40, 24, 20),
                              argv_extra=["--recdim=32", "--layer_num=3"], B=64, steps=3, seed=11)
            sampler_epoch(w, net)
        if want("kwai"):
            run_fixture(w, "kwai", "kwai", U=50, I=120, dims=(48,),
                        argv_extra=["--recdim=64", "--layer_num=2"], B=48, steps=3, seed=22)
        if want("ablate"):
            run_fixture(w, "ablate", "movielens", U=64, I=96, dims=(20, 36, 28),
                        argv_extra=["--recdim=16", "--layer_num=3", "--adj_type=norm", "--modality=va",
                                    "--mm_fusion_mode=mean"], B=80, steps=3, seed=33)
        if want("gcmc"):
            run_fixture(w, "gcmc", "movielens", U=60, I=100, dims=(16, 12, 24),
                        argv_extra=["--recdim=32", "--layer_num=4", "--adj_type=gcmc", "--modality=vt"], B=72, steps=3,
                        seed=44)
        if want("normal"):
            run_fixture(w, "normal", "movielens", U=48, I=130, dims=(12, 20, 8),
                        argv_extra=["--recdim=64", "--layer_num=2", "--adj_type=plain", "--predict_type=normal"], B=56,
                        steps=3, seed=55)
        if want("tiktok"):
            # the word-bag text path (models/EliMRec.py:371-378): t_feat = scatter-mean of word embeddings, NOT normalised, and
            # word_embedding.weight a parameter that keeps receiving gradients through the retained graph (main.py:100)
            run_fixture(w, "tiktok", "tiktok", U=56, I=90, dims=(24, 16, 128),
                        argv_extra=["--recdim=32", "--layer_num=3"], B=60, steps=3, seed=66)
        if want("metrics"):
            metrics_kat(w)
    finally:
        os.chdir("/")
        shutil.rmtree(w, ignore_errors=True)


if __name__ == "__main__":
    main()
