"""Shared helpers for the parity tests (fixture unpacking)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    if "word_rows" in g:      # the tiktok fixture keeps the occurring words' rows of word_embedding.weight: zero rows for the others
        rows, vocab = g["word_rows"], int(g["word_vocab"])
        for k in [k for k in g if k.endswith("@rows")]:
            full = np.zeros((vocab, g[k].shape[1]), g[k].dtype)
            full[rows] = g[k]
            g[k[:-5]] = full
    return g


def sub(g, prefix):
    """All arrays under 'prefix/' with the prefix stripped."""
    p = prefix + "/"
    return {k[len(p):]: v for k, v in g.items() if k.startswith(p)}


def csr_dict(g, split):
    users = g["%s_dict_users" % split]
    ptr = g["%s_dict_ptr" % split]
    items = g["%s_dict_items" % split]
    return {int(u): items[ptr[k]:ptr[k + 1]].tolist() for k, u in enumerate(users)}


def feats_of(g):
    return {m: g[m + "_feat"] for m in ("v", "a", "t") if (m + "_feat") in g}


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def row_err(a, b, rel=1e-4, floor=1e-7):
    """Row-wise criterion for gradient tables: max over rows r of
    ||a_r - b_r||_2 / (rel * ||b_r||_2 + floor * max_r ||b_r||_2). A value <= 1 means every row -- the hot items' and the
    rows a batch touched once alike -- is within `rel` of the reference's row (plus a floor of `floor` x the largest row
    norm for rows that are zero or pure round-off in the reference). 1-D tensors count as one row."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.ndim < 2:
        a, b = a.reshape(1, -1), b.reshape(1, -1)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    nb = np.sqrt((b * b).sum(1))
    nd = np.sqrt(((a - b) ** 2).sum(1))
    return float((nd / (rel * nb + floor * nb.max() + 1e-300)).max())


def assert_grad_close(mine, want, what, rel=1e-4, truth=None):
    """North-star tolerance on a gradient tensor, max-norm AND row by row against the reference arithmetic (fp32 oracle /
    fixture). `truth` (optional): the same gradient from the oracle evaluated in fp64. A row that misses the row-wise
    tolerance against `want` passes only if the fp32 reference ITSELF is that far from the exact value there (a sum
    that cancels) and the HIP result is no further from the exact value than twice the reference is."""
    e = rel_err(mine, want)
    assert e < rel, (what, "max-norm rel err", e)
    r = row_err(mine, want, rel=rel)
    if r > 1.0 and truth is not None:
        r_ref, r_mine = row_err(want, truth, rel=rel), row_err(mine, truth, rel=rel)
        assert r_mine <= max(1.0, 2.0 * r_ref), (what, "row-wise err / tolerance vs fp64", r_mine, "reference's own", r_ref)
        return
    assert r <= 1.0, (what, "row-wise err / tolerance", r)


# --------------------------------------------------------------------------- product-side builders
def make_config(argv_extra=()):
    """The real Configurator over the repo's NeuRec.properties / conf/EliMRec.properties."""
    from elimrec_amd import Configurator
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        return Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                            argv=["main.py"] + list(argv_extra))
    finally:
        os.chdir(cwd)


class FixtureDataset(object):
    """Dataset facade over a golden fixture (same accessors as elimrec_amd.dataset)."""

    def __init__(self, g):
        import scipy.sparse as sp
        import torch
        self.num_users, self.num_items = int(g["num_users"]), int(g["num_items"])
        self.dataset_name = str(g["dataset_name"])
        self._dicts = {s: csr_dict(g, s) for s in ("train", "valid", "test")}
        tu, ti = g["train_u"], g["train_i"]
        self.train_matrix = sp.csr_matrix((np.ones(len(tu)), (tu, ti)), shape=(self.num_users, self.num_items))
        for m in ("v", "a", "t"):
            if (m + "_feat") in g:
                setattr(self, m + "_feat", torch.from_numpy(g[m + "_feat"].copy()))
        if "words_tensor" in g:                      # data set "tiktok": (item id, word id) pairs (data/dataset.py:169-176)
            self.words_tensor = torch.from_numpy(g["words_tensor"].astype(np.int64))
        self._g = g

    def get_train_interactions(self):
        return self._g["train_u"].tolist(), self._g["train_i"].tolist()

    def feature_blocks(self, m):
        """--feature_load=block on a fixture: the reference's (already normalised) rows, served block by block."""
        from elimrec_amd.dataset import FeatureBlocks
        t = getattr(self, m + "_feat")
        return FeatureBlocks(t.shape[0], t.shape[1], lambda i0, i1: t[i0:i1], normalize=False)

    def get_user_train_dict(self, by_time=False):
        return self._dicts["train"]

    def get_user_valid_dict(self):
        return self._dicts["valid"]

    def get_user_test_dict(self):
        return self._dicts["test"]


def fixture_argv(g):
    argv = ["--recommender=EliMRec", "--data.input.dataset=%s" % str(g["dataset_name"]), "--alpha=%r" % float(g["alpha"]),
            "--loss=bpr_loss", "--recdim=%d" % int(g["recdim"]), "--layer_num=%d" % int(g["layer_num"]),
            "--adj_type=%s" % str(g["adj_type"]), "--modality=%s" % str(g["modality"]),
            "--mm_fusion_mode=%s" % str(g["mm_fusion_mode"]), "--verbose=0"]
    if "train_predict_type" in g:          # fixtures made before this field existed trained with the default (TIE)
        argv.append("--predict_type=%s" % str(g["train_predict_type"]))
    return argv


def build_model_from_fixture(g, device, params_prefix="init", extra_argv=()):
    """elimrec_amd.EliMRec on `device` holding the fixture's graph, features and parameters."""
    import torch
    from elimrec_amd import EliMRec
    cfg = make_config(fixture_argv(g) + list(extra_argv))
    model = EliMRec(cfg, FixtureDataset(g))
    with torch.no_grad():
        for m in ("v", "a", "t"):
            if (m + "_feat") in g and hasattr(model, m + "_feat") and torch.is_tensor(getattr(model, m + "_feat")):
                getattr(model, m + "_feat").copy_(torch.from_numpy(g[m + "_feat"]))   # exact reference bits
    sd = {k: torch.from_numpy(v.copy()) for k, v in sub(g, params_prefix).items()}
    model.load_state_dict(sd, strict=True)
    return model.to(device), cfg
