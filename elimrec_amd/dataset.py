"""Dataset objects exposing what the model, sampler and evaluator read from the reference's
`data.Dataset` (data/dataset.py:21-43,319-354; util/tool.py:70-79):
  num_users, num_items, train/valid/test CSR matrices, get_train_interactions(),
  get_user_{train,valid,test}_dict(), v_feat / a_feat / t_feat, dataset_name.

`SyntheticDataset` is the seeded generator of SURVEY.md §8(d) (no real data ships with the
reference). `Dataset` loads `<path>/<name>.{train,valid,test}` CSV files plus feature arrays the
way data/dataset.py:105-185 does (splitter=given, ids remapped by first appearance).
"""
import os

import numpy as np
import scipy.sparse as sp
import torch


def csr_to_user_dict(matrix):
    """util/tool.py:70-79: {row: [column indices]} for non-empty rows, ascending row order."""
    matrix = matrix.tocsr()
    out = {}
    indptr, indices = matrix.indptr, matrix.indices
    for row in range(matrix.shape[0]):
        if indptr[row + 1] > indptr[row]:
            out[row] = indices[indptr[row]:indptr[row + 1]].copy().tolist()
    return out


class _DatasetBase(object):
    dataset_name = "synthetic"
    train_matrix = valid_matrix = test_matrix = None
    num_users = num_items = 0

    def get_user_train_dict(self, by_time=False):
        return csr_to_user_dict(self.train_matrix)

    def get_user_valid_dict(self):
        return csr_to_user_dict(self.valid_matrix)

    def get_user_test_dict(self):
        return csr_to_user_dict(self.test_matrix)

    def get_train_interactions(self):
        """(users, items) of the de-duplicated training interactions (data/dataset.py:347-354)."""
        coo = self.train_matrix.tocoo()
        return coo.row.tolist(), coo.col.tolist()

    def train_csr_arrays(self):
        m = self.train_matrix.tocsr()
        m.sort_indices()
        return m.indptr.astype(np.int64), m.indices.astype(np.int32)

    def __str__(self):
        n = int(self.train_matrix.nnz + self.valid_matrix.nnz + self.test_matrix.nnz)
        return "\n".join(["Dataset name: %s" % self.dataset_name,
                          "The number of users: %d" % self.num_users,
                          "The number of items: %d" % self.num_items,
                          "The number of ratings: %d" % n])


def _ones_csr(u, i, shape):
    m = sp.csr_matrix((np.ones(len(u), dtype=np.float64), (u, i)), shape=shape)
    m.data[:] = 1.0      # duplicates collapse to a single interaction
    m.sort_indices()
    return m


class FeatureBlocks(object):
    """A raw item-feature table read BLOCK BY BLOCK (--feature_load=block; the raw features of BASELINE.json configs[4] are 100 M
    rows x 256 x 3: no host holds them whole per rank): `shape`, and `table[i0:i1]` = rows i0 .. i1 - 1 in model item order, row-
    normalised as the model's buffers are (models/EliMRec.py:366-381), fp32. `reader(i0, i1)` returns the raw rows; `rows_read`
    counts what was asked for (tests: a rank reads its own item block and nothing else)."""

    def __init__(self, n_rows, dim, reader, normalize=True):
        self.shape = (int(n_rows), int(dim))
        self._reader, self._normalize = reader, normalize
        self.rows_read = 0
        self.blocks = []

    def __getitem__(self, key):
        if not isinstance(key, slice) or key.step not in (None, 1):
            raise TypeError("a block-loaded feature table is read by contiguous row ranges only (table[i0:i1])")
        i0, i1, _ = key.indices(self.shape[0])
        raw = torch.as_tensor(self._reader(i0, i1)).float()
        if raw.shape != (i1 - i0, self.shape[1]):
            raise ValueError("feature block [%d, %d) came back as %s" % (i0, i1, tuple(raw.shape)))
        self.rows_read += i1 - i0
        self.blocks.append((i0, i1))
        return torch.nn.functional.normalize(raw, dim=1).contiguous() if self._normalize else raw.contiguous()

    def to(self, *a, **k):
        raise RuntimeError("a block-loaded feature table has no whole-table form: this path needs --feature_load=full")

    float = contiguous = to


class SyntheticDataset(_DatasetBase):
    """Seeded synthetic interactions + features: every user >= 3 items, the rest
    user-uniform x item-Zipf(0.8), de-duplicated, every item >= 1 edge, random 80/10/10 split."""

    def __init__(self, num_users, num_items, num_interactions, feat_dims=(128, 128, 128), seed=0,
                 name="synthetic", zipf=0.8):
        rs = np.random.RandomState(seed)
        U, I = int(num_users), int(num_items)
        p = 1.0 / np.arange(1, I + 1, dtype=np.float64) ** zipf
        p /= p.sum()
        perm = rs.permutation(I)                       # popularity is not monotone in the item id
        rest = max(int(num_interactions) - 3 * U - I, 0)
        u = np.concatenate([np.repeat(np.arange(U), 3), rs.randint(U, size=rest), rs.randint(U, size=I)])
        i = np.concatenate([perm[rs.choice(I, size=3 * U, p=p)], perm[rs.choice(I, size=rest, p=p)], np.arange(I)])
        key = np.unique(u.astype(np.int64) * I + i)
        rs.shuffle(key)
        u, i = (key // I).astype(np.int32), (key % I).astype(np.int32)
        n = len(key)
        n_train, n_valid = int(n * 0.8), int(n * 0.1)
        self.num_users, self.num_items = U, I
        self.dataset_name = name
        self.train_matrix = _ones_csr(u[:n_train], i[:n_train], (U, I))
        self.valid_matrix = _ones_csr(u[n_train:n_train + n_valid], i[n_train:n_train + n_valid], (U, I))
        self.test_matrix = _ones_csr(u[n_train + n_valid:], i[n_train + n_valid:], (U, I))
        names = ("v_feat", "a_feat", "t_feat")
        g = torch.Generator().manual_seed(seed + 1)
        for name_, dm in zip(names, feat_dims):
            setattr(self, name_, torch.randn(I, int(dm), generator=g, dtype=torch.float32))

    def feature_blocks(self, m):
        """The same table as `<m>_feat`, served block by block (a stand-in for a file: the rows are sliced out of the seeded tensor)."""
        t = getattr(self, m + "_feat")
        return FeatureBlocks(t.shape[0], t.shape[1], lambda i0, i1: t[i0:i1])


class Dataset(_DatasetBase):
    """`splitter=given` loader (data/dataset.py:105-185,194-238): three CSV files of `user,item`
    rows, ids remapped by first appearance over concat(train, test, valid); item features indexed
    by ORIGINAL item id."""

    def __init__(self, conf):
        import pandas as pd
        self.conf = conf
        self.dataset_name = conf["data.input.dataset"]
        path = conf["data.input.path"]
        sep = conf["data.convert.separator"]
        fmt = conf["data.column.format"]
        columns = {"UIRT": ["user", "item", "rating", "time"], "UIR": ["user", "item", "rating"],
                   "UI": ["user", "item"]}.get(fmt)
        if columns is None:
            raise ValueError("'%s' is an invalid data column format!" % fmt)
        if conf["splitter"] != "given":
            raise NotImplementedError("'%s' is not supported!" % conf["splitter"])
        prefix = os.path.join(path, self.dataset_name)
        frames = [pd.read_csv(prefix + ext, sep=sep, header=None, names=columns) for ext in (".train", ".test", ".valid")]
        every = pd.concat(frames)
        users = every["user"].unique()
        items = every["item"].unique()
        self.userids = {k: n for n, k in enumerate(users)}
        self.itemids = {k: n for n, k in enumerate(items)}
        self.num_users, self.num_items = len(users), len(items)
        mats = []
        for f in frames:
            u = f["user"].map(self.userids).to_numpy()
            i = f["item"].map(self.itemids).to_numpy()
            mats.append(_ones_csr(u, i, (self.num_users, self.num_items)))
        self.train_matrix, self.test_matrix, self.valid_matrix = mats
        if conf["with_item_vat"] if "with_item_vat" in conf else True:
            self._load_features(path, list(self.itemids.keys()))

    def feature_blocks(self, m):
        """`<m>_feat` block by block, without reading the file whole: the .npy files of the generic loader are memory-mapped and
        a block gathers its rows by ORIGINAL item id (data/dataset.py:181-185); the .pt files of tiktok / kwai are torch.load-ed
        with mmap=True."""
        ids = np.asarray(self._feature_ids)
        name, path = self.dataset_name, self._feature_path
        if name in ("tiktok", "kwai"):
            fn = {"v": "%s/%s_visual_feat.pt" if name == "tiktok" else "%s/%s_feat_v.pt", "a": "%s/%s_audio_feat.pt"}.get(m)
            if fn is None:
                raise ValueError("no block loader for the '%s' features of the %s data set" % (m, name))
            table = torch.load(fn % (path, name), mmap=True)
            return FeatureBlocks(len(ids), table.shape[1], lambda i0, i1: table[torch.as_tensor(ids[i0:i1])])
        tag = {"v": "FeatureVideo_normal", "a": "FeatureAudio_avg_normal", "t": "FeatureText_stl_normal"}[m]
        table = np.load("%s/%s_%s.npy" % (path, name, tag), mmap_mode="r")
        return FeatureBlocks(len(ids), table.shape[1], lambda i0, i1: np.ascontiguousarray(table[ids[i0:i1]]))

    def _load_features(self, path, original_item_ids):
        name = self.dataset_name
        self._feature_path, self._feature_ids = path, original_item_ids
        if "feature_load" in self.conf and str(self.conf["feature_load"]) == "block":
            return                                              # nothing is read here: feature_blocks(m) serves the rows
        if name == "tiktok":                                    # data/dataset.py:164-177
            self.v_feat = torch.load("%s/%s_visual_feat.pt" % (path, name))[original_item_ids]
            self.a_feat = torch.load("%s/%s_audio_feat.pt" % (path, name))[original_item_ids]
            words = torch.load("%s/%s_textual_feat.pt" % (path, name)).detach()
            keep = [[self.itemids[int(i)], int(w)] for i, w in words.T.tolist() if int(i) in self.itemids]
            self.words_tensor = torch.tensor(keep).T
        elif name == "kwai":                                    # :178-180
            self.v_feat = torch.load("%s/%s_feat_v.pt" % (path, name))[original_item_ids]
        else:                                                   # :181-185
            load = lambda tag: torch.from_numpy(np.load("%s/%s_%s.npy" % (path, name, tag))[original_item_ids])
            self.v_feat = load("FeatureVideo_normal")
            self.a_feat = load("FeatureAudio_avg_normal")
            self.t_feat = load("FeatureText_stl_normal")
