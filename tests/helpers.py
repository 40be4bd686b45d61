"""Shared helpers for the parity tests (fixture unpacking)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


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
