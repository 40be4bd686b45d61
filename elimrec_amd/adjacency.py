"""The propagation matrix built on the GPU (csrc/adj.hip): `build_adj_device` returns the CSR of
create_adj_mat (models/EliMRec.py:309-354) -- int32 row pointers / sorted columns, fp32 values bit-identical to the
scipy construction (model.create_adj_mat) -- without materialising scipy matrices of the whole graph on the host.

What stays on the host is O(max degree): numpy evaluates d^p for every possible degree once, with the dtype flow of the
reference's branch (float32 for 'pre' / 'gcmc' / the fall-through, float64 then narrowed for 'norm', whose A + I is a
float64 matrix), so the inexact step is numpy's own."""
import numpy as np
import torch

from . import _lib
from .ops import _dev, _stream

ADJ_TYPES = {"plain": 0, "pre": 1, "gcmc": 2, "norm": 3}


def _pow_table(adj_type, max_deg):
    k = np.arange(max_deg + 2)
    with np.errstate(divide="ignore"):
        if adj_type == "pre":
            t = np.power(k.astype(np.float32), -0.5)
        elif adj_type == "norm":
            t = np.power(k.astype(np.float64), -1.0)
        else:
            t = np.power(k.astype(np.float32), -1.0)
    t[np.isinf(t)] = 0.0
    return t.astype(np.float32)


def build_adj_device(train_users, train_items, num_users, num_items, adj_type, device):
    """(rowptr int32 [N+1], col int32 [nnz], val fp32 [nnz]) on `device`. The interactions must be unique pairs (the
    reference's Dataset de-duplicates them); a duplicate raises, as scipy would silently sum it."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("build_adj_device builds on the GPU; use model.create_adj_mat on the host")
    u = torch.as_tensor(np.asarray(train_users, dtype=np.int64)).to(dev)
    i = torch.as_tensor(np.asarray(train_items, dtype=np.int64)).to(dev)
    E, U, I = int(u.numel()), int(num_users), int(num_items)
    N = U + I
    code = ADJ_TYPES.get(adj_type, 4)
    with_diag = code >= 3
    max_deg = 0
    if E:
        max_deg = int(max(torch.bincount(u, minlength=U).max(), torch.bincount(i, minlength=I).max()))
    table = torch.from_numpy(_pow_table(adj_type if adj_type in ADJ_TYPES else "mean", max_deg)).to(dev)
    nnz = 2 * E + (N if with_diag else 0)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    ws = torch.empty(int(lib.elimrec_build_adj_workspace(E, N, 1 if with_diag else 0)), dtype=torch.uint8, device=dev)
    _lib.check(lib.elimrec_build_adj(_dev(u, "users", torch.int64), _dev(i, "items", torch.int64), E, U, I, code,
                                     _dev(table, "pow_table"), table.numel(), _dev(rowptr, "rowptr", torch.int32),
                                     _dev(col, "col", torch.int32), _dev(val, "val"), _dev(err, "err", torch.int32),
                                     _dev(ws, "workspace", torch.uint8), ws.numel(), _stream()), "build_adj")
    bits = int(err.item())
    if bits & 1:
        raise ValueError("build_adj_device: duplicate (user, item) interactions")
    if bits & 2:
        raise RuntimeError("build_adj_device: degree beyond the power table")
    return rowptr, col[:nnz], val[:nnz]
