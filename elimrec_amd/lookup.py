"""Row-sharded constant tables with an all-to-all id lookup (csrc/lookup.hip; include/elimrec_hip.h "row-sharded constant
tables").

The folded constants S_m = mean_k A^k [0 ; F_m] ([N x D_m], the propagated form of the reference's v_feat / a_feat / t_feat,
models/EliMRec.py:233-236,366-381) and c = mean_k A^k [0 ; 1] are the largest tables of the model and are only READ, at
the <= 3B active rows of a batch. `FeatureShard` keeps one rank's rows of them -- users [ub[o], ub[o+1]) and items
[ib[o], ib[o+1]) -- in fp32, fp16 or bf16 and serves the three device steps of a lookup: counts (the all_to_all split
sizes), pack (owner side) and unpack (requester side: compact fp32 rows in active-row order + c). The exchange itself is
torch.distributed's all_to_all_single over RCCL (shard.py).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .ops import _dev, _stream

DTYPES = {"f32": (0, torch.float32), "f16": (1, torch.float16), "bf16": (2, torch.bfloat16)}
MAX_RANKS = 16


class RowOwnerMap(object):
    """Who owns which rows: rank o has users [ub[o], ub[o+1]) and items [ib[o], ib[o+1]) ("GPU g owns users
    [gU/8,(g+1)U/8) and items likewise")."""

    def __init__(self, U, I, world, ub=None, ib=None):
        if not 1 <= world <= MAX_RANKS:
            raise ValueError("1..%d ranks" % MAX_RANKS)
        self.U, self.I, self.world = int(U), int(I), int(world)
        even = lambda n: np.array([n * o // world for o in range(world + 1)], dtype=np.int64)
        self.ub = even(self.U) if ub is None else np.asarray(ub, dtype=np.int64)
        self.ib = even(self.I) if ib is None else np.asarray(ib, dtype=np.int64)
        assert self.ub[0] == 0 and self.ib[0] == 0 and self.ub[-1] == self.U and self.ib[-1] == self.I
        self._ub = (ctypes.c_int64 * (world + 1))(*self.ub.tolist())
        self._ib = (ctypes.c_int64 * (world + 1))(*self.ib.tolist())

    def rows(self, o):
        """(n_users, n_items) rank o owns."""
        return int(self.ub[o + 1] - self.ub[o]), int(self.ib[o + 1] - self.ib[o])

    def nodes(self, o):
        """Node ids of rank o's rows in local-table order (own users, then own items)."""
        return np.concatenate([np.arange(self.ub[o], self.ub[o + 1]), self.U + np.arange(self.ib[o], self.ib[o + 1])])

    def args(self):
        return self.world, self.U, self.I, self._ub, self._ib


class FeatureShard(object):
    """One rank's rows of [S_1 | .. | S_n | c] in `dtype` storage."""

    def __init__(self, owners, rank, tables, c, dtype="f32", device=None):
        """tables: the S_m as [N x D_m] fp32 tensors (or anything indexable by a LongTensor of node ids that returns such
        rows); c: [N] fp32."""
        if dtype not in DTYPES:
            raise ValueError("feature storage dtype must be one of %s (got %r)" % (sorted(DTYPES), dtype))
        self.owners, self.rank, self.dtype = owners, int(rank), dtype
        self.code, tdt = DTYPES[dtype]
        self.dims = [int(t.shape[1]) for t in tables]
        self.sum_d = sum(self.dims)
        device = tables[0].device if device is None else device
        es = 4 if self.code == 0 else 2
        n_c = 1 if self.code == 0 else 2
        self.row_elems = ((self.sum_d + n_c) * es + 15) // 16 * 16 // es
        self.row_bytes = self.row_elems * es
        nodes = torch.from_numpy(owners.nodes(self.rank)).to(tables[0].device)
        loc = torch.zeros(len(nodes), self.row_elems, dtype=tdt, device=device)
        off = 0
        for t in tables:
            loc[:, off:off + t.shape[1]] = t[nodes].to(device=device, dtype=tdt)
            off += t.shape[1]
        cr = c[nodes].to(device=device, dtype=torch.float32)
        if self.code == 0:
            loc[:, off] = cr
        else:                                   # c = hi + lo: 2 x 11 (fp16) / 2 x 8 (bf16) significant bits
            hi = cr.to(tdt)
            loc[:, off] = hi
            loc[:, off + 1] = (cr - hi.float()).to(tdt)
        self.table = loc
        self.device = loc.device

    @classmethod
    def from_local(cls, owners, rank, local_tables, local_c, dtype="f32"):
        """The same shard from rows this rank already holds: local_tables [own rows x D_m] fp32 in local-table order (own
        users, then own items), local_c [own rows] -- what the distributed fold (shard.py) produces; no rank ever held the
        full tables."""
        self = cls.__new__(cls)
        if dtype not in DTYPES:
            raise ValueError("feature storage dtype must be one of %s (got %r)" % (sorted(DTYPES), dtype))
        self.owners, self.rank, self.dtype = owners, int(rank), dtype
        self.code, tdt = DTYPES[dtype]
        self.dims = [int(t.shape[1]) for t in local_tables]
        self.sum_d = sum(self.dims)
        es = 4 if self.code == 0 else 2
        n_c = 1 if self.code == 0 else 2
        self.row_elems = ((self.sum_d + n_c) * es + 15) // 16 * 16 // es
        self.row_bytes = self.row_elems * es
        n = sum(owners.rows(self.rank))
        dev = local_tables[0].device
        loc = torch.zeros(n, self.row_elems, dtype=tdt, device=dev)
        off = 0
        for t in local_tables:
            assert t.shape[0] == n
            loc[:, off:off + t.shape[1]] = t.to(tdt)
            off += t.shape[1]
        cr = local_c.to(torch.float32)
        if self.code == 0:
            loc[:, off] = cr
        else:
            hi = cr.to(tdt)
            loc[:, off] = hi
            loc[:, off + 1] = (cr - hi.float()).to(tdt)
        self.table, self.device = loc, dev
        return self

    def nbytes(self):
        return self.table.numel() * self.table.element_size()

    def counts(self, acts, out=None):
        """acts int32 [W x R] -> int32 [W x W] on the device: [requester][owner]."""
        W, R = acts.shape
        assert W == self.owners.world and acts.is_contiguous()
        if out is None:
            out = torch.empty(W, W, dtype=torch.int32, device=acts.device)
        w, U, I, ub, ib = self.owners.args()
        _lib.check(_lib.load().elimrec_lookup_counts(_dev(acts, "acts", torch.int32), w, R, U, I, ub, ib,
                                                     _dev(out, "counts", torch.int32), _stream()), "lookup_counts")
        return out

    def pack(self, acts, send, send_off=None):
        """Owner side: send (uint8 [>= rows x row_bytes]) <- my rows of every requester's list, requester by requester."""
        W, R = acts.shape
        assert W == self.owners.world and acts.is_contiguous() and send.is_contiguous()
        w, U, I, ub, ib = self.owners.args()
        _lib.check(_lib.load().elimrec_lookup_pack(_dev(acts, "acts", torch.int32), w, R, U, I, ub, ib, self.rank,
                                                   self.table.data_ptr(), self.row_bytes, send.data_ptr(),
                                                   _dev(send_off, "send_off", torch.int32), _stream()), "lookup_pack")

    def unpack(self, act, rows, S_out, c_out, direct=False):
        """Requester side: rows (received chunks, owner by owner; direct: the local table, one rank) -> S_out fp32
        [R x sum_d] in the order of `act` (this rank's list) and c_out [R]."""
        R = act.numel()
        assert S_out.stride(1) == 1 and S_out.shape[1] == self.sum_d and S_out.shape[0] >= R and c_out.numel() >= R
        w, U, I, ub, ib = self.owners.args()
        src = self.table if direct else rows
        _lib.check(_lib.load().elimrec_lookup_unpack(_dev(act, "act", torch.int32), w, R, U, I, ub, ib, self.rank,
                                                     src.data_ptr(), self.row_bytes, self.code, self.sum_d, 1 if direct else 0,
                                                     _dev(S_out, "S_out"), S_out.stride(0), _dev(c_out, "c_out"), _stream()),
                   "lookup_unpack")
