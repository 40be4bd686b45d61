"""Item-sharded full-catalogue evaluation (last row): every rank holds the cached rows of ALL users and of
ITS block of the items and scores every user block against that block only.

    reference                                             here, per user block
    predict(): ui = sigmoid(U I^T) over the catalogue     phase 1: sum_i sigmoid(u.i) over MY items
      NDE uses mean_i ui (models/EliMRec.py:107)            all_reduce(SUM) of [B] floats -> the catalogue-wide mean
    scores [B x I] -> host -> top-K (uni_evaluator.py)    phase 2: TIE/TE scores of MY items, train items masked, MY top-K
                                                           all_gather of W x [B x K] (id, score) -> merge by (score desc, id asc)

The orchestration is written against two small interfaces so that the same code runs on the HIP kernels over RCCL
(`HipShardBackend`, `Collectives`) and, in the CPU tests, on a torch restatement under gloo.
"""
import torch
import torch.distributed as dist


class Collectives(object):
    """The three collectives of a sharded evaluation on a process group. RCCL ("nccl") takes device tensors as they are; a
    gloo group with device tensors (several test processes on ONE GPU) stages through the host."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._gloo = dist.get_backend(group) == "gloo"

    def _host(self, t):
        return self._gloo and t.is_cuda

    def all_reduce_sum(self, t):
        if self._host(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, t):
        """[world] + t.shape, rank order (equal shapes on every rank)."""
        src = t.cpu() if self._host(t) else t
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=src.device)
        dist.all_gather_into_tensor(out.view(-1), src.contiguous().view(-1), group=self.group)
        return out.to(t.device)

    def all_gather_rows(self, t, counts):
        """Rows of every rank's [counts[r] x ...] tensor, concatenated in rank order (padded to the largest block on the wire)."""
        cap = max(counts)
        pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = self.all_gather(pad)
        return torch.cat([parts[r, :counts[r]] for r in range(self.world)])

    def all_to_all_rows(self, send, in_rows, out_rows):
        """send: [sum(in_rows) x c] (chunk r goes to rank r); returns [sum(out_rows) x c] (chunk r came from rank r)."""
        c = send.shape[1]
        src = send.cpu() if self._host(send) else send
        out = torch.empty(sum(out_rows), c, dtype=send.dtype, device=src.device)
        dist.all_to_all_single(out, src.contiguous(), output_split_sizes=list(out_rows), input_split_sizes=list(in_rows), group=self.group)
        return out.to(send.device)


class LocalCollectives(object):
    """One rank without a process group (lean tables on a single GPU): every collective is the identity."""
    world, rank = 1, 0

    def all_reduce_sum(self, t):
        return t

    def all_gather(self, t):
        return t.unsqueeze(0)

    def all_gather_rows(self, t, counts):
        return t

    def all_to_all_rows(self, send, in_rows, out_rows):
        return send


def collectives_for(group, world):
    return Collectives(group) if world > 1 else LocalCollectives()


class ItemShardScorer(object):
    """Top-K / scores of user blocks over an item-sharded catalogue. backend: row_sums(users) -> [B] fp32 partial sums
    (None when the predict type has no catalogue-wide mean); score(users, row_sum, K, train_ptr, train_items, want_scores)
    -> (scores [B x I_mine] or None, idx [B x K] catalogue ids or None, val [B x K] or None); merge(cand_val, cand_idx, K)
    -> (idx, val)."""

    def __init__(self, backend, coll, item_bounds):
        self.backend, self.coll = backend, coll
        self.bounds = [int(x) for x in item_bounds]          # [world + 1] item-id block boundaries
        self.i0, self.i1 = self.bounds[coll.rank], self.bounds[coll.rank + 1]

    def local_train_csr(self, train_ptr, train_items):
        """The masked-item CSR restricted to my item block, in local ids (index plumbing: three small torch ops)."""
        if train_ptr is None:
            return None, None
        keep = (train_items >= self.i0) & (train_items < self.i1)
        csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=keep.device), torch.cumsum(keep.to(torch.int64), 0)])
        ptr = csum[train_ptr].contiguous()
        items = (train_items[keep] - self.i0).to(torch.int32)
        if items.numel() == 0:
            items = torch.zeros(1, dtype=torch.int32, device=keep.device)
        return ptr, items.contiguous()

    def _row_sum(self, users):
        part = self.backend.row_sums(users)
        return None if part is None else self.coll.all_reduce_sum(part)

    def topk(self, users, K, train_ptr=None, train_items=None):
        total = self._row_sum(users)
        lp, li = self.local_train_csr(train_ptr, train_items)
        _, idx, val = self.backend.score(users, total, K, lp, li, False)
        W = self.coll.world
        ci = self.coll.all_gather(idx).permute(1, 0, 2).reshape(idx.shape[0], W * K).contiguous()
        cv = self.coll.all_gather(val).permute(1, 0, 2).reshape(idx.shape[0], W * K).contiguous()
        return self.backend.merge(cv, ci, K)

    def scores(self, users, train_ptr=None, train_items=None):
        """[B x I] score matrix on every rank (predict(): the reference returns the whole row)."""
        total = self._row_sum(users)
        lp, li = self.local_train_csr(train_ptr, train_items)
        sc, _, _ = self.backend.score(users, total, 0, lp, li, True)
        counts = [self.bounds[r + 1] - self.bounds[r] for r in range(self.coll.world)]
        cols = self.coll.all_gather_rows(sc.t().contiguous(), counts)         # [I x B]
        return cols.t().contiguous()


class HipShardBackend(object):
    """ItemShardScorer's backend on csrc/eval.hip: Y = [all user rows ; my item rows] of the cached tables."""

    def __init__(self, model, Y, i0, i1):
        from . import ops
        self.ops, self.m, self.Y, self.i0, self.i1 = ops, model, Y, int(i0), int(i1)
        self.n_items = self.i1 - self.i0
        self.sqn = torch.empty(Y.shape[0], 1 + model.S, dtype=torch.float32, device=Y.device)
        ops.row_sqnorms(Y, model.latent_dim, 1 + model.S, self.sqn)
        self._ws = None

    def _workspace(self, B, K, want_scores, phase):
        m = self.m
        if phase == 1:       # row sums: no score block (the library accepts the smaller of the two layouts)
            need = min(self.ops.score_workspace(B, m.num_users, self.n_items, m.S, 1, topk_only=True),
                       self.ops.score_workspace(B, m.num_users, self.n_items, m.S, 1))
        else:
            need = self.ops.score_workspace(B, m.num_users, self.n_items, m.S, max(K, 1), topk_only=not want_scores and K > 0,
                                            d=m.latent_dim)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.Y.device)
        return self._ws

    def _call(self, users, phase, row_sum, K, ptr, items, scores, idx, val):
        m = self.m
        self.ops.score_topk_shard(self.Y, m.num_users, self.n_items, users, m.latent_dim, m.S, m._head_mask(), m.fusion_mode,
                                  m.predict_type, self._workspace(users.numel(), K, scores is not None, phase), phase, row_sum,
                                  m.num_items, self.i0, scores=scores, K=K, topk_idx=idx, topk_val=val, train_ptr=ptr,
                                  train_items=items, sqnorm=self.sqn)

    def row_sums(self, users):
        if self.m.predict_type != "TIE":
            return None
        out = torch.zeros(users.numel(), dtype=torch.float32, device=self.Y.device)
        self._call(users, 1, out, 0, None, None, None, None, None)
        return out

    def score(self, users, row_sum, K, ptr, items, want_scores):
        B, dev = users.numel(), self.Y.device
        if row_sum is None:
            row_sum = torch.zeros(B, dtype=torch.float32, device=dev)
        sc = torch.empty(B, self.n_items, dtype=torch.float32, device=dev) if want_scores else None
        idx = torch.empty(B, K, dtype=torch.int32, device=dev) if K else None
        val = torch.empty(B, K, dtype=torch.float32, device=dev) if K else None
        self._call(users, 2, row_sum, K, ptr, items, sc, idx, val)
        return sc, idx, val

    def merge(self, cand_val, cand_idx, K):
        B = cand_val.shape[0]
        idx = torch.empty(B, K, dtype=torch.int32, device=cand_val.device)
        val = torch.empty(B, K, dtype=torch.float32, device=cand_val.device)
        self.ops.topk_merge(cand_val, cand_idx, K, idx, val)
        return idx, val
