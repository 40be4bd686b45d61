"""The N>1 training path (elimrec_amd/dist.py) on CPU: two gloo ranks, the oracle standing in
for the HIP kernels behind the same engine interface (batch_keys / forward_local / backward_global). Checks
that a world_size-2 step with per-rank batches B equals ONE single-process step on the
concatenated batch of 2B triplets, and that the replicas stay bitwise in sync."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, feats_of, load_golden, sub


class OracleEngine(object):
    """CPU stand-in for EliMRec's engine API, built from oracle/ (tests only)."""

    def __init__(self, g):
        from oracle import elimrec_oracle as eo
        self.eo = eo
        adj = eo.build_adj(g["train_u"], g["train_i"], int(g["num_users"]), int(g["num_items"]), str(g["adj_type"]))
        self.m = eo.OracleEliMRec(int(g["num_users"]), int(g["num_items"]), int(g["recdim"]), int(g["layer_num"]), adj,
                                  feats_of(g), sub(g, "init"), float(g["alpha"]), dataset_name=str(g["dataset_name"]),
                                  modality=str(g["modality"]), mm_fusion_mode=str(g["mm_fusion_mode"]))

    def named_parameters(self):
        return list(self.m.params.items())

    def _tables(self):
        """Y [N x Cy] with an autograd graph back to the parameters."""
        m = self.m
        au, ai = m.compute()
        s = m.gcn_cf()
        mods = ["v"] if m.kwai else ["v", "a", "t"]
        blocks = [torch.cat([au, ai])] + [torch.cat([s["pre_fusion_user_" + k], s["pre_fusion_item_" + k]]) for k in mods]
        return torch.cat(blocks, dim=1), mods

    def batch_keys(self, users, pos, neg):
        U = self.m.U
        return torch.stack([users, U + pos, U + neg], dim=1).reshape(-1).to(torch.int32)

    def forward_local(self, users, pos, neg, all_keys=None, rank=0, world_size=1):
        m, U, d = self.m, self.m.U, self.m.d
        self.Y, mods = self._tables()
        keys = self.batch_keys(users, pos, neg)
        self.all_keys = keys if all_keys is None else all_keys.clone()
        assert torch.equal(self.all_keys[rank * len(keys):(rank + 1) * len(keys)], keys)
        rows = self.Y.detach()[keys.long()].clone().requires_grad_(True)       # [3B x Cy]
        r3 = rows.view(-1, 3, rows.shape[1])
        w = [1.0] + [m.alpha if k in m.modality else 0.0 for k in mods]
        loss = 0
        for b, wk in enumerate(w):
            if wk:
                blk = r3[:, :, b * d:(b + 1) * d]
                loss = loss + wk * m.original_bpr_loss(blk[:, 0], blk[:, 1], blk[:, 2])
        loss.backward()
        return loss.detach(), rows.grad.detach()

    def backward_global(self, grad_rows, scale):
        dY = torch.zeros_like(self.Y)
        dY.index_add_(0, self.all_keys.long(), grad_rows * scale)
        self.m.zero_grad()
        self.Y.backward(dY)
        return self.m.grads()


class ShardedHeadOracleEngine(OracleEngine):
    """The same stand-in behind the engine interface of a sharded head backward (EliMRec in "batch" mode): local
    forward + head backward -> dOut rows of the local active nodes and the head-weight gradients; the trainer
    all-gathers the rows and all-reduces the weight-gradient buffer; the propagation backward is replicated."""
    dp_shards_head = True
    HEAD = ("embedding_user_after_GCN", "embedding_item_after_GCN", "s_dense_")

    def _out_tables(self):
        """Out [N x C] (graph + feature projections), with an autograd graph back to the parameters."""
        m = self.m
        m.compute()
        mods = ["v"] if m.kwai else ["v", "a", "t"]
        return torch.cat([m.m_emb[k] for k in ["i"] + mods], dim=1), mods

    def _head(self, out):
        """Y [N x Cy] from Out with the live head parameters."""
        m, d = self.m, self.m.d
        mods = ["v"] if m.kwai else ["v", "a", "t"]
        fused_in = out if m.mm_fusion_mode == "concat" else out.view(out.shape[0], -1, d).mean(1)
        y0 = torch.cat([m._linear("embedding_user_after_GCN", fused_in[:m.U]), m._linear("embedding_item_after_GCN", fused_in[m.U:])])
        heads = [m._linear("s_dense_%s" % k, out[:, (h + 1) * d:(h + 2) * d]) for h, k in enumerate(mods)]
        return torch.cat([y0] + heads, dim=1), mods

    def forward_local(self, users, pos, neg, all_keys=None, rank=0, world_size=1):
        m, d = self.m, self.m.d
        self.Out, _ = self._out_tables()
        self.out_leaf = self.Out.detach().clone().requires_grad_(True)
        Y, mods = self._head(self.out_leaf)
        self.keys = self.batch_keys(users, pos, neg)
        r3 = Y[self.keys.long()].view(-1, 3, Y.shape[1])
        w = [1.0] + [m.alpha if k in m.modality else 0.0 for k in mods]
        loss = 0
        for b, wk in enumerate(w):
            if wk:
                blk = r3[:, :, b * d:(b + 1) * d]
                loss = loss + wk * m.original_bpr_loss(blk[:, 0], blk[:, 1], blk[:, 2])
        self.loss = loss
        return loss.detach(), None

    def backward_local(self, scale):
        m = self.m
        m.zero_grad()
        (self.loss * scale).backward()
        n = len(self.keys)
        act = torch.unique(self.keys.long())
        rows = torch.zeros(n, self.Out.shape[1])
        rows[:len(act)] = self.out_leaf.grad[act]
        keys = torch.arange(n, dtype=torch.int32)         # padding slot r: node r with a zero row
        keys[:len(act)] = act.to(torch.int32)
        self.head_names = [k for k, v in m.params.items() if k.startswith(self.HEAD) and v.grad is not None]
        self.wbuf = torch.cat([m.params[k].grad.reshape(-1) for k in self.head_names])
        return rows, keys, self.wbuf

    def backward_rows_global(self, all_rows, all_keys):
        m = self.m
        head = {}
        off = 0
        for k in self.head_names:                      # views: the trainer's all-reduce may still be in flight
            n = m.params[k].numel()
            head[k] = self.wbuf[off:off + n].view_as(m.params[k])
            off += n
        d_out = torch.zeros_like(self.Out)
        d_out.index_add_(0, all_keys.long(), all_rows)
        m.zero_grad()
        self.Out.backward(d_out)
        grads = m.grads()
        grads.update(head)
        return grads


class OracleOpt(object):
    def __init__(self, engine, g):
        self.inner = engine.eo.OracleAdam(engine.m.params, lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))

    def step(self):
        self.inner.step()


def _worker(rank, world, port, out_dir, sharded):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elimrec_amd.dist import DataParallelTrainer
    g = load_golden("ml3")
    eng = (ShardedHeadOracleEngine if sharded else OracleEngine)(g)
    trainer = DataParallelTrainer(eng, OracleOpt(eng, g), world_size=world, rank=rank)
    losses = []
    for t in (1, 2):
        u, p, n = (torch.from_numpy(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg"))
        half = len(u) // 2
        sl = slice(rank * half, (rank + 1) * half)
        loss = trainer.step(u[sl], p[sl], n[sl])
        losses.append(float(trainer.global_loss(loss)))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.array(losses),
             **{k: v.detach().numpy() for k, v in eng.m.params.items()})
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("sharded", [False, True], ids=["gathered-grad-rows", "sharded-head-backward"])
def test_two_rank_step_equals_single_process_big_batch(tmp_path, sharded):
    """Both data-parallel flows of elimrec_amd/dist.py: the whole backward on all-gathered head-gradient rows, and the
    head backward sharded per rank (all-gather of dOut rows + all-reduce of the head-weight gradients)."""
    world = 2
    port = 29500 + (os.getpid() % 2000) + (7 if sharded else 0)
    mp.spawn(_worker, args=(world, port, str(tmp_path), sharded), nprocs=world, join=True)
    r0 = dict(np.load(tmp_path / "rank0.npz"))
    r1 = dict(np.load(tmp_path / "rank1.npz"))
    for k in r0:
        assert np.array_equal(r0[k], r1[k]), k                    # replicas bitwise in sync
    # single process, full batches (same 2*half triplets per step)
    g = load_golden("ml3")
    eng = OracleEngine(g)
    from oracle import elimrec_oracle as eo
    opt = eo.OracleAdam(eng.m.params, lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    losses = []
    for t in (1, 2):
        u, p, n = (g["step%d/%s" % (t, k)] for k in ("users", "pos", "neg"))
        m = (len(u) // 2) * 2
        losses.append(eo.train_step(eng.m, opt, u[:m], p[:m], n[:m]))
    assert np.allclose(r0["losses"], losses, atol=1e-6)
    for k, v in eng.m.params.items():
        assert np.abs(r0[k] - v.detach().numpy()).max() < 2e-5, k


def test_single_rank_trainer_matches_reference_fixture():
    """world_size 1 through the same trainer reproduces the golden losses."""
    from elimrec_amd.dist import DataParallelTrainer
    g = load_golden("ml3")
    eng = OracleEngine(g)
    trainer = DataParallelTrainer(eng, OracleOpt(eng, g))
    for t in (1, 2, 3):
        u, p, n = (torch.from_numpy(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg"))
        loss = trainer.step(u, p, n)
        assert abs(float(loss) - float(g["step%d/loss" % t])) < 1e-6
