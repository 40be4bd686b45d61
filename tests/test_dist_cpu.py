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
import torch.nn.functional as F

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


class ColumnShardOracleEngine(OracleEngine):
    """CPU stand-in behind the cs_* interface of elimrec_amd/shard.py (ColumnShardTrainer): rank q owns columns
    [q*dl, (q+1)*dl) of [E_u ; E_i], propagates only those, and exchanges the layer means / adjoint sources of the active
    rows. The folded algebra (constant feature tables propagated once) is restated here with torch autograd."""
    PAD = -(1 << 30)
    lookup = False           # True: the constants S_m / c are ROW-sharded; a step fetches its active rows from the owners

    # ---- row-sharded constants (elimrec_amd/lookup.py restated with torch indexing; rows travel as raw fp32 bytes)
    def _lookup_setup(self):
        from elimrec_amd.lookup import RowOwnerMap
        m = self.m
        self.owners = RowOwnerMap(m.U, m.I, self.world)
        mine = torch.from_numpy(self.owners.nodes(self.rank))
        self.first_node = {int(n): k for k, n in enumerate(mine.tolist())}
        self.loc = torch.cat([self.S[k][mine] for k in self.mods] + [self.c[mine]], dim=1).contiguous()    # [own rows x (sumD + 1)]
        self.lookup_row_bytes = 4 * self.loc.shape[1]
        self.S = self.c = None                         # from here on the full tables do not exist on this rank
        self.owner_of = np.zeros(m.U + m.I, np.int64)
        for o in range(self.world):
            self.owner_of[self.owners.nodes(o)] = o

    def _owned(self, ids, o):
        ids = ids[ids >= 0].long()
        return ids[torch.from_numpy(self.owner_of[ids.numpy()] == o)]

    def cs_lookup_counts(self, acts):
        W = acts.shape[0]
        return torch.tensor([[len(self._owned(acts[r], o)) for o in range(W)] for r in range(W)], dtype=torch.int32)

    def cs_lookup_plan_counts(self, batches):
        out = np.zeros((len(batches), self.world), np.int64)
        for k, (u, p, n) in enumerate(batches):
            act = torch.unique(self.batch_keys(u, p, n).long())
            out[k] = np.bincount(self.owner_of[act.numpy()], minlength=self.world)
        return out

    def cs_lookup_pack(self, acts):
        rows = [self.loc[[self.first_node[int(n)] for n in self._owned(acts[r], self.rank)]] for r in range(acts.shape[0])]
        return torch.cat(rows).contiguous().view(torch.uint8).reshape(-1)

    def cs_lookup_recv(self, nbytes):
        return torch.empty(nbytes, dtype=torch.uint8)

    def cs_lookup_unpack(self, recv):
        rows = recv.view(torch.float32).view(-1, self.loc.shape[1])          # owner by owner, ascending inside an owner
        act = self.act[:self.n_act].long()
        order = torch.cat([self._owned(act, o) for o in range(self.world)])
        assert len(order) == len(rows) == self.n_act
        pos = torch.searchsorted(act, order)
        full = torch.empty_like(rows)
        full[pos] = rows
        self.rows_S, off = {}, 0
        for k in self.mods:
            D = self.m.feats[k].shape[1]
            self.rows_S[k] = full[:, off:off + D]
            off += D
        self.rows_c = full[:, off:off + 1]

    def cs_setup(self, world, rank, optimizer):
        m = self.m
        self.world, self.rank, self.opt = world, rank, optimizer
        self.dl = m.d // world
        self.cols = slice(rank * self.dl, (rank + 1) * self.dl)
        E = torch.cat([m.params["embedding_user.weight"], m.params["embedding_item.weight"]]).detach()
        self.shard = E[:, self.cols].clone().requires_grad_(True)
        self.m1, self.m2, self.t = torch.zeros_like(self.shard), torch.zeros_like(self.shard), 0
        self.mods = ["v"] if m.kwai else ["v", "a", "t"]
        N = m.U + m.I
        with torch.no_grad():
            def mean_prop(x0):
                xs = [x0]
                for _ in range(m.L):
                    xs.append(torch.sparse.mm(m.adj, xs[-1]))
                return torch.stack(xs).mean(0)
            self.S = {k: mean_prop(torch.cat([torch.zeros(m.U, m.feats[k].shape[1]), m.feats[k]])) for k in self.mods}
            self.c = mean_prop(torch.cat([torch.zeros(m.U, 1), torch.ones(m.I, 1)]))
        self.tail = [k for k in m.params if not k.startswith(("embedding_user.", "embedding_item."))]
        if self.lookup:
            self._lookup_setup()

    def _tables(self):
        """out0 = mean_k A^k X0 and the shared part (users: even k, items: odd k) for my columns, with autograd."""
        m = self.m
        xs = [self.shard]
        for k in range(1, m.L + 1):
            xs.append(torch.sparse.mm(m.adj, xs[-1]))
        inv = 1.0 / (m.L + 1)
        out0 = sum(xs) * inv
        nu = sum(x[:m.U] for k, x in enumerate(xs) if k % 2 == 0) * inv
        ni = sum(x[m.U:] for k, x in enumerate(xs) if k % 2 == 1) * inv if m.L >= 1 else torch.zeros(m.I, self.dl)
        return out0, torch.cat([nu, ni])

    def cs_plan(self, users, pos, neg):
        self.keys = self.batch_keys(users, pos, neg).long()
        act = torch.unique(self.keys)
        self.n_act = len(act)
        R = len(self.keys)
        self.act = torch.cat([act, torch.full((R - len(act),), self.PAD, dtype=torch.long)]).to(torch.int32)
        return self.act

    def cs_forward_hops(self):
        self.out0, self.narrow = self._tables()

    def cs_forward_rows(self, acts):
        W, R = acts.shape
        send = torch.zeros(W, R, 2 * self.dl)
        for p in range(W):
            ok = acts[p] >= 0
            r = acts[p][ok].long()
            send[p, :len(r), :self.dl] = self.out0.detach()[r]
            send[p, :len(r), self.dl:] = self.narrow.detach()[r]
        return send if W > 1 else None

    def cs_head(self, recv):
        m, d, n = self.m, self.m.d, self.n_act
        act = self.act[:n].long()
        if recv is None:
            o, nr = self.out0.detach()[act], self.narrow.detach()[act]
        else:
            W = recv.shape[0]
            o = torch.cat([recv[p, :n, :self.dl] for p in range(W)], dim=1)
            nr = torch.cat([recv[p, :n, self.dl:] for p in range(W)], dim=1)
        self.o_leaf, self.n_leaf = o.clone().requires_grad_(True), nr.clone().requires_grad_(True)
        blocks = [self.o_leaf]
        if self.lookup and self.world == 1 and recv is None:       # one rank without a trainer-side exchange: its own rows
            self.cs_lookup_unpack(self.cs_lookup_pack(self.act.view(1, -1)))
        for k in self.mods:
            Sk, ck = (self.rows_S[k], self.rows_c) if self.lookup else (self.S[k][act], self.c[act])
            blocks.append(F.linear(Sk, m.params[k + "_dense.weight"]) + ck * m.params[k + "_dense.bias"] + self.n_leaf)
        out = torch.cat(blocks, dim=1)
        fused_in = out if m.mm_fusion_mode == "concat" else out.view(n, -1, d).mean(1)
        is_user = (act < m.U)[:, None]
        y0 = torch.where(is_user, m._linear("embedding_user_after_GCN", fused_in), m._linear("embedding_item_after_GCN", fused_in))
        heads = [m._linear("s_dense_%s" % k, out[:, (h + 1) * d:(h + 2) * d]) for h, k in enumerate(self.mods)]
        Y = torch.cat([y0] + heads, dim=1)
        slot = torch.searchsorted(act, self.keys)
        r3 = Y[slot].view(-1, 3, Y.shape[1])
        w = [1.0] + [m.alpha if k in m.modality else 0.0 for k in self.mods]
        loss = 0
        for b, wk in enumerate(w):
            if wk:
                blk = r3[:, :, b * d:(b + 1) * d]
                loss = loss + wk * m.original_bpr_loss(blk[:, 0], blk[:, 1], blk[:, 2])
        self.loss = loss
        return loss.detach()

    def cs_backward_local(self, scale):
        m, d, n, W = self.m, self.m.d, self.n_act, self.world
        m.zero_grad()
        (self.loss * scale).backward()
        G, H = self.o_leaf.grad, self.o_leaf.grad + self.n_leaf.grad
        R = len(self.keys)
        send = torch.zeros(W, R, 2 * self.dl)
        for p in range(W):
            cs = slice(p * self.dl, (p + 1) * self.dl)
            send[p, :n, :self.dl], send[p, :n, self.dl:] = H[:, cs], G[:, cs]
        self.tail_names = [k for k in self.tail if m.params[k].grad is not None]
        self.wbuf = torch.cat([m.params[k].grad.reshape(-1) for k in self.tail_names])
        return send, self.wbuf

    def cs_backward_hops(self, recv2, acts):
        W, R = acts.shape
        N = self.m.U + self.m.I
        d_out0, d_nar = torch.zeros(N, self.dl), torch.zeros(N, self.dl)
        for p in range(W):                                   # rank order
            ok = acts[p] >= 0
            r = acts[p][ok].long()
            Hp, Gp = recv2[p, :len(r), :self.dl], recv2[p, :len(r), self.dl:]
            d_out0.index_add_(0, r, Gp)
            d_nar.index_add_(0, r, Hp - Gp)
        self.shard.grad = None
        ((self.out0 * d_out0).sum() + (self.narrow * d_nar).sum()).backward()

    @torch.no_grad()
    def cs_update(self):
        import math
        m, o = self.m, self.opt.inner
        self.t += 1
        g = self.shard.grad.add(self.shard, alpha=o.wd)
        self.m1.lerp_(g, 1 - o.b1)
        self.m2.mul_(o.b2).addcmul_(g, g, value=1 - o.b2)
        denom = (self.m2.sqrt() / math.sqrt(1 - o.b2 ** self.t)).add_(o.eps)
        self.shard.addcdiv_(self.m1, denom, value=-o.lr / (1 - o.b1 ** self.t))
        m.zero_grad()
        off = 0
        for k in self.tail_names:
            nel = m.params[k].numel()
            m.params[k].grad = self.wbuf[off:off + nel].view_as(m.params[k]).clone()
            off += nel
        self.opt.step()


class OracleOpt(object):
    def __init__(self, engine, g):
        self.inner = engine.eo.OracleAdam(engine.m.params, lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))

    def step(self):
        self.inner.step()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elimrec_amd.dist import DataParallelTrainer
    g = load_golden("ml3")
    eng = OracleEngine(g)
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


def test_two_rank_step_equals_single_process_big_batch(tmp_path):
    """The data-parallel flow of elimrec_amd/dist.py (the unfolded row-major forms): the whole backward on all-gathered
    head-gradient rows."""
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
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


def _cs_worker(rank, world, port, out_dir, lookup="off"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elimrec_amd.shard import ColumnShardTrainer
    g = load_golden("ml3")
    eng = ColumnShardOracleEngine(g)
    eng.lookup = lookup != "off"
    trainer = ColumnShardTrainer(eng, OracleOpt(eng, g), world_size=world, rank=rank)
    batches = []
    for t in (1, 2):
        u, p, n = (torch.from_numpy(g["step%d/%s" % (t, k)]) for k in ("users", "pos", "neg"))
        h = len(u) // world
        sl = slice(rank * h, (rank + 1) * h)
        batches.append((u[sl].clone(), p[sl].clone(), n[sl].clone()))
    if lookup == "planned":
        trainer.plan_lookup(batches)              # the epoch's split sizes ahead of time: no step reads its own back
    if lookup == "stale":
        # a plan made while the tensors held OTHER triplets (the two batches swapped), then rewritten in place: the entries
        # belong to tensor versions that are gone, and every step must learn its split sizes from the device instead
        right = [tuple(t.clone() for t in b) for b in batches]
        for b, other in zip(batches, right[::-1]):
            for t, o in zip(b, other):
                t.copy_(o)
        trainer.plan_lookup(batches)
        for b, mine in zip(batches, right):
            for t, o in zip(b, mine):
                t.copy_(o)
    losses = [float(trainer.global_loss(trainer.step(*b))) for b in batches]
    if lookup == "planned":
        # the same numbers in OTHER tensor objects (an epoch whose tensors are new): not the planned ones either
        again = [tuple(t.clone() for t in b) for b in batches]
        assert all(trainer._planned(*b) is not None for b in batches) and all(trainer._planned(*b) is None for b in again)
    extra = {}
    if eng.lookup:
        assert eng.S is None and eng.loc.shape[0] < eng.m.U + eng.m.I        # this rank never held the full constants
        assert trainer.lookup_syncs == (0 if lookup == "planned" else len(batches))
        extra["lookup_bytes"] = np.array([trainer.xgmi_bytes["all_to_all_lookup"], eng.lookup_row_bytes, eng.n_act, eng.loc.shape[0]])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.array(losses), shard=eng.shard.detach().numpy(),
             xgmi=np.array([trainer.xgmi_bytes[k] for k in ("all_gather", "all_to_all_fwd", "all_to_all_bwd", "all_reduce")]),
             **extra, **{k: eng.m.params[k].detach().numpy() for k in eng.tail})
    dist.destroy_process_group()


@pytest.mark.parametrize("lookup", ["planned", "synced", "stale"])
@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_constants_with_all_to_all_lookup_equal_single_process(tmp_path, world, lookup):
    """north_star's row shards + all-to-all index lookup for the V/A/T (folded-constant) tables, under gloo: every rank
    holds 1/world of the rows of S_m / c and NOTHING else of them; a step all-gathers the active ids (as before), every
    owner packs the rows the others asked for, one variable-size all_to_all moves them, the requester puts them in
    active-row order. `world` ranks equal ONE process on the whole batch (oracle); the bytes on the wire are the rows a
    rank does not own itself; with plan_lookup() no step synchronises to learn its split sizes; a plan whose tensors were
    rewritten in place since ("stale") is not used."""
    port = 33500 + (os.getpid() % 2000) + world + {"planned": 10, "synced": 0, "stale": 20}[lookup]
    mp.spawn(_cs_worker, args=(world, port, str(tmp_path), lookup), nprocs=world, join=True)
    rs = [dict(np.load(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    g = load_golden("ml3")
    eng = OracleEngine(g)
    from oracle import elimrec_oracle as eo
    opt = eo.OracleAdam(eng.m.params, lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    losses = []
    for t in (1, 2):
        u, p, n = (g["step%d/%s" % (t, k)] for k in ("users", "pos", "neg"))
        mlen = (len(u) // world) * world
        losses.append(eo.train_step(eng.m, opt, u[:mlen], p[:mlen], n[:mlen]))
    assert np.allclose(rs[0]["losses"], losses, atol=1e-6)
    E = torch.cat([eng.m.params["embedding_user.weight"], eng.m.params["embedding_item.weight"]]).detach().numpy()
    assert np.abs(np.concatenate([r["shard"] for r in rs], axis=1) - E).max() < 2e-5
    skip = ("losses", "shard", "xgmi", "lookup_bytes")
    for k in rs[0]:
        if k in skip:
            continue
        assert np.abs(rs[0][k] - eng.m.params[k].detach().numpy()).max() < 2e-5, k
        for r in rs[1:]:
            assert np.array_equal(r[k], rs[0][k]), k
    assert sum(int(r["lookup_bytes"][3]) for r in rs) == eng.m.U + eng.m.I              # the shards partition the rows
    for r in rs:
        sent, row_bytes, n_act, _ = (int(x) for x in r["lookup_bytes"])
        assert 0 < sent <= (world - 1) * 3 * (len(g["step2/users"]) // world) * row_bytes and sent % row_bytes == 0


@pytest.mark.parametrize("world", [2, 4])
def test_column_shard_ranks_equal_single_process_big_batch(tmp_path, world):
    """elimrec_amd/shard.py under gloo: `world` ranks, each owning recdim/world columns of [E_u ; E_i] and 1/world of
    the triplets, equal ONE process on the whole batch (oracle): loss, every embedding column, every projection weight;
    the projection weights are bitwise identical on all ranks; the bytes a rank sends per step are the closed form
    DESIGN.md quotes."""
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_cs_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [dict(np.load(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    g = load_golden("ml3")
    eng = OracleEngine(g)
    from oracle import elimrec_oracle as eo
    opt = eo.OracleAdam(eng.m.params, lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    losses = []
    for t in (1, 2):
        u, p, n = (g["step%d/%s" % (t, k)] for k in ("users", "pos", "neg"))
        mlen = (len(u) // world) * world
        losses.append(eo.train_step(eng.m, opt, u[:mlen], p[:mlen], n[:mlen]))
    assert np.allclose(rs[0]["losses"], losses, atol=1e-6)
    E = torch.cat([eng.m.params["embedding_user.weight"], eng.m.params["embedding_item.weight"]]).detach().numpy()
    assert np.abs(np.concatenate([r["shard"] for r in rs], axis=1) - E).max() < 2e-5
    for k in rs[0]:
        if k in ("losses", "shard", "xgmi"):
            continue
        assert np.abs(rs[0][k] - eng.m.params[k].detach().numpy()).max() < 2e-5, k
        for r in rs[1:]:
            assert np.array_equal(r[k], rs[0][k]), k
    R, d = 3 * (len(g["step1/users"]) // world), int(g["recdim"])
    dl = d // world
    n_tail = sum(rs[0][k].size for k in rs[0] if k not in ("losses", "shard", "xgmi"))
    assert list(rs[0]["xgmi"]) == [4 * R * (world - 1), 4 * R * 2 * dl * (world - 1), 4 * R * 2 * dl * (world - 1), 4 * n_tail]


# ----------------------------------------------------------------------------- item-sharded evaluation (shard_eval.py)
class TorchShardBackend(object):
    """ItemShardScorer's backend restated with the oracle's predict() formulas on one item block (tests only): the
    oracle object holds ALL users' cached rows and the rows of items [i0, i1)."""

    def __init__(self, om, i0, i1):
        self.om, self.i0, self.i1 = om, i0, i1

    def _ui(self, users):
        return torch.sigmoid(self.om.all_users[users] @ self.om.all_items.t())

    def row_sums(self, users):
        return self._ui(users).sum(1) if self.om.predict_type == "TIE" else None

    def score(self, users, row_sum, K, ptr, items, want_scores):
        om, ui = self.om, self._ui(users)
        if om.predict_type == "TIE":
            mean = (row_sum / float(self.n_total)).view(-1, 1)
            sc = torch.sigmoid(om.general_cm_fusion(ui, users) - om.general_cm_fusion(mean, users))
        elif om.predict_type == "TE":
            sc = torch.sigmoid(om.general_cm_fusion(ui, users))
        else:
            sc = torch.sigmoid(ui)
        sc = sc.clone()
        if ptr is not None:
            for b in range(len(users)):
                sc[b, items[ptr[b]:ptr[b + 1]].long()] = -float("inf")
        if not K:
            return sc, None, None
        order = torch.from_numpy(np.argsort(-sc.numpy(), axis=1, kind="stable")[:, :K].copy())
        return (sc if want_scores else None), (order + self.i0).to(torch.int32), torch.gather(sc, 1, order)

    def merge(self, cand_val, cand_idx, K):
        key = np.lexsort((cand_idx.numpy(), -cand_val.numpy()), axis=1)[:, :K]        # score desc, then id asc
        key = torch.from_numpy(key.copy())
        return torch.gather(cand_idx, 1, key), torch.gather(cand_val, 1, key)


def _eval_worker(rank, world, port, out_dir, ptype):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elimrec_amd.lookup import RowOwnerMap
    from elimrec_amd.shard_eval import Collectives, ItemShardScorer
    from helpers import csr_dict
    g = load_golden("ml3")
    om = OracleEngine(g).m
    om.predict_type = ptype
    cache = sub(g, "cache")
    own = RowOwnerMap(om.U, om.I, world)
    i0, i1 = int(own.ib[rank]), int(own.ib[rank + 1])
    om.set_cache(cache["all_users"], cache["all_items"][i0:i1],
                 {k: (v[i0:i1] if "_item_" in k else v) for k, v in cache.items() if k.startswith("pre_fusion")})
    backend = TorchShardBackend(om, i0, i1)
    backend.n_total = om.I
    scorer = ItemShardScorer(backend, Collectives(), own.ib)
    users = torch.from_numpy(g["evalbatch/users"]).long()
    train = csr_dict(g, "train")
    lists = [train.get(int(u), []) for u in users.tolist()]
    ptr = torch.zeros(len(users) + 1, dtype=torch.int64)
    ptr[1:] = torch.tensor(np.cumsum([len(x) for x in lists]))
    items = torch.tensor([i for x in lists for i in x], dtype=torch.int32)
    K = int(g["evalbatch/top_k"])
    idx, val = scorer.topk(users, K, ptr, items)
    full = scorer.scores(users, ptr, items)
    np.savez(os.path.join(out_dir, "eval%d.npz" % rank), idx=idx.numpy(), val=val.numpy(), scores=full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("ptype", ["TIE", "TE"])
@pytest.mark.parametrize("world", [2, 3])
def test_item_sharded_evaluation_equals_whole_catalogue(tmp_path, world, ptype):
    """shard_eval.ItemShardScorer under gloo (a torch backend built from the oracle's predict() formulas): every rank holds
    1/world of the item rows, phase-1 row sums are all-reduced into the catalogue-wide NDE mean, per-shard top-K lists are
    all-gathered and merged by (score desc, id asc). All ranks end with the same lists; they are the reference fixture's
    masked score matrix ranked under that rule, and the gathered score matrix is that matrix (1e-6)."""
    from helpers import csr_dict
    port = 36500 + (os.getpid() % 2000) + world + (10 if ptype == "TE" else 0)
    mp.spawn(_eval_worker, args=(world, port, str(tmp_path), ptype), nprocs=world, join=True)
    rs = [dict(np.load(tmp_path / ("eval%d.npz" % r))) for r in range(world)]
    for r in rs[1:]:
        assert np.array_equal(r["idx"], rs[0]["idx"]) and np.array_equal(r["val"], rs[0]["val"])
    g = load_golden("ml3")
    K = int(g["evalbatch/top_k"])
    if ptype == "TIE":
        ref = g["evalbatch/masked_scores"]                       # the reference's own predict() + train mask
    else:
        om = OracleEngine(g).m
        om.predict_type = "TE"
        c = sub(g, "cache")
        om.set_cache(c["all_users"], c["all_items"], {k: v for k, v in c.items() if k.startswith("pre_fusion")})
        ref = om.predict(g["evalbatch/users"]).numpy().copy()
        train = csr_dict(g, "train")
        for b, u in enumerate(g["evalbatch/users"].tolist()):
            ref[b, train.get(int(u), [])] = -np.inf
    got = rs[0]["scores"]
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    fin = ~np.isinf(ref)
    assert np.abs(got[fin] - ref[fin]).max() < 1e-6
    order = np.argsort(-got, axis=1, kind="stable")[:, :K]
    assert np.array_equal(rs[0]["idx"], order)                    # the merged lists rank the gathered matrix exactly
    ref_order = np.argsort(-ref, axis=1, kind="stable")[:, :K]
    moved = (order != ref_order).any(1)
    for r in np.nonzero(moved)[0]:                                # ... and the reference's, up to sub-1e-6 near-ties
        assert np.abs(ref[r][order[r]] - ref[r][ref_order[r]]).max() < 1e-6
    assert moved.sum() <= max(1, len(order) // 20)
