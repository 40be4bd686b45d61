"""Column-sharded training of EliMRec: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI),
every rank owns a COLUMN slice of the graph table and a share of the triplets.

Why columns. With the constant feature tables folded out of the graph (model.py), the only
table that goes through the LightGCN hops (models/EliMRec.py:238-248) is X^0 = [E_u ; E_i], N x d. A hop
X^k = A X^(k-1) is independent per column, so rank q can own columns [q*dl, (q+1)*dl), dl = d / world, of
  * the embedding parameters, their gradient and their Adam moments (master copy, slab-major, csrc/slab.hip),
  * every layer table of the forward and of the adjoint propagation,
and run all 2L hops on its slice with NO communication: per-rank propagation work is 1/world of the single-GPU
work. A row partition (the reference has no counterpart; SURVEY 8(e) lists it as the north star's default) needs
every hop's full input on every rank: an all-gather of N x d floats per hop, 6 x 29 MB per step at the Tiktok shape.
What a column partition exchanges per step is only what the batch touches -- per rank, with R = 3B slots:
  forward   all_gather  R int32 node ids of the rank's active rows               (12 KiB at B = 2048)
            all_to_all  R x 2*dl floats per peer: the layer means (id block | shared part) of the PEER's active
                        rows in MY columns                                        (R*2d*4 B = 3.1 MB per rank in total)
  backward  all_to_all  R x 2*dl floats per peer: [H | G] adjoint sources of MY active rows in the PEER's columns
            all_reduce  the projection-weight gradients                           (0.3 MB)
Everything after the graph is row-wise and runs on the rank's own B triplets with replicated projection weights
(data parallel). Rows of the same node contributed by several ranks are summed in rank order by one kernel
(elimrec_slab_merge_rows): no float atomics, every rank's update is bitwise reproducible, and the dense weights stay
in lock-step because every rank applies the same all-reduced gradient.

world = 1 is the same code without the collectives; it is also the fastest single-GPU path (bench.py): the adjoint
writes the gradient, and Adam updates the master copy, in the layout the hops read.

The trainer talks to an `engine` through the cs_* methods below; ColumnShardEngine implements them on the HIP
kernels, tests/test_dist_cpu.py injects a CPU stand-in built from the oracle to run the same trainer under gloo.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops, program, slab
from .lookup import FeatureShard, RowOwnerMap

PAD_KEY = -(1 << 30)


def _all_gather_parts(loc, world, group):
    """[world] tensors like `loc`, one per rank (host-staged when the group is gloo and `loc` lives on a device)."""
    if loc.is_cuda and dist.get_backend(group) == "gloo":
        host = loc.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return [p.to(loc.device) for p in parts]
    parts = [torch.empty_like(loc) for _ in range(world)]
    dist.all_gather(parts, loc, group=group)
    return parts


LOSS_RING = 1 << 16       # steps whose loss tensors stay valid (256 KB of device memory)


def _once(fn):
    """Memoise a no-argument predicate per engine: the launch-form switches read the environment once, not every step."""
    key = "_once_" + fn.__name__

    def wrapper(self):
        v = self.__dict__.get(key)
        if v is None:
            v = self.__dict__[key] = bool(fn(self))
        return v
    wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
    return wrapper


class ColumnShardTrainer(object):
    def __init__(self, engine, optimizer, world_size=1, rank=0, group=None):
        self.engine, self.opt, self.world, self.rank, self.group = engine, optimizer, int(world_size), int(rank), group
        self.profile_kernels = False
        self._events = []
        self._scale = None
        self._buf = {}
        engine.cs_setup(self.world, self.rank, optimizer)
        # several ranks, or ELIMREC_SHARD_MULTI=1: one rank runs the multi-rank code path, collectives included (a one-rank
        # RCCL group exercises every collective call of the step on a single GPU -- tests)
        self.multi = getattr(engine, "multi", self.world > 1)
        self._gloo = None
        if isinstance(engine, ColumnShardEngine) and self.multi:
            engine.multi_aux = True         # plan / weight packing / source bits on the second stream, as with one rank
            engine.defer_wgrads = True      # weight gradients behind the adjoint hops' tiles, their all-reduce under the last hop
        # the HIP engine's phases without their torch.no_grad() wrappers: step() enters no_grad once (host time)
        self._hip_engine = isinstance(engine, ColumnShardEngine)

        def phase(name):
            f = getattr(type(engine), name)
            return getattr(f, "__wrapped__", f).__get__(engine) if self._hip_engine else getattr(engine, name)
        self._ph = {n: phase(n) for n in ("cs_plan", "cs_forward_hops", "cs_forward_rows", "cs_head", "cs_backward_local",
                                          "cs_backward_hops", "cs_update")}
        ctl = getattr(getattr(engine, "model", None), "_plugin", None) if self._hip_engine else None
        if ctl is not None:
            ctl.attach(engine, self)        # model.bpr_loss -> backward -> optimizer.step runs on THIS engine (plugin.py)
        self.xgmi_bytes = dict(all_gather=0, all_to_all_fwd=0, all_to_all_bwd=0, all_reduce=0)   # sent per rank, last step
        # row-sharded constant tables (engine.lookup): the all_to_all split sizes of every planned batch, as host integers
        self.lookup = bool(getattr(engine, "lookup", False)) and bool(getattr(engine, "lookup_exchange", True))
        self._lookup_plan = {}
        self._lookup_owner = {}
        self.lookup_syncs = 0          # steps that had to read their split sizes back from the device (no plan entry)
        if self.lookup:
            self.xgmi_bytes["all_to_all_lookup"] = 0

    def prestage(self, batches):
        """The (users, pos, neg) tensors of the coming steps, complete on the device now (ColumnShardEngine.prestage)."""
        if self._hip_engine:
            self.engine.prestage(list(batches))

    def set_optimizer(self, optimizer):
        """The optimizer whose hyper-parameters and projection-weight state the engine's Adam launches use (a trainer made
        before the caller's optimizer existed: plugin.py)."""
        self.opt = self.engine.opt = optimizer
        if self._hip_engine:
            self.engine._tail_plan = None

    def _like(self, name, t, lead=None):
        shape = tuple(t.shape) if lead is None else (lead,) + tuple(t.shape)
        key = (name, shape, t.dtype, t.device)                    # one buffer per batch size: stable addresses
        b = self._buf.get(key)
        if b is None:
            b = self._buf[key] = torch.empty(shape, dtype=t.dtype, device=t.device)
        return b

    # ---- collectives. RCCL ("nccl") takes the device buffers as they are. A gloo group with device tensors -- the
    # multi-process tests that run several ranks on ONE GPU, where RCCL refuses duplicate devices -- stages through the
    # host: same trainer, same engine, same kernels, only the transport differs.
    class _Done(object):
        def wait(self):
            return True

    class _OnStream(object):
        """Handle of a collective enqueued on the exchange stream: wait() orders the CURRENT stream behind it."""

        def __init__(self, stream, recorded=None):
            self.stream, self.recorded = stream, recorded

        def wait(self):
            if self.recorded is not None:       # the collective alone: not what was enqueued behind it on that stream since
                program.wait(torch.cuda.current_stream(), self.recorded)
            else:
                program.sync(torch.cuda.current_stream(), self.stream)
            return True

    def _native_comm(self):
        """The library's own RCCL communicator (csrc/program.hip) for the step's exchanges: collectives become C-ABI calls
        enqueued on our streams -- no torch.distributed host path (~30 us per collective), and a native step program can list
        them. Backend "nccl" only (ELIMREC_NATIVE_COMM=0: torch.distributed's collectives, as the gloo-staged tests use)."""
        if self.__dict__.get("_comm") is None:
            import os
            self._comm = False
            if (self.multi and self._hip_engine and os.environ.get("ELIMREC_NATIVE_COMM", "1") != "0" and dist.is_available()
                    and dist.is_initialized() and dist.get_backend(self.group) == "nccl"):
                lib = _lib.load()
                dev = self.engine.model._device()
                ident = (ctypes.c_char * 128)()
                ok = 1
                if self.rank == 0:
                    ok = 0 if lib.elimrec_comm_unique_id(ident) else 1
                t = torch.frombuffer(bytearray(bytes(ident)), dtype=torch.uint8).to(dev)
                if self.world > 1:
                    dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                raw = bytes(t.cpu().numpy().tobytes())
                ident = (ctypes.c_char * 128).from_buffer_copy(raw)
                handle = ctypes.c_void_p()
                if ok and lib.elimrec_comm_create(ident, self.world, self.rank, ctypes.byref(handle)) != 0:
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                if self.world > 1:
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)      # all ranks or none
                if int(flag.item()) == 1:
                    # the exchanges that run beside the main stream (ids, constants' rows, weight gradients) share the engine's
                    # second stream with the plan they follow: one hand-over fewer on the way plan -> ids -> rows (each costs
                    # 10-20 us of command-processor latency), and the stream's order is the order they are needed in
                    aux = self.engine._aux_stream() if hasattr(self.engine, "_aux_stream") else None
                    self._comm, self._comm_stream = handle, (aux if aux is not None else torch.cuda.Stream())
                elif handle:
                    lib.elimrec_comm_destroy(handle)
        return self._comm or None

    def _staged(self, t):
        if self._gloo is None:
            self._gloo = dist.get_backend(self.group) == "gloo"
        return self._gloo and t.is_cuda

    def _all_gather(self, out, inp, after=None):
        """after: the stream that produced `inp` when it is not the current one (the planner's second stream)."""
        comm = self._native_comm()
        if comm is not None:            # on the exchange stream, under whatever the caller enqueues next on its own
            src = after if after is not None else torch.cuda.current_stream()
            if src is not self._comm_stream and src.cuda_stream != self._comm_stream.cuda_stream:
                program.sync(self._comm_stream, src)
            with torch.cuda.stream(self._comm_stream):
                _lib.check(_lib.load().elimrec_comm_all_gather(comm, inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size(),
                                                               ops._stream()), "comm_all_gather")
            # (an event of its own: the constants' row exchange is enqueued on this stream right behind, and whoever needs the
            # ids only -- the rows launch, the source bits -- must not wait for that)
            return self._OnStream(self._comm_stream, program.record(self._comm_stream))
        if after is not None:
            program.sync(torch.cuda.current_stream(), after)
        if not self._staged(inp):
            return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu(), group=self.group)
        out.copy_(host)
        return self._OnStream(torch.cuda.current_stream()) if out.is_cuda else self._Done()

    def _all_to_all(self, out, inp):
        comm = self._native_comm()
        if comm is not None:            # on the compute stream, between the kernels that produce and consume the rows
            _lib.check(_lib.load().elimrec_comm_all_to_all(comm, inp.data_ptr(), out.data_ptr(), inp[0].numel() * inp.element_size(),
                                                           ops._stream()), "comm_all_to_all")
            return
        if not self._staged(inp):
            # (synchronous form = on the compute stream; async_op + wait costs 24 us less host time per step and 23 us
            # more of stream hand-over: 0.461 against 0.438 ms per step over a one-rank RCCL group)
            return dist.all_to_all_single(out, inp, group=self.group)
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), group=self.group)
        out.copy_(host)

    def _all_to_all_v(self, out, inp, out_splits, in_splits, sizes=None):
        """Variable-size exchange of flat uint8 buffers (the looked-up rows): split sizes in bytes, per peer. sizes: the same
        numbers as a host int64 array [in_splits | out_splits] for the library's communicator."""
        comm = self._native_comm()
        if comm is not None:
            if sizes is None:
                sizes = (ctypes.c_int64 * (2 * self.world))(*(list(in_splits) + list(out_splits)))
            _lib.check(_lib.load().elimrec_comm_all_to_all_v(comm, inp.data_ptr(), out.data_ptr(), sizes, ops._stream()), "comm_all_to_all_v")
            return
        if not self._staged(inp):
            return dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
        out.copy_(host)

    def plan_lookup(self, batches):
        """Row-sharded constant tables: the split sizes of every batch's row exchange, computed ahead for a whole epoch (one
        device pass + one device->host copy + one all_gather of a [batches x world] int table) so that no step has to stop
        and read its own sizes back. `batches`: the (users, pos, neg) the following step() calls will get, in order; the
        plan replaces the previous one (an epoch's tensors may reuse the addresses of the last epoch's). A step whose batch
        is not in the plan reads its sizes from the device (one synchronisation per step)."""
        self._lookup_plan = {}
        self._lookup_size_arrays = {}
        self._lookup_owner = {}
        if not self.lookup:
            return
        eng, W = self.engine, self.world
        if W > 1:
            # the [batches x W] table below is all_gathered: every rank must bring the same number of batches
            nb = torch.tensor([len(batches), -len(batches)], dtype=torch.int64)
            if dist.get_backend(self.group) == "gloo":
                dist.all_reduce(nb, op=dist.ReduceOp.MAX, group=self.group)
            else:
                nb_d = nb.to(eng.model._device())
                dist.all_reduce(nb_d, op=dist.ReduceOp.MAX, group=self.group)
                nb = nb_d.cpu()
            if int(nb[0]) != -int(nb[1]):
                raise ValueError("plan_lookup: the ranks bring different numbers of batches (%d here, between %d and %d over the job)"
                                 % (len(batches), -int(nb[1]), int(nb[0])))
        if not batches:
            return
        mine = eng.cs_lookup_plan_counts(batches)                 # int64 [n_batches, W] (host): my active rows per owner
        if W > 1:
            loc = torch.from_numpy(mine)
            parts = [torch.empty_like(loc) for _ in range(W)]
            if dist.get_backend(self.group) == "gloo":
                dist.all_gather(parts, loc, group=self.group)
            else:
                dev = batches[0][0].device
                parts_d = [p.to(dev) for p in parts]
                dist.all_gather(parts_d, loc.to(dev), group=self.group)
                parts = [p.cpu() for p in parts_d]
            allc = torch.stack(parts).numpy()                     # [requester, batch, owner]
        else:
            allc = mine[None]
        for k, (u, p, n) in enumerate(batches):
            key = (u.data_ptr(), int(u.numel()))
            self._lookup_plan[key] = allc[:, k, :]
            # a plan entry belongs to the three tensor OBJECTS it was computed from (kept alive here, so their addresses cannot
            # be recycled while the plan stands) at their current versions: another tensor at the same address, or the same
            # tensors rewritten in place, take the synchronising path instead of stale split sizes
            self._lookup_owner[key] = ((u, p, n), (u._version, p._version, n._version))

    def _planned(self, users, pos=None, neg=None):
        """The plan key of this batch if the plan holds an entry made from these very tensors, else None."""
        key = (users.data_ptr(), int(users.numel()))
        own = self._lookup_owner.get(key)
        if own is None:
            return None
        (u, p, n), vers = own
        if u is not users or u._version != vers[0] or (pos is not None and (p is not pos or p._version != vers[1])) \
                or (neg is not None and (n is not neg or n._version != vers[2])):
            return None
        return key

    def _lookup_counts(self, users, acts):
        key = self._planned(users)
        c = self._lookup_plan.get(key) if key is not None else None
        if c is None:
            self.lookup_syncs += 1
            c = self.engine.cs_lookup_counts(acts).cpu().numpy()  # device -> host: the step waits for its plan
        return c

    def _lookup_exchange(self, users, acts):
        """Looked-up rows of the row-sharded constant tables: owners pack, all_to_all, requesters unpack."""
        eng, q = self.engine, self.rank
        sizes = self._lookup_sizes(users, acts)                   # host int64 [2W]: bytes to requester r | bytes from owner o
        W = self.world
        in_splits, out_splits = list(sizes[:W]), list(sizes[W:2 * W])
        send = eng.cs_lookup_pack(acts)
        recv = eng.cs_lookup_recv(sum(out_splits))
        self._all_to_all_v(recv, send[:sum(in_splits)], out_splits, in_splits, sizes=sizes)
        self.xgmi_bytes["all_to_all_lookup"] = sum(in_splits) - in_splits[q]
        eng.cs_lookup_unpack(recv)

    def _lookup_sizes(self, users, acts, peek=False):
        """The exchange's split sizes in bytes as a host int64 array (kept per planned batch: its address is an argument of
        the step's program)."""
        key = self._planned(users)
        hit = self.__dict__.setdefault("_lookup_size_arrays", {}).get(key) if key is not None else None
        if hit is not None or peek:
            return hit
        counts = self._lookup_counts(users, acts)                 # [requester][owner] rows
        rb, q, W = self.engine.lookup_row_bytes, self.rank, self.world
        arr = (ctypes.c_int64 * (2 * W))(*([int(counts[r][q]) * rb for r in range(W)] + [int(counts[q][o]) * rb for o in range(W)]))
        if key is not None and key in self._lookup_plan:
            self._lookup_size_arrays[key] = arr
        return arr

    def _all_reduce_async(self, t):
        comm = self._native_comm()
        if comm is not None:
            program.sync(self._comm_stream, torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                _lib.check(_lib.load().elimrec_comm_all_reduce_f32(comm, t.data_ptr(), t.numel(), ops._stream()), "comm_all_reduce")
            return self._OnStream(self._comm_stream)
        if not self._staged(t):
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        t.copy_(host)
        return self._Done()

    def step(self, users, pos, neg, loss=None, publish=False):
        """One training step on this rank's triplets; returns the local loss (0-dim tensor). loss: the device tensor the
        step's loss goes to (default: the next slot of the engine's loss ring) -- plugin.py hands the slot out when the
        caller asks for the loss and runs the step when the caller's optimizer steps. publish: the launch that sums the loss
        also stores it into coherent host memory (engine.loss_publisher(), sequence number engine.last_pub_seq): a caller that
        reads every step's loss on the host (main.py:102 of the reference) waits for that launch, not for the step's end."""
        eng = self.engine
        eng.publish_loss = bool(publish) and not self.multi and eng._fused_head_ok()
        eng.last_pub_seq = None
        if loss is not None:
            self.engine._loss_given = loss
        try:
            if self._hip_engine and torch.is_grad_enabled():
                with torch.no_grad():
                    return self._step(users, pos, neg)
            return self._step(users, pos, neg)
        finally:
            if eng.publish_loss:
                eng.last_pub_seq = eng.loss_publisher().issued()
                eng.publish_loss = False
            if loss is not None:
                self.engine._loss_given = None

    # ---- the step as one host call (program.py / csrc/program.hip): one rank, after a few ordinary steps
    NATIVE_WARM = 1          # ordinary steps before tracing starts (buffers of this batch size allocated, the loss ring in place):
                             # steps 2-5 are traced, the sixth step of a run is the first one issued from C

    def _native_state(self):
        st = self.__dict__.get("_native")
        if st is None:
            import os
            st = self._native = dict(on=os.environ.get("ELIMREC_NATIVE_STEP", "1") != "0", traces={}, programs={}, failed=None,
                                     steps=0, native_steps=0, checks=0,
                                     # every N-th step of a program is issued the ordinary way under a tracer and compared with it
                                     check_every=int(os.environ.get("ELIMREC_PROGRAM_CHECK", "0") or 0))
        return st

    def _native_eligible(self, users, pos, neg):
        eng = self.engine
        if not (self._hip_engine and not self.profile_kernels and eng.kernel_events is None and not eng.keep_grad
                and not getattr(eng, "word_train", False)        # (its gradient path mixes torch ops into the step)
                and eng.model.mm_fusion_mode == "concat" and eng.model._use_replay
                and all(t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() for t in (users, pos, neg))
                and users.numel() == pos.numel() == neg.numel()):
            return False
        if not self.multi:
            return True
        # several ranks (or the multi-rank path on one): the exchanges must be the library's own RCCL calls, the received
        # columns turned into rows by a kernel (fused head or compact constants), the lookup's split sizes planned ahead
        if self._native_comm() is None or not (eng._fused_head_ok() or eng.lookup):
            return False
        return not self.lookup or self._planned(users, pos, neg) is not None

    def _step(self, users, pos, neg):
        st = self._native_state()
        if not st["on"] or st["failed"] or not self._native_eligible(users, pos, neg):
            return self._step_python(users, pos, neg)
        eng = self.engine
        B = int(users.numel())
        g = self.opt.param_groups[0]
        # everything a step's launches take by VALUE besides the per-step patches: a program is frozen on them
        key = (B, tuple(eng.model._block_weights()), g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"],
               eng.prestaged_ok(users, pos, neg) and eng.ahead_safe(),   # (... and the planner's place in the second stream's order)
               bool(getattr(eng, "publish_loss", False)))                # (... and which launch sums the loss)
        progs = st["programs"].get(key)
        if progs is not None and progs[eng.cur] is not None:
            if st["check_every"] and (st["native_steps"] + st["checks"] + 1) % st["check_every"] == 0:
                return self._step_checked(progs[eng.cur], st, key, users, pos, neg)
            return self._step_native(progs[eng.cur], users, pos, neg, B)
        st["steps"] += 1
        if st["steps"] <= self.NATIVE_WARM or eng._loss_ring is None:
            return self._step_python(users, pos, neg)
        # trace this step (launched from Python, every call recorded) -- two traces per buffer parity make a program
        known = dict(users=users.data_ptr(), pos=pos.data_ptr(), neg=neg.data_ptr(),
                     loss=eng._peek_loss_slot(), step=eng.step_count + 1)
        if self.multi and self.lookup:
            known["sizes"] = ctypes.addressof(self._lookup_sizes(users, None))
        parity = eng.cur
        m = eng.model
        m._use_replay = False
        try:
            with program.StepTracer() as tr:
                loss = self._step_python(users, pos, neg)
        finally:
            m._use_replay = True
        got = st["traces"].setdefault(key, {0: [], 1: []})[parity]
        got.append((tr.items, known))
        if len(got) >= 2 and progs is None:
            progs = st["programs"][key] = [None, None]
        if len(got) >= 2:
            try:
                (ta, ka), (tb, kb) = got[-2], got[-1]
                if any(ka[k] == kb[k] for k in ka):
                    raise ValueError("a per-step value did not change between the two traced steps")
                items, varying = program.diff_traces(ta, tb, ka, kb)
                missing = set(ka) - set(varying.values())
                if getattr(eng, "_tail_base", 0):
                    missing.discard("step")          # the step count travels in the job array native_prologue refreshes
                if missing:
                    raise ValueError("per-step values %s appear in no call" % sorted(missing))
                st["programs"][key][parity] = program.StepProgram(items, varying, keep=(ta, tb), system_scope_events=self.multi)
            except (ValueError, KeyError, TypeError) as e:
                # two traces of different STRUCTURE (one of them followed a step of another batch size and carries the extra
                # stream hand-over of a buffer-set switch): the newer one is kept and paired with the next; anything else, or
                # eight such pairs in a row, and the ordinary path stays -- the reason is kept for inspection
                st["mismatch"] = st.get("mismatch", 0) + 1
                if "traces differ in" not in str(e) or st["mismatch"] >= 8:
                    st["failed"] = str(e)
        return loss

    def _step_checked(self, prog, st, key, users, pos, neg):
        """ELIMREC_PROGRAM_CHECK=N: this step launch by launch under a tracer, compared call by call with the program that would
        have issued it. A difference drops the programs of this batch size (the ordinary path carries on) and keeps the reason."""
        eng = self.engine
        m = eng.model
        B = int(users.numel())
        if B not in eng._bufs or getattr(eng, "_ws_new_seen", None) != getattr(m, "_ws_new", 0):
            return self._step_native(prog, users, pos, neg, B)       # fresh buffers add a stream hand-over: not a step to compare
        known = dict(users=users.data_ptr(), pos=pos.data_ptr(), neg=neg.data_ptr(), loss=eng._peek_loss_slot(), step=eng.step_count + 1)
        if self.multi and self.lookup:
            known["sizes"] = ctypes.addressof(self._lookup_sizes(users, None))
        m._use_replay = False
        try:
            with program.StepTracer() as tr:
                loss = self._step_python(users, pos, neg)
        finally:
            m._use_replay = True
        st["checks"] += 1
        what = prog.verify(tr.items, known)
        if what is not None:
            st["failed"] = "program check: " + what
            st["programs"].pop(key, None)
        return loss

    def _step_native(self, prog, users, pos, neg, B):
        eng = self.engine
        m = eng.model
        eng._workspace(B)                                     # this step's set of batch buffers (they alternate; an epoch's ragged
                                                              # last batch switches the size too)
        loss = eng._next_loss_slot()
        eng.native_prologue()
        eng._last_plan_cur = eng.cur                          # (the program's planner writes this buffer set: ahead_safe)
        values = dict(users=users.data_ptr(), pos=pos.data_ptr(), neg=neg.data_ptr(), loss=loss.data_ptr(), step=eng.step_count + 1)
        if self.multi and self.lookup:
            sizes = self._lookup_sizes(users, None)
            values["sizes"] = ctypes.addressof(sizes)
            self.xgmi_bytes["all_to_all_lookup"] = sum(sizes[:self.world]) - sizes[self.rank]
        prog.run(values)
        eng.native_epilogue(3 * B)
        self._native["native_steps"] += 1
        return loss

    def _step_python(self, users, pos, neg):
        return self._backward_python(self._forward_python(users, pos, neg, whole=True))

    def forward_only(self, users, pos, neg, loss=None):
        """The forward half of a step, launch by launch: tables propagated, head and cosine-BPR rows at the batch's rows, the
        loss summed at once. Returns the context backward_only() continues from (plugin.py: a caller that reads the loss, the
        cached tables or a gradient between bpr_loss() and optimizer.step())."""
        if loss is not None:
            self.engine._loss_given = loss
        try:
            with torch.no_grad():
                return self._forward_python(users, pos, neg, whole=False)
        finally:
            self.engine._loss_given = None

    def backward_only(self, ctx, grads_only=False, scale=None):
        """The backward half behind forward_only(). grads_only: stop before the optimizer -- the last adjoint hop writes the
        gradient table instead of carrying the Adam step, nothing is updated (engine.grad, engine._grads hold the gradients).
        scale: device fp32[1], d(total) / d(this loss) (default 1 / world)."""
        with torch.no_grad():
            return self._backward_python(ctx, grads_only=grads_only, scale=scale)

    def _forward_python(self, users, pos, neg, whole):
        eng, W, ph = self.engine, self.world, self._ph
        if self.profile_kernels and getattr(eng, "kernel_events", None) is None:
            eng.kernel_events = self._events
        # one rank: the hops go to the GPU before the plan's host work (they do not need it; the plan runs on a second stream)
        early = self._hip_engine and eng.cs_fork()
        if early:
            ph["cs_forward_hops"]()
        act = ph["cs_plan"](users, pos, neg)                       # int32 [R]: sorted unique node ids, negative padding
        h_ids = None
        if self.multi:
            # the id exchange runs on the exchange stream under the forward hops, which do not need it
            acts = self._like("acts", act, W)
            h_ids = self._all_gather(acts.view(-1), act, after=eng.plan_stream() if self._hip_engine else None)
            self.xgmi_bytes["all_gather"] = act.numel() * 4 * (W - 1)
        else:
            acts = act.view(1, -1)
        if not early:
            ph["cs_forward_hops"]()                                # hops 1..L-1 of my column slice: no communication
        lookup_early = False
        if (self.multi and self.lookup and h_ids is not None and isinstance(h_ids, self._OnStream) and self._native_comm() is not None):
            # the row exchange of the constants needs the gathered ids and nothing of the graph: on the exchange stream, right
            # behind the id exchange and UNDER the forward hops (pack, all_to_all_v, unpack); the head joins it
            with torch.cuda.stream(self._comm_stream):
                self._lookup_exchange(users, acts)
                lookup_rec = program.record(self._comm_stream)
            lookup_early = True
        if self.multi and self._hip_engine:
            eng.cs_gathered_ids(acts, h_ids)                       # second stream, behind the exchanges: the adjoint's source bits
        if self.multi and self._hip_engine and not eng._long_wanted_only():
            eng.cs_forward_long()                                  # the split rows of hop L need neither ids nor plan: ahead of the waits
        if h_ids is not None:
            h_ids.wait()
            if (self._hip_engine and isinstance(h_ids, self._OnStream) and h_ids.recorded is not None
                    and h_ids.stream is eng.plan_stream()):
                eng.plan_joined()                                  # the id exchange followed the plan on ITS stream: one wait covers both
        send = ph["cs_forward_rows"](acts)                         # [W, R, 2*dl]: layer means of the peers' rows, my columns
        if self.multi:
            if self.lookup and not lookup_early:
                self._lookup_exchange(users, acts)                 # S_m / c rows of MY active rows from their owners
            recv = self._like("recv_f", send)
            self._all_to_all(recv, send)
            self.xgmi_bytes["all_to_all_fwd"] = send[0].numel() * 4 * (W - 1)
            if lookup_early:
                program.wait(torch.cuda.current_stream(), lookup_rec)             # the looked-up rows: the head reads them
        else:
            recv = send
        if self._hip_engine:
            assert getattr(eng, "_loss_late", None) is None, "a step was abandoned between its head and its last hop"
            eng._step_in_flight = bool(whole)                       # a whole step: cs_head may leave work to its later launches
        try:
            loss = ph["cs_head"](recv)                             # my rows, every rank's columns -> loss, head backward
        finally:
            if self._hip_engine:
                eng._step_in_flight = False
        return dict(loss=loss, acts=acts)

    def _backward_python(self, ctx, grads_only=False, scale=None):
        eng, W, ph = self.engine, self.world, self._ph
        loss, acts = ctx["loss"], ctx["acts"]
        if self._scale is None:
            self._scale = torch.full((1,), 1.0 / W, dtype=torch.float32, device=loss.device)
        if self._hip_engine:
            eng.grads_only = bool(grads_only)
        try:
            send2, wgrads = ph["cs_backward_local"](self._scale if scale is None else scale)   # [W, R, 2*dl]: my rows, the peers' columns
            h_w = None
            late = self.multi and self._hip_engine and eng.wgrads_deferred()
            if self.multi:
                recv2 = self._like("recv_b", send2)
                self._all_to_all(recv2, send2)
                self.xgmi_bytes["all_to_all_bwd"] = send2[0].numel() * 4 * (W - 1)
                self.xgmi_bytes["all_reduce"] = wgrads.numel() * 4
                if not late:
                    # the projection-weight gradients are needed by the optimizer step only: reduced under the adjoint hops
                    h_w = self._all_reduce_async(wgrads)
            else:
                recv2 = send2
            if late and W == 1:
                # a group of one rank (the multi-rank step measured on one GPU): the sums are global as they are -- no all-reduce,
                # and the projection weights' optimizer spans ride in the last hop's launch as the one-rank step's do
                ph["cs_backward_hops"](recv2, acts, None, True)
            elif late:
                # the weight gradients are FINISHED by the adjoint's first two hop launches (behind their tiles); they are reduced
                # under the last hop, and the projection weights' optimizer spans follow in a launch of their own (cs_update)
                ph["cs_backward_hops"](recv2, acts, None, lambda: self._all_reduce_async(wgrads))
                h_w = eng.wgrads_handle
            elif h_w is not None and self._hip_engine:
                ph["cs_backward_hops"](recv2, acts, h_w)
            else:
                ph["cs_backward_hops"](recv2, acts)
            if h_w is not None:
                h_w.wait()
            if not grads_only:
                ph["cs_update"]()
        finally:
            if self._hip_engine:
                eng.grads_only = False
        return loss

    def global_loss(self, loss):
        if self.world > 1:
            loss = loss.clone()
            if self._staged(loss):
                host = loss.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                loss.copy_(host)
            else:
                dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
            loss /= self.world
        return loss

    def kernel_time_ms(self, name=None):
        total, launches = 0.0, 0
        for e0, e1, n in self._events:
            total += e0.elapsed_time(e1)
            launches += n
        return total, launches


class ColumnShardEngine(object):
    """The cs_* interface on the HIP kernels, around an EliMRec model (which keeps the plan, the folded constants, the
    head kernels' workspace, the projection weights and the cached tables predict() reads)."""

    def __init__(self, model, group=None, feature_shard=None, feature_dtype=None):
        """feature_shard: "replicated" (default: every rank holds the folded constants S_m / c of all N rows) or "row" (config
        key --feature_shard): rank o holds the rows of its 1/world of the users and of the items only, and a step fetches
        the rows of its active nodes from their owners with an all_to_all (elimrec_amd/lookup.py) -- the north star's row
        shards + all-to-all index lookup, for the tables that are N x sum(D_m) and only read. Bitwise the replicated
        result in fp32. feature_dtype: "f32" | "f16" | "bf16" storage of those constants (--feature_dtype; BASELINE.json
        configs[4] says fp16): rows are widened to fp32 when they are looked up, all arithmetic stays fp32."""
        cfg = model.config
        if feature_shard is None:
            feature_shard = str(cfg["feature_shard"]) if "feature_shard" in cfg else "replicated"
        if feature_dtype is None:
            feature_dtype = str(cfg["feature_dtype"]) if "feature_dtype" in cfg else "f32"
        if feature_shard not in ("replicated", "row"):
            raise ValueError("feature_shard must be 'replicated' or 'row' (got %r)" % (feature_shard,))
        if feature_dtype not in ("f32", "f16", "bf16"):
            raise ValueError("feature_dtype must be 'f32', 'f16' or 'bf16' (got %r)" % (feature_dtype,))
        self.lean = bool(getattr(model, "_lean", False))
        if self.lean:
            feature_shard = "row"        # lean tables: the constants exist as this rank's rows only, built by the distributed fold
        self.feature_shard, self.feature_dtype = feature_shard, feature_dtype
        # the lookup path (compact rows of the constants per step) serves the row shards and the 16-bit storage alike
        self.lookup = feature_shard == "row" or feature_dtype != "f32" or bool(getattr(model, "_wide", False))
        # ... and only row shards exchange rows between ranks; otherwise every rank widens / gathers from its own full copy
        self.lookup_exchange = feature_shard == "row"
        if "table_dtype" in cfg and str(cfg["table_dtype"]) != "f32":
            raise ValueError("--table_dtype=%s: the graph tables are fp32 (bf16 storage of the layer tables was removed in round 4: "
                             "it saved 9 %% of the table bytes and was slower, DESIGN.md section 7); 16-bit storage exists for the "
                             "feature constants: --feature_dtype=f16|bf16" % cfg["table_dtype"])
        if not getattr(model, "_lazy", False):
            raise ValueError("the column-sharded engine needs the folded propagation with batch head rows "
                             "(bipartite adjacency: adj_type pre/plain/gcmc; layer_num >= 2; --head_rows=batch)")
        self.model, self.group = model, group
        self.kernel_events = None
        self.world = None

    loss_ring_len = LOSS_RING      # the loss tensor a step returns is overwritten that many steps later
    materialize_rows = 1 << 18     # rows per owner and all_to_all when the cached tables are materialised row-sharded
    mask = property(lambda self: self._masks[self.cur])

    # ------------------------------------------------------------------ set-up
    def cs_setup(self, world, rank, optimizer):
        m = self.model
        dev = m._require_gpu()
        d, N = m.latent_dim, m.num_users + m.num_items
        if d % world != 0 or (d // world) % 4 != 0:
            raise ValueError("recdim %d cannot be split into %d column slices of a multiple of 4" % (d, world))
        self.world, self.rank, self.opt = world, rank, optimizer
        import os
        self.multi = world > 1 or os.environ.get("ELIMREC_SHARD_MULTI", "0") == "1"
        self._forked = False
        self.dl, self.col0 = d // world, rank * (d // world)
        self.ns, self.w = slab.choose_slabs(self.dl, N)
        self.gs = slab.choose_groups(self.ns)
        # an adjacency with a diagonal (adj_type norm / mean + I): the graph carries the E_u-borne and the E_i-borne part side by
        # side in WIDE tables of 2 dl columns (csrc/wide.hip); `hns` / `hgs` = slabs / slab groups of the tables the hops run on
        self.wide = bool(getattr(m, "_wide", False))
        if self.wide and world > 1 and self.feature_shard != "row":
            raise ValueError("an adjacency with a diagonal on several ranks needs --feature_shard=row")
        self.hns = 2 * self.ns if self.wide else self.ns
        self.hgs = slab.choose_groups(self.hns) if self.wide else self.gs
        adj = m._scipy_adj()
        ipw = 64 // max(1, (self.hns // self.hgs) * (self.w // 4))   # lane groups per wave of this geometry
        # launch form of a hop (Tiktok shape, us per hop): the one-launch form over wave tiles (whole table, 64-neighbour tiles:
        # 31 against 36 for hop + fix-up kernels; an 8-column shard, 32-neighbour tiles: 20.3 against 21.6; 16 columns: 19.5
        # against 21.7), which also lets the last adjoint hop carry the Adam step
        import os
        tiered = os.environ.get("ELIMREC_SLAB_TIERED", "1") != "0"
        kw = dict(side_split=m.num_users, ipw=ipw, tiered=tiered,
                  threshold=(64 if world == 1 else 32) if tiered else slab.LONG_ROW_THRESHOLD)
        if tiered and getattr(m, "_plan_build", "host") == "device":
            # the wave-tile plan from a device CSR (csrc/plan.hip: the host build's arrays bit for bit, without its sorts over all
            # non-zeros on the host): the model's device copy of the adjacency, or an upload of the host matrix (lean tables)
            def device_plan(name, mat):
                if hasattr(m, name + "_rowptr"):
                    rp, cl, vl = getattr(m, name + "_rowptr").to(dev).long(), getattr(m, name + "_col").to(dev), getattr(m, name + "_val").to(dev)
                else:
                    mat = mat.tocsr()
                    mat.sort_indices()
                    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(dev)
                    rp, cl, vl = up(mat.indptr, np.int64), up(mat.indices, np.int32), up(mat.data, np.float32)
                return slab.SellPlan.on_device(rp, cl, vl, N, threshold=kw["threshold"], side_split=kw["side_split"], ipw=ipw)
            self.plan = device_plan("adj", adj)
            self.planT = self.plan if m._adj_symmetric else device_plan("adjT", adj.T)
        else:
            self.plan = slab.SellPlan(adj, dev, **kw)
            self.planT = self.plan if m._adj_symmetric else slab.SellPlan(adj.T.tocsr(), dev, **kw)
        # a column slice beyond the Infinity Cache (configs[3] on one GPU; up to 2^27 non-zeros unless forced): the user rows of every WHOLE hop
        # by the window sweep (csrc/sweep.hip), the item rows by a tile plan of their own; the masked hop, the split-rows hop of
        # the last forward layer and the batch-row kernels keep the whole plan. The hops then carry no tails (Adam, weight
        # gradients): those run as launches of their own, a few tens of microseconds beside hops of a millisecond.
        deg = np.diff(adj.tocsr().indptr)
        self._split_share = float(deg[deg > kw["threshold"]].sum()) / max(float(deg.sum()), 1.0)      # non-zeros in split rows
        self.sweep = bool(tiered and not self.wide and self.w in (16, 32) and slab.sweep_tiles_xcds(self.ns)
                          and slab.sweep_wanted(N, self.dl, adj.nnz))
        if self.sweep:
            self.plan.sweep = slab.SweepPlan(self.plan, adj, m.num_users, dev, kw["threshold"], ipw)
            if self.planT is not self.plan:
                self.planT.sweep = slab.SweepPlan(self.planT, adj.T.tocsr(), m.num_users, dev, kw["threshold"], ipw)
        tab = lambda: slab.SlabTable(N, self.ns, self.w, dev)
        L = m.n_layers
        self.master = [tab(), tab()]
        self.cur = 0
        self.long_tab = torch.empty(self.ns * max(self.plan.n_long, 1) * self.w, dtype=torch.float32, device=dev)
        self.xL = None                                            # full hop-L table, only when predict() needs it
        self.grad = tab()
        self.m1 = torch.zeros_like(self.grad.data)
        self.m2 = torch.zeros_like(self.grad.data)
        if self.wide:
            wtab = lambda: slab.SlabTable(N, self.hns, self.w, dev)
            self.x0w = wtab()                                     # layer 0: [E_u | 0] on user rows, [0 | E_i] on item rows
            self.layers = [None] + [wtab() for _ in range(L)]     # X^1 .. X^L, all of them whole (no row-list last hop)
            self.srcW, self.tmp = wtab(), [wtab(), wtab()]
            half = self.ns * N * self.w                           # the adjoint source [H | G]: left / right halves as narrow tables
            self.srcA = slab.SlabTable(N, self.ns, self.w, dev, data=self.srcW.data[:half])
            self.srcB = slab.SlabTable(N, self.ns, self.w, dev, data=self.srcW.data[half:])
        else:
            self.layers = [None] + [tab() for _ in range(L - 1)]
            self.srcA, self.srcB, self.tmp = tab(), tab(), [tab(), tab()]
        # row bitmap of the adjoint's sources (= the planner's key bitmap on one rank): one per step parity, like the batch
        # buffers -- the planner of step t + 1 may run while step t's adjoint hops still read theirs (cs_plan)
        self._masks = [torch.zeros((N + 31) // 32 + 2, dtype=torch.int32, device=dev) for _ in range(2)]
        self.step_count = 0
        self._aux = None
        self._aux_pending = False
        self._adam_in_hop = False
        self._out0_src = self._nar_src = self._peer_src = None
        self._pairs = {}
        self._split_head = os.environ.get("ELIMREC_HEAD_SPLIT", "1") != "0"     # feature blocks of the head beside the forward hops
        self._head_split = False
        # ... and the layer means of the active rows evaluated by the head's main-stream launch itself (no rows launch): one rank
        # that owns all 64 columns, fp32 tables, the two-launch head
        self._rows_in_head_on = os.environ.get("ELIMREC_ROWS_IN_HEAD", "1") != "0"
        self._rows_in_head = False
        self.send_b = None
        self._loss_ring, self._loss_at = None, 0
        self._bits_ready = False
        self.keep_grad = False
        self._fused = None
        self._tail_plan = None
        self._bufs = {}
        self._x0_fwd = None
        # where the folded constants come from: "model" = EliMRec._fold_constants (every rank computes all N rows, then keeps
        # its own); "sharded" = the distributed fold below (no rank ever holds more than a column slice or its own rows --
        # what BASELINE.json configs[4] needs). Default: sharded for row shards over several ranks of a real process group.
        fold_mode = os.environ.get("ELIMREC_FOLD", "")
        if self.lean or self.wide:
            fold_mode = "sharded"            # (wide: the model's fold kernels are the bipartite ones)
        if fold_mode not in ("model", "sharded"):
            fold_mode = "sharded" if (self.feature_shard == "row" and world > 1 and dist.is_available() and dist.is_initialized()
                                      and dist.get_world_size(self.group) == world) else "model"
        self.fold_mode = fold_mode
        m._skip_fold = fold_mode == "sharded"
        ws = m._workspace(1)
        self.fshard = None
        if self.lookup:
            owners = RowOwnerMap(m.num_users, m.num_items, world if self.feature_shard == "row" else 1)
            frank = rank if self.feature_shard == "row" else 0
            if fold_mode == "sharded":
                tabs, c_loc = self._fold_sharded(owners, frank)
                self.fshard = FeatureShard.from_local(owners, frank, tabs, c_loc, dtype=self.feature_dtype)
            else:
                fold = ws["fold"]
                self.fshard = FeatureShard(owners, frank, [fold[k] for k in m._mods], fold["c"], dtype=self.feature_dtype, device=dev)
                if self.feature_shard == "row" and world > 1:
                    ws["fold"] = None                # the full tables are gone: every consumer goes through the shards
            self.lookup_row_bytes = self.fshard.row_bytes
            self._lookup_bufs = {}
        # ONE rank holding every row of 16-bit constants: the fused head reads them where they lie and widens in registers
        # (elimrec_head_fwd_fused_src16) -- no widening pass over the batch's rows in front of the head, and the step keeps the
        # shape of the fp32 one (feature blocks beside the forward hops, rows evaluated by the head's launch); the head's feature
        # launch leaves the widened rows behind for the backward half's weight gradients
        self._direct16 = bool(self.lookup and self.fshard is not None and self.fshard.code != 0 and not self.multi
                              and self.fshard.owners.world == 1 and not self.wide
                              and os.environ.get("ELIMREC_DIRECT16", "1") != "0")
        # data set "tiktok": word_embedding keeps receiving the gradient the reference's retained graph gives it (default), or stays
        # frozen (--word_embedding=frozen)
        self.word_train = False
        if hasattr(m, "word_embedding") and m.dataset_name == "tiktok" and "feature_modalities" not in m.config:
            mode = str(m.config["word_embedding"]) if "word_embedding" in m.config else "train"
            if mode not in ("train", "frozen"):
                raise ValueError("word_embedding must be train or frozen (got %r)" % mode)
            if mode == "train" and (world > 1 or self.lean or self.wide or self.multi):
                from .logger import Logger
                Logger.info("word_embedding stays frozen: its gradient (the reference's retained graph) is computed on one rank with "
                            "regular tables only")
            elif mode == "train":
                self._word_setup()
                self.word_train = True
        self.load_from_model()
        m._slab_engine = self
        m._regions = {}                  # recorded launches of an engine attached earlier hold ITS tables' addresses
        # the embeddings are updated here, not by the caller's optimizer; the projection weights go through it
        self._tail = [(n, p) for n, p in m.named_parameters() if not n.startswith(("embedding_user.", "embedding_item."))]
        return ws

    @torch.no_grad()
    def load_from_model(self):
        """Master copy <- the model's embedding parameters (start-up, load_state_dict)."""
        m = self.model
        ws = m._workspace(m._ws_key[1] if m._ws_key else 1, parity=m._ws_key[3] if m._ws_key else 0)
        if self.lean:       # host parameters: upload my column slice (users, then items) and lay it out slab-major
            dev, U = m._device(), m.num_users
            x0 = torch.empty(U + m.num_items, self.dl, dtype=torch.float32, device=dev)
            x0[:U].copy_(m.embedding_user.weight.detach()[:, self.col0:self.col0 + self.dl])
            x0[U:].copy_(m.embedding_item.weight.detach()[:, self.col0:self.col0 + self.dl])
            self.master[self.cur].from_rows(x0, col0=0)
            del x0
        else:
            self.master[self.cur].from_rows(ws["X0d"], col0=self.col0)
        ws["snap"].copy_(ws["flat_param"][ws["tail_off"]:])        # from here on the snapshot is refreshed by cs_update

    @torch.no_grad()
    def sync_to_model(self):
        """The model's embedding parameters <- the master copy (before a checkpoint; all ranks must call it)."""
        ctl = getattr(self.model, "_plugin", None)
        if ctl is not None and ctl.engine is self:
            ctl.master_newer = False          # (first: the write-back below touches the parameters it is about to refresh)
        self._sync_to_model()
        if ctl is not None and ctl.engine is self:
            ctl.synced()

    @torch.no_grad()
    def _sync_to_model(self):
        m = self.model
        if self.lean:       # every rank's column slice into every rank's host parameters (checkpoint time only)
            U = m.num_users
            loc = self.master[self.cur].dense()
            parts = [loc] if self.world == 1 else _all_gather_parts(loc, self.world, self.group)
            eu, ei = m.embedding_user.weight.data, m.embedding_item.weight.data
            for q, part in enumerate(parts):
                host = part.cpu()
                eu[:, q * self.dl:(q + 1) * self.dl] = host[:U]
                ei[:, q * self.dl:(q + 1) * self.dl] = host[U:]
            return
        x0d = m._ws["X0d"]
        if self.world == 1:
            self.master[self.cur].to_rows(x0d, col0=0)
            return
        loc = self.master[self.cur].dense()
        x0d.copy_(torch.cat(_all_gather_parts(loc, self.world, self.group), dim=1))

    def _workspace(self, B):
        m = self.model
        if B not in self._bufs:                   # first use of this batch size: BOTH sets of its batch buffers now (cs_plan)
            m._workspace(B, 3 * B, parity=1 - self.cur)
        ws = m._workspace(B, 3 * B, parity=self.cur)
        dev, d, R, W = m._device(), m.latent_dim, 3 * B, self.world
        N = m.num_users + m.num_items
        if not self.lookup and (getattr(self, "narrow_x", None) is None or ws["Narrow"].data_ptr() != self.narrow_x.data_ptr()):
            # (by-node copy of the shared part: only the path without compact rows reads it)
            self.narrow_x = torch.zeros(N + 1, d, dtype=torch.float32, device=dev)   # spare row: padded slots land there
            ws["Narrow"] = self.narrow_x[:N]
            m._regions = {}                                       # recorded regions hold the old buffer's address
        bufs = self._bufs.get(B)
        if bufs is None:                                          # per batch size (an epoch ends with a ragged batch)
            bufs = self._bufs[B] = dict(send_f=torch.empty(W, R, 2 * self.dl, dtype=torch.float32, device=dev) if self.multi else None,
                                        send_b=torch.zeros(W, R, 2 * self.dl, dtype=torch.float32, device=dev) if (self.multi or self.wide) else None)
        self.send_f, self.send_b = bufs["send_f"], bufs["send_b"]
        if "nar_act" not in bufs:
            bufs["nar_act"] = torch.zeros(R, d, dtype=torch.float32, device=dev)       # shared part of Out, compact rows
        self.nar_act = bufs["nar_act"]
        if self.lookup:
            if "s_rows" not in bufs:       # the constants' rows of this batch's active nodes, fp32, in active-row order
                sd = self.fshard.sum_d
                bufs["s_rows"] = torch.zeros(R, sd, dtype=torch.float32, device=dev)
                bufs["c_rows"] = torch.zeros(R, dtype=torch.float32, device=dev)
                bufs["iota"] = torch.arange(R, dtype=torch.int32, device=dev)
            self.s_rows, self.c_rows, self.iota = bufs["s_rows"], bufs["c_rows"], bufs["iota"]
            views, off = {}, 0
            for k, D in zip(m._mods, self.fshard.dims):
                views[k] = self.s_rows[:, off:off + D]
                off += D
            m._fold_rows = dict(S=views, c=self.c_rows, narrow=self.nar_act, R=R)
        return ws

    def _fused_head_ok(self):
        """The one-launch head forward (csrc/head.hip) covers recdim 64, concat fusion and row tiles that fit LDS."""
        m = self.model
        if self._fused is None:
            import os
            dims = [getattr(m, k + "_feat").shape[1] for k in m._mods]
            lds = 4 * (2 * 32 * 65 + sum(32 * (D + 1) for D in dims) + 32 * (m.C + 1) + (1 + m.S) * 32 * 64)
            self._fused = (os.environ.get("ELIMREC_FUSED_HEAD", "1") != "0" and m.latent_dim == 64 and 1 <= m.S <= 3
                           and m.mm_fusion_mode == "concat" and lds <= 158 * 1024)
            self._pack_bwd_off = 0
            if self._fused:
                self._pack = torch.empty(ops.head_pack_floats(dims), dtype=torch.float32, device=m._device())
                self._pack_bwd_off = ops.head_pack_bwd_offset(dims)
        return self._fused

    def _aux_stream(self):
        """Second HIP stream for the two launches of a step that do not depend on the forward hops and are latency-bound
        (one workgroup, or a few): the batch plan and the packing of the head's weights. Issued there, they run UNDER the
        hops instead of in front of / behind them (ELIMREC_AUX_STREAM=0: everything on one stream). One rank only: with
        peers the plan feeds the id exchange at once."""
        if self._aux is None:
            import os
            on = (not self.multi or getattr(self, "multi_aux", False)) and os.environ.get("ELIMREC_AUX_STREAM", "1") != "0"
            self._aux = torch.cuda.Stream() if on else False
        return self._aux or None

    @_once
    def _fuse_adam(self):
        """Adam of the embeddings (this rank's column shard) as the epilogue of the adjoint's last hop: fp32 tables, the
        tiered plan, at least two layers (ELIMREC_FUSE_ADAM=0 keeps the separate optimizer launch; `keep_grad` = True also
        stores the gradient table, for tests that read it)."""
        import os
        return (not self.wide and self.planT.tiered and self.model.n_layers >= 2
                and os.environ.get("ELIMREC_FUSE_ADAM", "1") != "0")

    @_once
    def _fuse_reduce(self):
        """The weight gradients' slab reduce as extra workgroups of the adjoint's first hop: one rank, fp32 tables, the
        tiered plan, 'concat' fusion, and a plain masked hop to carry it (ELIMREC_FUSE_REDUCE=0: its own launch)."""
        import os
        m = self.model
        hops_in_region = m.n_layers - (1 if self._fuse_adam() else 0)
        # (not with the swept form: at the configs[3] shape the masked hop that carries the reduce takes 898 us against 700 + 14 apart)
        return (os.environ.get("ELIMREC_FUSE_REDUCE", "1") != "0" and not self.wide and not self.sweep and self.planT.tiered
                and hops_in_region >= 1 and m.mm_fusion_mode == "concat")

    @_once
    def _long_wanted_only(self):
        """Hop L at the split rows of the batch only (elimrec_slab_hop, seg_only with the wanted-rows bitmap -- several ranks: of every
        rank's batch): graphs whose split rows hold 30 % of the non-zeros or more (every user row of configs[3]; 15 % at the Tiktok
        shape, where the popular items that are split rows are in nearly every batch) and the swept form (ELIMREC_LONG_WANTED=1 / 0
        forces it on / off)."""
        import os
        e = os.environ.get("ELIMREC_LONG_WANTED")
        return (not self.wide and self.planT.tiered and self.plan.tiered
                and ((self.sweep or self._split_share >= 0.3) if e is None else e == "1"))

    @_once
    def _fuse_bwd_w(self):
        """Both phases of the weight gradients behind adjoint hops' tiles: needs two plain hops before the Adam hop."""
        import os
        hops_in_region = self.model.n_layers - (1 if self._fuse_adam() else 0)
        return os.environ.get("ELIMREC_FUSE_BWDW", "1") != "0" and hops_in_region >= 2

    @_once
    def _sources_in_head(self):
        """One rank, recdim 64, packed head weights: the head backward's kernel writes the adjoint sources at the active rows
        (each listed once) and the planner's key bitmap is their row bitmap -- no merge at all (ELIMREC_HEAD_SOURCES=0: the
        merge rides in the weight-gradient launch)."""
        import os
        return (not self.multi and not self.wide and self._fused_head_ok() and bool(self._pack_bwd_off) and self.model.latent_dim == 64
                and os.environ.get("ELIMREC_HEAD_SOURCES", "1") != "0")

    @_once
    def _loss_sum_late(self):
        """The loss rows are summed by an extra workgroup of the Adam hop (ELIMREC_LOSS_LATE=0: by the BPR launch's last workgroup)."""
        import os
        return self._fuse_adam() and os.environ.get("ELIMREC_LOSS_LATE", "1") != "0"

    @_once
    def _split_in_head(self):
        import os
        return (not self.wide and self._fused_head_ok() and bool(self._pack_bwd_off) and self.model.latent_dim == 64
                and os.environ.get("ELIMREC_HEAD_SOURCES", "1") != "0")

    @_once
    def _fuse_merge(self):
        import os
        return (os.environ.get("ELIMREC_FUSE_MERGE", "1") != "0" and not self.wide
                and (self.model.num_users + self.model.num_items) <= (1 << 27))

    def _timed(self, fn, hops):
        ev = self.kernel_events
        if ev is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ev.append((e0, e1, hops))

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def cs_plan(self, users, pos, neg):
        m = self.model
        B = int(users.numel())
        ws = self._workspace(B)
        users, pos, neg = m._index_tensors(users, pos, neg)
        R = 3 * B
        keys = ws["keys"][:R]
        act, seg = ws["active_rows"][:R], ws["seg_info"]
        m._plan_n = R
        m._last_block_weights = m._block_weights()
        self._keys = keys
        err = m._index_err()

        aux = self._aux_stream()
        self._head_split = self._rows_in_head = False
        early_bits = aux is not None and self.planT.tiered and not self.multi      # (several ranks: cs_gathered_ids)
        fresh = aux is not None and getattr(self, "_ws_new_seen", None) != getattr(m, "_ws_new", 0)
        if fresh:
            # a buffer set allocated -- and zero-filled ON THE MAIN STREAM -- a moment ago (first step of this batch size): the
            # second stream behind the main one before the planner writes into it
            program.sync(aux, torch.cuda.current_stream())
            self._ws_new_seen = getattr(m, "_ws_new", 0)
        # the planner ahead of the previous step's end (prestage): one rank, the triplets announced as complete -- and the pass
        # before this one planned into the OTHER set of batch buffers (ahead_safe: a forward_only / gradients-only pass does not
        # flip the sets, and its head and adjoint kernels may still be reading the one this planner is about to write)
        ahead = (aux is not None and self._forked and not fresh and not self.multi and self.prestaged_ok(users, pos, neg)
                 and self.ahead_safe())
        self._last_plan_cur = self.cur

        # the per-line source bits are needed by the first ADJOINT hop only. Issued on the second stream right behind the planner,
        # before the weight packing and the feature blocks: one join per step (the second stream's forward work is then 72 us
        # against the main stream's 74). Behind the forward's join instead -- under the head kernels, with a second join before the
        # adjoint -- measured 0.2834 against 0.2821 ms at the end of round 4 (three pairs of 300 steps) and was removed

        # a large batch (more than 8192 slots: the device-wide planner, feature blocks of tens of microseconds) makes the second
        # stream the forward's critical path: the source bits then follow the forward's join instead (cs_forward_rows) and the
        # adjoint joins them -- a second join for 13 us less in front of the head
        late_bits = early_bits and R > 8192
        self._late_bits = late_bits

        def plan_only():  # node ids of the slots, unique active rows + slot map, padded tail: one launch
            ops.batch_plan(users, pos, neg, m.num_users, m.num_items, keys, act, seg, ws["slot_seg"][:R], ws["plan_ws"], err, PAD_KEY,
                           key_bitmap=self.mask if (early_bits or self._sources_in_head()) else None)

        def bits_only():  # the planner's bitmap of the active rows IS the first adjoint hop's source bitmap (one rank): its per-line
            if early_bits and not late_bits:           # bits (they live in the plan's one scratch area: never ahead of the step before)
                slab.source_bits(self.planT, self.hns, self.w, self.hgs, self.mask)

        def plan():
            plan_only()
            bits_only()

        def pack():       # the head's weights in MFMA fragment order (they changed in the last optimizer step)
            self._head_fused_call(ws, R, phase=1)

        def features():   # ... and the head's feature blocks (constants' rows x projections: nothing of the graph) beside the hops
            self._head_fused_call(ws, R, phase=3)
        self._bits_ready = early_bits
        if aux is None:
            m._region("cs_plan", (m._ws_gen, users.data_ptr(), pos.data_ptr(), neg.data_ptr(), R), plan)
            return act
        if not self._forked:
            program.sync(aux, torch.cuda.current_stream())       # the triplets, and last step's readers of the plan buffers
        fork_rec = getattr(self, "_fork_rec", None) if self._forked else None
        self._forked, self._fork_rec = False, None
        with torch.cuda.stream(aux):
            if ahead:
                m._region("cs_plan_a", (m._ws_gen, users.data_ptr(), pos.data_ptr(), neg.data_ptr(), R), plan_only)
                program.wait(aux, fork_rec)                      # everything else of this stream: behind the previous step
                m._region("cs_bits_a", (m._ws_gen, R, early_bits, late_bits), bits_only)
            else:
                if fork_rec is not None:
                    program.wait(aux, fork_rec)
                m._region("cs_plan", (m._ws_gen, users.data_ptr(), pos.data_ptr(), neg.data_ptr(), R), plan)
            if self._fused_head_ok():
                m._region("cs_pack", (m._ws_gen, R), pack)
                self._head_split = (not self.lookup or self._direct16) and self._split_head
                if self._head_split:
                    m._region("cs_head_features", (m._ws_gen, R), features)
                self._rows_in_head = (self._head_split and self._rows_in_head_on and not self.multi and not self.wide
                                      and self.dl == 64 and self.ns * self.w == 64)
        self._aux_pending = True
        # several ranks: the second stream goes on to the adjoint's source bits once the ids are gathered (cs_gathered_ids); what
        # the forward joins is the plan and the packed weights, recorded here
        self._plan_rec = program.record(aux) if self.multi else None
        self._plan_waited = False
        return act

    def plan_stream(self):
        """The stream cs_plan's launches went to when it is not the caller's (the second stream), else None."""
        return self._aux_stream()

    def plan_joined(self):
        """The caller's stream has waited for something enqueued on the second stream BEHIND this step's plan and packed weights
        (the id exchange's event): cs_forward_rows need not wait for the plan's own event again (each wait on another queue's
        event costs the main queue ~5 us of command-processor latency, satisfied or not)."""
        if getattr(self, "_plan_rec", None) is not None:
            self._plan_rec = None
            self._plan_waited = True

    def wgrads_deferred(self):
        """Several ranks: the weight gradients ride behind the adjoint hops' tiles (both phases) and are all-reduced late."""
        return bool(getattr(self, "defer_wgrads", False)) and self.multi and self._fuse_reduce() and self._fuse_bwd_w()

    @torch.no_grad()
    def cs_gathered_ids(self, acts, handle):
        """Several ranks, second stream: as soon as every rank's active ids are here, the row bitmap of the adjoint's sources
        (all ranks' rows) and the masked hop's per-line source bits -- a whole forward pass before the rows arrive."""
        aux = self._aux_stream()
        m = self.model
        if aux is None or not self.planT.tiered:
            return
        with torch.cuda.stream(aux):
            handle.wait()                                   # the second stream behind the id exchange

            key = (m._ws_gen, acts.data_ptr(), acts.shape[0], acts.shape[1])
            m._region("cs_row_bits", key, lambda: slab.rows_bitmap(acts, m.num_users + m.num_items, self.mask))
            # (the bitmap of every rank's active rows is also the wanted-rows bitmap of hop L's split rows: cs_forward_rows)
            self._rows_rec = program.record(aux) if self._long_wanted_only() else None
            m._region("cs_bits", key, lambda: slab.source_bits(self.planT, self.hns, self.w, self.hgs, self.mask))
        self._bits_ready = True
        self._bits_join = True                              # the adjoint's first hop joins the second stream for them

    def cs_fork(self):
        """One rank with the second stream: order it behind what the main stream holds NOW, so that the caller can enqueue the
        forward hops (which need nothing of the plan) before cs_plan's host work -- the GPU starts on the step at once.
        Returns whether it did (then cs_plan does not fork again)."""
        aux = self._aux_stream()
        if aux is None or self.model._ws is None:
            return False
        self._fork_rec = program.record(torch.cuda.current_stream())     # waited for by the second stream in cs_plan
        self._forked = True
        return True

    def prestage(self, batches):
        """The caller states that these (users, pos, neg) tensors are complete on the device NOW (an epoch's triplets sampled
        ahead: main.py, bench.py). For such a batch the planner of a step does not wait for the step before it: it reads only the
        triplets and writes this step's own set of batch buffers (two sets, alternating), so on the second stream it runs UNDER the
        previous step's backward -- at large batches (device-wide planner: 60-110 us) that takes it off the forward's critical
        path. Batches not announced here keep the conservative order (their triplets may be the product of work still queued on
        the caller's stream)."""
        aux = self._aux_stream()
        self._prestaged = {}
        if aux is None or self.multi:
            return
        program.sync(aux, torch.cuda.current_stream())       # once: whatever produced them is done before any of their planners runs
        self._prestaged = {id(u): (u, p, n) for u, p, n in batches}

    def ahead_safe(self):
        """May this pass's planner write its batch buffers before the previous pass has drained? Only if that pass planned into
        the other buffer set, i.e. the sets have flipped since (cs_update / native_epilogue flip them; forward_only and
        backward_only(grads_only=True) -- a loss read without a step, gradients for a foreign optimizer -- do not)."""
        return getattr(self, "_last_plan_cur", None) != self.cur

    def prestaged_ok(self, users, pos, neg):
        e = getattr(self, "_prestaged", {}).get(id(users))
        return e is not None and e[0] is users and e[1] is pos and e[2] is neg

    @torch.no_grad()
    def cs_forward_hops(self):
        m = self.model
        L = m.n_layers
        x0 = self.master[self.cur]
        self._x0_fwd = x0
        tabs = self._tabs = [x0] + self.layers[1:]
        srcs = self._srcs = [x0] + self.layers[1:]                 # what the hops gather from

        if self.wide:
            tabs = self._tabs = self._srcs = [self.x0w] + self.layers[1:]

            def wide_hops():
                slab.wide_from_master(x0, m.num_users, self.x0w)
                for k in range(1, L + 1):
                    slab.hop(self.plan, tabs[k - 1], tabs[k], gs=self.hgs)
            self._timed(lambda: m._region("cs_fwd_hops%d" % self.cur, (m._ws_gen,), wide_hops), L)
            return

        def hops():
            for k in range(1, L):
                slab.hop(self.plan, srcs[k - 1], tabs[k], gs=self.gs)
        self._timed(lambda: m._region("cs_fwd_hops%d" % self.cur, (m._ws_gen,), hops), L - 1)

    @torch.no_grad()
    def cs_forward_rows(self, acts):
        m = self.model
        ws, L, U = m._ws, m.n_layers, m.num_users
        W, R = acts.shape
        tabs = self._tabs
        self._acts = acts
        late_wait = self._aux_pending                           # the long-rows hop needs nothing of the plan: join after it

        def join_plan():
            rec = getattr(self, "_plan_rec", None)
            if getattr(self, "_plan_waited", False):             # plan_joined(): already behind it
                self._plan_waited = False
            elif rec is not None:                                  # (several ranks: the plan's event, not the stream's tail)
                program.wait(torch.cuda.current_stream(), rec)
                self._plan_rec = None
            else:
                program.sync(torch.cuda.current_stream(), self._aux)
        self._aux_pending = False
        if self.multi:
            counts = None                                        # the gathered lists are padded with negative keys
            out0, narrow, by_node = self.send_f.view(W * R, 2 * self.dl)[:, :self.dl], self.send_f.view(W * R, 2 * self.dl)[:, self.dl:], False
        else:
            counts = ws["seg_info"][0:1]
            if self._fused_head_ok() or self.lookup or self.wide:
                out0, narrow, by_node = ws["OutAct"][:R, :m.latent_dim], self.nar_act, False
            else:
                out0, narrow, by_node = ws["OutAct"][:R, :m.latent_dim], ws["Narrow"], True

        long_done = getattr(self, "_long_done", False)          # cs_forward_long issued it already (several ranks)
        self._long_done = False

        # the swept form (tables beyond the caches: every user row of configs[3] is a split row): layer L only at the split rows
        # of the BATCH -- the planner's bitmap of the active rows says which -- so the join comes first (651 -> 60 us there; at the
        # Tiktok shape the split rows are the popular items, nearly all of them in every batch, and the join ahead of the hop costs more)
        wanted = self.mask if (self._long_wanted_only() and self._bits_ready and not long_done) else None
        joined = False
        if wanted is not None and self.multi:
            rec = getattr(self, "_rows_rec", None)
            if rec is None:
                wanted = None                                    # (no second stream / no bitmap of the gathered ids: every split row)
            else:
                program.wait(torch.cuda.current_stream(), rec)   # the gathered ids' row bitmap (behind the plan on the second stream)
                self._rows_rec = self._plan_rec = None
                if late_wait:
                    late_wait, joined = False, True
        elif wanted is not None and late_wait:
            join_plan()
            late_wait, joined = False, True

        def long_rows():
            if self.plan.n_long and not self.wide and not long_done:
                slab.hop(self.plan, self._srcs[L - 1], self.long_tab, gs=self.gs, seg_only=True, add_mask=wanted)

        def rows():
            if not late_wait:
                long_rows()
            if self.wide:         # all L + 1 wide tables are whole: layer means at the listed rows (padding ids are negative)
                assert not by_node
                slab.wide_rows([t.data for t in tabs], tabs[0].n, self.ns, self.w, acts.reshape(-1), W * R, out0, narrow)
            elif self._rows_in_head:
                pass                  # the head's launch evaluates the rows itself (cs_head: elimrec_head_fwd_fused_rows)
            else:
                slab.rows(self.plan, self.ns, self.w, L, U, [t.data for t in tabs] + [None], self.long_tab, acts, counts, R, W,
                          out0, narrow, by_node)
            if self.lookup and not (self.multi and self.lookup_exchange) and not (self._direct16 and self._fused_head_ok()):
                # this rank holds every row: no exchange, widen in place (16-bit rows under the fused head: read by the head itself)
                self.fshard.unpack(ws["active_rows"][:R], None, self.s_rows, self.c_rows, direct=True)
        if late_wait:
            m._region("cs_fwd_long%d" % self.cur, (m._ws_gen,), long_rows)
            join_plan()                                        # the plan (and the packed weights) from the second stream
            joined = True
        if joined and getattr(self, "_late_bits", False):      # (large batches) the adjoint's source bits: behind the forward's join
            self._late_bits = False
            with torch.cuda.stream(self._aux):
                m._region("cs_late_bits", (m._ws_gen, R), lambda: slab.source_bits(self.planT, self.hns, self.w, self.hgs, self.mask))
            self._bits_join = True
        m._region("cs_fwd_rows%d" % self.cur, (m._ws_gen, acts.data_ptr(), R, W, narrow.data_ptr(), late_wait, self._rows_in_head, wanted is not None), rows)
        return self.send_f if self.multi else None

    @torch.no_grad()
    def cs_forward_long(self):
        """Hop L at the split rows (the part the rows launch cannot evaluate inline) right behind the forward hops: it needs
        nothing of the batch, so with several ranks it runs before the main stream waits for the gathered ids."""
        m = self.model
        if not self.plan.n_long or self.wide:
            return
        L = m.n_layers
        m._region("cs_fwd_long_early%d" % self.cur, (m._ws_gen,),
                  lambda: slab.hop(self.plan, self._srcs[L - 1], self.long_tab, gs=self.gs, seg_only=True))
        self._long_done = True

    def cs_forward(self, acts):
        self.cs_forward_hops()
        return self.cs_forward_rows(acts)

    @torch.no_grad()
    def cs_head(self, recv):
        m = self.model
        ws, d = m._ws, m.latent_dim
        R = m._plan_n
        B = R // 3
        fused = self._fused_head_ok()
        if recv is not None:                                      # [W, R, (out0 | narrow)] -> my rows, all columns
            W = recv.shape[0]
            r = recv.view(W, R, 2, self.dl)
            self._peer_src = None
            if fused and not self._direct16 and self.dl % 4 == 0:
                # the fused head reads the peers' pieces where the exchange left them and writes block 0 of OutAct itself
                self._peer_src = recv.view(W, R, 2 * self.dl)
                self._out0_src = self._nar_src = None
            elif fused:
                # one pass: [R, (out0 | narrow), d]; the fused head reads both halves with a 2d row stride and writes
                # block 0 of OutAct itself
                pair = self._pair(R, d)
                ops.peer_cols_to_rows(recv.view(W, R, 2 * self.dl), pair[:, 0, :], pair[:, 1, :])
                self._out0_src, self._nar_src = pair[:, 0, :], pair[:, 1, :]
            elif self.lookup:
                ops.peer_cols_to_rows(recv.view(W, R, 2 * self.dl), ws["OutAct"][:R, :d], self.nar_act[:R])
                self._out0_src = self._nar_src = None
            else:
                ws["OutAct"][:R, :d].unflatten(1, (W, self.dl)).copy_(r[:, :, 0].permute(1, 0, 2))
                act = ws["active_rows"][:R].long()
                idx = torch.where(act >= 0, act, torch.full_like(act, self.narrow_x.shape[0] - 1))
                self.narrow_x.index_copy_(0, idx, r[:, :, 1].permute(1, 0, 2).reshape(R, d))
                self._out0_src = self._nar_src = None
        else:
            self._out0_src = self._nar_src = self._peer_src = None
        m._slab_fwd = True
        if fused:
            return self._head_forward_fused(ws, R, B)
        m._fwd_head(ws, self._keys, R, B, 0, ws["grad_rows"])
        loss = self._next_loss_slot()
        ops.fixed_order_sum(ws["loss_rows"], loss)
        return loss

    def loss_publisher(self):
        """The host-visible loss ring of this engine (ops.LossPublisher), created on first use."""
        pub = self.__dict__.get("_loss_pub")
        if pub is None:
            pub = self._loss_pub = ops.LossPublisher(64)
        return pub

    def _next_loss_slot(self):
        """The next slot of the loss ring (a caller holding the tensors of earlier steps -- main.py stacks an epoch's losses
        before it copies them to the host -- does not see them change for LOSS_RING steps)."""
        given = getattr(self, "_loss_given", None)
        if given is not None:                 # handed out earlier (trainer.step(..., loss=...)): this step fills it
            self._loss_given = None
            return given
        return self.new_loss_slot()

    def new_loss_slot(self):
        if self._loss_ring is None:
            self._loss_ring = torch.zeros(LOSS_RING, dtype=torch.float32, device=self.model._device())
            self._loss_ticket = torch.zeros(1, dtype=torch.int32, device=self.model._device())
        loss = self._loss_ring[self._loss_at]
        self._loss_at = (self._loss_at + 1) % LOSS_RING
        return loss

    def _peek_loss_slot(self):
        """Address of the tensor the next _next_loss_slot() returns."""
        given = getattr(self, "_loss_given", None)
        if given is not None:
            return given.data_ptr()
        return self._loss_ring.data_ptr() + 4 * self._loss_at

    def _head_forward_fused(self, ws, R, B):
        """Feature blocks, fused Linear, single-modal heads at the active rows in one launch, then the cosine-BPR rows."""
        m = self.model
        d, fold, W = m.latent_dim, ws["fold"], ws["live_views"]
        act, seg = ws["active_rows"][:R], ws["seg_info"]
        OutAct, YAct = ws["OutAct"][:R], ws["YAct"][:R]
        bw = m._last_block_weights

        packed = self._aux is not None and self._aux is not False      # cs_plan packed the weights on the second stream

        split = packed and getattr(self, "_head_split", False)     # cs_plan ran the feature blocks on the second stream

        with_rows = split and self._rows_in_head

        def head():
            if with_rows:
                self._head_fused_rows_call(ws, R)
            else:
                self._head_fused_call(ws, R, phase=4 if split else (2 if packed else 0))
        m._region("cs_head_fused%d" % (self.cur if with_rows else 2), (m._ws_gen, R, B, packed, split, with_rows, self.nar_act.data_ptr(),
                                    0 if self._out0_src is None else self._out0_src.data_ptr(),
                                    0 if self._peer_src is None else self._peer_src.data_ptr()), head)
        # the cosine-BPR rows and the batch loss in one launch (the workgroup that finishes last adds the loss rows in
        # elimrec_sum's order). Issued directly: the loss goes to the next slot of a ring, so that a caller holding the
        # tensors of earlier steps -- main.py stacks an epoch's losses before it copies them to the host -- does not see them
        # change (LOSS_RING steps back; `loss_ring_len` lets a caller that keeps more clone them).
        loss = self._next_loss_slot()
        if getattr(self, "_step_in_flight", False) and getattr(self, "publish_loss", False):
            # the caller reads every step's loss on the host: summed HERE (the workgroup that finishes last) and stored into
            # coherent host memory by that workgroup -- the host's read returns 120 us into the step and step t + 1 is enqueued
            # under step t's adjoint hops
            ops.bpr_head_rows_sum_pub(YAct, ws["slot_seg"][:3 * B], d, bw, ws["loss_rows"], ws["grad_rows"], loss, self._loss_ticket,
                                      self.loss_publisher().handle)
        elif getattr(self, "_step_in_flight", False) and self._loss_sum_late():
            # a whole step: only the host reads the loss, so its fixed-order sum rides in the LAST launch of the step
            # (an extra workgroup of the Adam hop) instead of ending this one behind a ticket and an acquire
            ops.bpr_head_rows(YAct, ws["slot_seg"][:3 * B], d, bw, ws["loss_rows"], ws["grad_rows"])
            self._loss_late = (ws["loss_rows"][:B], loss)
        else:
            ops.bpr_head_rows_sum(YAct, ws["slot_seg"][:3 * B], d, bw, ws["loss_rows"], ws["grad_rows"], loss, self._loss_ticket)
        m._publish_cache(ws["Y"], dirty=True)
        return loss

    def _pair(self, R, d):
        b = self._pairs.get(R)
        if b is None:
            b = self._pairs[R] = torch.zeros(R, 2, d, dtype=torch.float32, device=self.model._device())
        return b

    def _head_fused_rows_call(self, ws, R):
        """Phase 4 of the fused head with the rows launch folded in: the layer tables of THIS step's buffers."""
        m = self.model
        d, fold, W = m.latent_dim, ws["fold"], ws["live_views"]
        OutAct, YAct = ws["OutAct"][:R], ws["YAct"][:R]
        wu, wi = m._fusion_weights(W)
        tabs = self._tabs
        rows = dict(plan=self.plan, ns=self.ns, w=self.w, L=m.n_layers, U=m.num_users, layers=[t.data for t in tabs] + [None],
                    long_tab=self.long_tab, narrow=self.nar_act)
        if self._direct16:       # (phase 4 reads neither S nor c: the compact buffers stand in for the table arguments)
            c_tab, s_tabs = self.c_rows, [m._fold_rows["S"][k] for k in m._mods]
        else:
            c_tab, s_tabs = fold["c"], [fold[k] for k in m._mods]
        ok = ops.head_fwd_fused_rows(rows, ws["active_rows"][:R], ws["seg_info"], c_tab, s_tabs,
                                     [W[k + "_dense.weight"] for k in m._mods], [W[k + "_dense.bias"] for k in m._mods], wu,
                                     W["embedding_user_after_GCN.bias"], wi, W["embedding_item_after_GCN.bias"],
                                     [W["s_dense_%s.weight" % k] for k in m._mods], [W["s_dense_%s.bias" % k] for k in m._mods],
                                     self._pack, OutAct, YAct, d)
        if not ok:
            raise RuntimeError("fused head forward (with rows) refused a shape _fused_head_ok accepted")

    def _head_fused_call(self, ws, R, phase):
        m = self.model
        d, fold, W = m.latent_dim, ws["fold"], ws["live_views"]
        OutAct, YAct = ws["OutAct"][:R], ws["YAct"][:R]
        wu, wi = m._fusion_weights(W)
        out0 = self._out0_src if self._out0_src is not None else OutAct[:, :d]
        nar = self._nar_src if self._nar_src is not None else self.nar_act
        if self._direct16:
            # 16-bit constants where they lie; the launches that read them (every phase but 4) leave the widened rows of the
            # active nodes in s_rows / c_rows (the backward half's weight gradients read those)
            if phase == 1:
                rows_src = (None, None)
            else:
                rows_src = (self.s_rows, self.c_rows) if phase != 4 else (None, None)
            ok = ops.head_fwd_fused_src16(self.fshard, rows_src[0], rows_src[1], ws["active_rows"][:R], ws["seg_info"], out0, nar,
                                          [W[k + "_dense.weight"] for k in m._mods], [W[k + "_dense.bias"] for k in m._mods], wu,
                                          W["embedding_user_after_GCN.bias"], wi, W["embedding_item_after_GCN.bias"],
                                          [W["s_dense_%s.weight" % k] for k in m._mods], [W["s_dense_%s.bias" % k] for k in m._mods],
                                          self._pack, OutAct, YAct, d, phase=phase)
            if not ok:
                raise RuntimeError("fused head forward (16-bit constants) refused a shape _fused_head_ok accepted")
            return
        if self.lookup:      # compact rows of the constants (looked up / widened for this batch): row r, not row act[r]
            rows, c_tab, s_tabs = self.iota[:R], self.c_rows, [m._fold_rows["S"][k] for k in m._mods]
        else:
            rows, c_tab, s_tabs = ws["active_rows"][:R], fold["c"], [fold[k] for k in m._mods]
        ok = ops.head_fwd_fused(rows, ws["seg_info"], out0, nar, c_tab,
                                s_tabs, [W[k + "_dense.weight"] for k in m._mods],
                                [W[k + "_dense.bias"] for k in m._mods], wu, W["embedding_user_after_GCN.bias"], wi,
                                W["embedding_item_after_GCN.bias"], [W["s_dense_%s.weight" % k] for k in m._mods],
                                [W["s_dense_%s.bias" % k] for k in m._mods], self._pack, OutAct, YAct, d, phase=phase,
                                peers=self._peer_src if phase in (0, 2, 4) else None)
        if not ok:
            raise RuntimeError("fused head forward refused a shape _fused_head_ok accepted")

    @torch.no_grad()
    def cs_backward_local(self, scale):
        m = self.model
        ws, d = m._ws, m.latent_dim
        R = m._plan_n
        # the head backward reads its weight operands from the packed copy the fused forward left behind (16-row forms)
        pack_bwd = self._pack[self._pack_bwd_off:] if (self._fused_head_ok() and self._pack_bwd_off) else None
        # one rank: the merge of the dOut rows into the adjoint sources rides in the weight-gradient launch (both read the
        # head backward's rows and nothing of each other; ELIMREC_FUSE_MERGE=0: a launch of its own before the hops)
        merge = None
        sources = (self.srcA, self.srcB) if (pack_bwd is not None and self._sources_in_head()) else None
        if self.multi and pack_bwd is not None and self._split_in_head():
            sources = ("split", self.send_b, self.world)      # the head backward fills the peers' [H | G] slices itself
        if sources is not None:
            pass                       # the head backward writes the sources itself; their row bitmap is the planner's
        elif not self.multi and self._fuse_merge():
            merge = dict(rows=ws["dOutR"][:R].view(R, m.C), keys=self._acts.reshape(-1), world=1, U=m.num_users, I=m.num_items,
                         srcA=self.srcA, srcB=self.srcB, mask=self.mask, M=m.M)
        self._merged = merge is not None or sources is not None
        # ... and the weight gradients' slab reduce, needed by the optimizer only, in the adjoint's first hop launch
        defer = (not self.multi and self._fuse_reduce())
        if defer and merge is None and self._fuse_bwd_w():
            defer = "all"              # ... and the partial launch too: behind the first hop's tiles, the reduce behind the second's
        if self.multi and self.wgrads_deferred():
            defer = "all"              # several ranks: the same, and the all-reduce waits for the second hop (cs_backward_hops)
        self._grads = m._backward_batch_rows(ws, scale, ws["grad_rows"], R, head_only=True, pack_bwd=pack_bwd,
                                             merge=merge, defer_reduce=defer, sources=sources)
        self._reduce = (m._bwd_w_reduce, 0 if defer == "all" else 1) if defer else None
        if self.word_train:
            self._word_grad(ws, R)
        wg = ws["flat_grad"][ws["tail_off"]:]
        if not self.multi and not self.wide:        # one rank owns every column: the merge reads the dOut rows themselves
            return ws["dOutR"][:R].view(1, R, m.C), wg
        # [H | G]: all the adjoint needs of a dOut row, written straight into the peers' column slices [W, R, 2*dl] (rows
        # beyond the active count are never read: their keys are negative)
        W = self.world
        send = self.send_b
        if sources is not None:
            return send, wg
        m._region("cs_sources", (m._ws_gen, R, send.data_ptr()),
                  lambda: ops.source_rows_split(ws["dOutR"][:R], ws["seg_info"][0:1], d, m.M, W, send))
        return send, wg

    def join_source_bits(self):
        """The caller's stream behind the second stream's source bits of the adjoint's first hop (several ranks: cs_gathered_ids;
        large batches: cs_forward_rows), once per step."""
        if getattr(self, "_bits_join", False):
            program.sync(torch.cuda.current_stream(), self._aux)
            self._bits_join = False

    @torch.no_grad()
    def cs_backward_hops(self, recv2, acts, grads_ready=None, reduce_wgrads=None):
        """grads_ready (several ranks): the handle of the weight gradients' all-reduce. Waited for before the LAST hop, whose
        launch then carries the projection weights' optimizer spans as one rank's does; without it they run in cs_update.
        reduce_wgrads (several ranks, deferred weight gradients): called once the hops that finish the weight gradients are
        enqueued; returns the all-reduce's handle (kept in self.wgrads_handle), which then runs under the last hop. True: a
        group of one rank -- nothing to reduce, the spans ride in the last hop."""
        m = self.model
        U, I, L = m.num_users, m.num_items, m.n_layers
        W, R = acts.shape
        inv = 1.0 / (L + 1)

        fuse = self._fuse_adam() and not getattr(self, "grads_only", False)
        last = 1 if fuse else 0                                    # the hops the recorded region covers: L-1 .. last

        single = not self.multi
        merged = single and getattr(self, "_merged", False)
        self.join_source_bits()
        reduce = getattr(self, "_reduce", None) if (single or reduce_wgrads is not None) else None
        self._reduce = None
        self.wgrads_handle = None

        if self.wide:
            # the wide form: ONE source [H | G] for every layer (left <- H, right <- G, no user / item side swap), L whole hops
            # T <- Src + A^T T, then the parameters' gradient = left half on user rows / right half on item rows
            def wide_hops():
                slab.merge_rows(recv2.view(W * R, 2 * self.dl), acts.reshape(-1), W, U, I, self.srcA, self.srcB, self.mask, M=-1)
                t, tmask = self.srcW, self.mask
                for k in range(L - 1, -1, -1):
                    dst = self.tmp[k & 1]
                    slab.hop(self.planT, t, dst, gs=self.hgs, src_mask=tmask, add=self.srcW, add_mask=self.mask, scale=1.0,
                             bits_ready=tmask is not None and self._bits_ready)
                    t, tmask = dst, None
                slab.wide_grad(t, U, inv, self.grad)
            self._timed(lambda: m._region("cs_bwd_hops", (m._ws_gen, recv2.data_ptr(), acts.data_ptr(), R, W, self._bits_ready), wide_hops), L)
            self._adam_in_hop = self._tail_in_hop = False
            if reduce_wgrads is not None and reduce_wgrads is not True:
                self.wgrads_handle = reduce_wgrads()
            return

        def hops():
            phase = None if reduce is None else reduce[1]
            if merged:
                pass                                               # done beside the weight gradients (cs_backward_local)
            elif single:
                slab.merge_rows(recv2.view(R, m.C), acts.reshape(-1), 1, U, I, self.srcA, self.srcB, self.mask, M=m.M)
            else:
                slab.merge_rows(recv2.view(W * R, 2 * self.dl), acts.reshape(-1), W, U, I, self.srcA, self.srcB, self.mask)
            t, tmask = (self.srcB if (L & 1) else self.srcA), self.mask          # T^L = S^L (row-sparse)
            for k in range(L - 1, last - 1, -1):
                dst = self.grad if k == 0 else self.tmp[k & 1]
                slab.hop(self.planT, t, dst, gs=self.gs, src_mask=tmask, add=self.srcB if (k & 1) else self.srcA,
                         add_mask=self.mask, scale=inv if k == 0 else 1.0, bits_ready=tmask is not None and self._bits_ready,
                         bwd_w=reduce[0] if phase is not None and phase <= 1 else None, bwd_w_phase=phase)
                phase = None if phase is None else phase + 1
                t, tmask = dst, None
        self._timed(lambda: m._region("cs_bwd_hops", (m._ws_gen, recv2.data_ptr(), acts.data_ptr(), R, W, fuse, self._bits_ready, merged,
                                                               0 if reduce is None else ctypes.addressof(reduce[0][0]), 0 if reduce is None else reduce[1]), hops), L - last)
        self._adam_in_hop = fuse
        self._tail_in_hop = False
        if fuse:
            # the last hop's output is the gradient: the embeddings' Adam step is its epilogue (no gradient table written and
            # read back). One rank: the projection weights' spans ride along as extra workgroups of the same launch (their
            # gradients are complete since the head's backward; with peers they still wait for the all-reduce -> cs_update).
            # Issued outside the recorded region: the bias-correction constants change every step.
            g = self.opt.param_groups[0]
            nxt = 1 - self.cur
            in_hop = single or grads_ready is not None
            if reduce_wgrads is True:                         # (a group of one rank: nothing to reduce)
                in_hop = True
            elif reduce_wgrads is not None:                   # finished by the hops above: reduce them under this last hop
                self.wgrads_handle = reduce_wgrads()
                in_hop = False
            if grads_ready is not None:
                grads_ready.wait()                            # reduced under the hops issued so far
            tail = self._tail_jobs() if in_hop else []        # (advances the weights' step counts: called once per step)
            if len(tail) > 8:
                raise RuntimeError("more than 8 optimizer spans")
            if getattr(self, "_tail_arr", None) is None:
                self._tail_arr = (_lib.AdamJob * 8)()         # one array for every step: its address is part of the step's
            for i, job in enumerate(tail):                    # recorded arguments (program.py), its step counts change in place
                self._tail_arr[i] = job
            self._tail_n, self._tail_base = len(tail), 0
            tail = (self._tail_arr, len(tail))
            self._tail_in_hop = in_hop
            late = getattr(self, "_loss_late", None)          # cs_head left the loss rows' sum to this launch
            self._loss_late = None
            self._timed(lambda: slab.hop_adam(self.planT, self.tmp[1], self.grad if self.keep_grad else None, self.gs, self.srcA,
                                              self.mask, inv, self.master[self.cur].data, self.master[nxt].data, self.m1, self.m2,
                                              g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                                              self.step_count + 1, tail_jobs=tail, loss_sum=late), 1)

    @torch.no_grad()
    def cs_update(self):
        """Adam (coupled L2) on the column shard of the embeddings (buffer `cur` -> the other one)
        and on the projection weights that received a gradient, in ONE launch; the same launch copies the projection
        weights as they were BEFORE the update into the snapshot predict()'s lazily built tables use."""
        m = self.model
        g = self.opt.param_groups[0]
        self.step_count += 1
        nxt = 1 - self.cur
        jobs = []
        if not getattr(self, "_adam_in_hop", False):              # else the last adjoint hop has already applied it
            jobs.append(_lib.AdamJob(self.master[self.cur].data.data_ptr(), self.master[nxt].data.data_ptr(),
                                     self.grad.data.data_ptr(),
                                     self.m1.data_ptr(), self.m2.data_ptr(), None, self.grad.data.numel(), self.step_count))
        if not getattr(self, "_tail_in_hop", False):              # else they ran as extra workgroups of the last hop
            jobs += self._tail_jobs()
        if len(jobs) > 8:
            raise RuntimeError("more than 8 optimizer spans")
        if jobs:
            if getattr(self, "_tail_arr", None) is None:
                self._tail_arr = (_lib.AdamJob * 8)()         # one array for every step (its address is a recorded argument)
            for i, job in enumerate(jobs):
                self._tail_arr[i] = job
            self._tail_base = 0 if getattr(self, "_adam_in_hop", False) else 1           # slot 0: the embeddings' span
            self._tail_n = len(jobs) - self._tail_base
            _lib.check(_lib.load().elimrec_adam_multi(self._tail_arr, len(jobs), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                                      g["weight_decay"], ops._stream()), "adam_multi")
        self.cur = nxt
        self._updated()

    def _updated(self):
        """The master copy now holds newer embeddings than the model's parameters: whoever reads those next writes it back
        first (plugin.py: EmbeddingParameter, state_dict)."""
        ctl = getattr(self.model, "_plugin", None)
        if ctl is not None and ctl.engine is self:
            ctl.master_newer = True

    def native_prologue(self):
        """What the ordinary step's Python does besides launching, ahead of the launches of a native step: the projection
        weights' step counts (optimizer state and the persistent job array the last hop's launch reads)."""
        base = getattr(self, "_tail_base", 0)
        if base:            # the embeddings' Adam is a span of the optimizer launch (wide form), not part of the last hop:
            nxt = 1 - self.cur          # this step's buffers and step count into the persistent job array the launch reads
            self._tail_arr[0] = _lib.AdamJob(self.master[self.cur].data.data_ptr(), self.master[nxt].data.data_ptr(),
                                             self.grad.data.data_ptr(),
                                             self.m1.data_ptr(), self.m2.data_ptr(), None, self.grad.data.numel(), self.step_count + 1)
        if self._tail_plan is not None and getattr(self, "_tail_n", 0):
            _, spans, states = self._tail_plan[:3]
            for st in states:
                st["step"] += 1
            i = base
            for sp in spans:
                if sp["upd"]:
                    sp["step"] += 1
                    self._tail_arr[i].step = sp["step"]
                i += 1

    def native_epilogue(self, R):
        """... and behind them: the buffer flip, the step count, the markers predict() / evaluate() go by."""
        m = self.model
        x0 = self.master[self.cur]
        self._x0_fwd = x0
        self._tabs = [x0] + self.layers[1:]
        self._srcs = [x0] + self.layers[1:]
        m._plan_n = R
        m._slab_fwd = True
        m._publish_cache(m._ws["Y"], dirty=True)
        self.step_count += 1
        self.cur = 1 - self.cur
        self._updated()

    def _tail_jobs(self):
        """Spans of the flat parameter buffer behind the embeddings: runs of adjacent projection weights that have a
        gradient (Adam in place, moments = the caller's FusedAdam state, whose step counts advance) and the runs that
        have none (copied to the snapshot only). The span list is fixed for a given set of gradients."""
        m, ws = self.model, self.model._ws
        have = tuple(name for name, _ in self._tail if self._grads.get(name) is not None)
        if self._tail_plan is None or self._tail_plan[0] != have:
            base_p, base_g, snap = ws["flat_param"].data_ptr(), ws["flat_grad"].data_ptr(), ws["snap"].data_ptr()
            tail_off = ws["tail_off"]
            spans, states = [], []
            for name, p in self._tail:
                off, numel = ws["param_off"][name]
                upd = name in have
                st = self.opt._state_for(p) if upd else None
                if upd:
                    states.append(st)
                if spans and spans[-1]["upd"] == upd and spans[-1]["end"] + 3 >= off and (not upd or (
                        st["step"] == spans[-1]["step"] and st["exp_avg"].data_ptr() == spans[-1]["m"] + 4 * (off - spans[-1]["off"]))):
                    spans[-1]["end"] = off + numel
                else:
                    spans.append(dict(off=off, end=off + numel, upd=upd, step=st["step"] if upd else 0,
                                      m=st["exp_avg"].data_ptr() if upd else 0, v=st["exp_avg_sq"].data_ptr() if upd else 0))
            self._tail_plan = (have, spans, states, base_p, base_g, snap, tail_off)
        _, spans, states, base_p, base_g, snap, tail_off = self._tail_plan
        for st in states:
            st["step"] += 1
        jobs = []
        for sp in spans:
            n, o = sp["end"] - sp["off"], sp["off"]
            if sp["upd"]:
                sp["step"] += 1
                jobs.append(_lib.AdamJob(base_p + 4 * o, base_p + 4 * o, base_g + 4 * o, sp["m"], sp["v"],
                                         snap + 4 * (o - tail_off), n, sp["step"]))
            else:
                jobs.append(_lib.AdamJob(base_p + 4 * o, None, None, None, None, snap + 4 * (o - tail_off), n, 0))
        return jobs

    # ------------------------------------------------------------------ row-sharded constants: the distributed fold
    @torch.no_grad()
    def _horner_mean(self, x0_rows, plan=None):
        """mean_k A^k X0 of a row-major [N x cols] table with the engine's own hop kernels: t <- X0 + A t, L times, the last with
        the 1/(L+1) scale (plan: the matrix's wave-tile plan, default the adjacency's; its transpose's gives the adjoint).
        cols is padded to the column count per slab group of the engine's plan (the wave-tile plan is laid out for that many
        lanes per row piece)."""
        m, N, L = self.model, x0_rows.shape[0], self.model.n_layers
        per_group = (self.hns // self.hgs) * self.w                  # columns one lane group covers
        cols = x0_rows.shape[1]
        padded = (cols + per_group - 1) // per_group * per_group
        if padded != cols:
            x0_rows = torch.cat([x0_rows, torch.zeros(N, padded - cols, dtype=torch.float32, device=x0_rows.device)], dim=1)
        ns, gs = padded // self.w, padded // per_group
        mk = lambda: slab.SlabTable(N, ns, self.w, x0_rows.device)
        x0, a, b = mk().from_rows(x0_rows.contiguous()), mk(), mk()
        t = x0
        for k in range(L):
            dst = a if t is not a else b
            slab.hop(self.plan if plan is None else plan, t, dst, gs=gs, add=x0, scale=1.0 / (L + 1) if k == L - 1 else 1.0)
            t = dst
        return t.dense()[:, :cols]

    # ------------------------------------------------------------------ the tiktok data set's word embeddings
    def _word_setup(self):
        """models/EliMRec.py:371-378 + main.py:100 (`retain_graph=True`): on the data set "tiktok" t_feat is built ONCE from
        word_embedding (scatter-mean of the items' words) and never again, but its graph is retained -- so every step's backward
        still reaches word_embedding, and Adam (with its weight decay) keeps moving a parameter nothing reads. To train the
        checkpoint the reference trains, the engine does the same: per step, dL/dt_feat = (mean_k (A^T)^k dOut_t)[items] W_t
        (one more adjoint propagation of a d-column table, with the hop kernels) and dL/dword_embedding = the scatter-mean's
        adjoint, a fixed sparse matrix [words x items] of 1 / (words of the item) applied by the deterministic row kernel
        (elimrec_block_spmm). `--word_embedding=frozen` skips all of it (losses and scores are the same either way: nothing reads
        the parameter after start-up). One rank, regular tables."""
        import scipy.sparse as sp
        m = self.model
        words = m.dataset.words_tensor
        it, wd = words[0].numpy().astype(np.int64), words[1].numpy().astype(np.int64)
        cnt = np.bincount(it, minlength=m.num_items).astype(np.float64)
        vals = (1.0 / cnt[it]).astype(np.float32)
        V = m.word_embedding.weight.shape[0]
        mat = sp.csr_matrix((vals, (wd, it)), shape=(V, m.num_items))       # duplicate (word, item) pairs add up, as the mean's adjoint does
        dev = m._device()
        self._word_csr = ops.Csr.from_scipy(mat, dev, C=m.word_embedding.weight.shape[1])
        N, d = m.num_users + m.num_items, m.latent_dim
        self._word_full = torch.zeros(N + 1, d, dtype=torch.float32, device=dev)          # + one row that collects the unused slots
        self._word_dF = torch.empty(m.num_items, m.word_embedding.weight.shape[1], dtype=torch.float32, device=dev)

    @torch.no_grad()
    def _word_grad(self, ws, R):
        m = self.model
        d, U = m.latent_dim, m.num_users
        N = U + m.num_items
        k = m._mods.index("t")
        blk = ws["dOutR"][:R, (k + 1) * d:(k + 2) * d]
        act = ws["active_rows"][:R].long()
        full = self._word_full
        full.zero_()
        full.index_copy_(0, torch.where(act >= 0, act, torch.full_like(act, N)), blk)       # active rows are listed once
        g_items = self._horner_mean(full[:N], plan=self.planT)[U:].contiguous()            # [I x d]: dL / d(F_t W_t^T + b_t)
        wt = ws["live_views"]["t_dense.weight"]                                            # [d x D_t]
        ops.linear_fwd_batched([(g_items, wt.t().contiguous(), None, self._word_dF)])      # dL/dt_feat = g_items . W_t
        gview = ws["grad_views"]["word_embedding.weight"]
        ops.block_spmm(self._word_csr, self._word_dF, Xout=gview)
        self._grads["word_embedding.weight"] = gview

    @torch.no_grad()
    def _fold_sharded(self, owners, frank):
        """S_m = mean_k A^k [0 ; F_m] and c = mean_k A^k [0 ; 1] WITHOUT any rank holding a full [N x D_m] table: the rank takes
        its item block of the raw features (models/EliMRec.py:366-381: a slice of the host tensor, or -- --feature_load=block --
        the one block the rank reads from the file, dataset.FeatureBlocks), one all_to_all turns row blocks into COLUMN slices [I x D_m/W], every rank propagates its
        slice through the (replicated) graph with the hop kernels -- no communication, like the training hops -- and a second
        all_to_all hands every owner its rows of every slice. c is one column: every rank computes it, keeps its rows."""
        from .shard_eval import collectives_for
        m, W = self.model, owners.world
        U, I, dev = m.num_users, m.num_items, m._device()
        q = frank
        real = W > 1 and dist.is_available() and dist.is_initialized()      # (ranks emulated in one process: whole tables, own rows)
        coll = collectives_for(self.group, W) if real else None
        nodes = [torch.from_numpy(owners.nodes(o)).to(dev) for o in range(W)]
        rows = [len(n) for n in nodes]
        i_rows = [owners.rows(o)[1] for o in range(W)]
        i0, i1 = int(owners.ib[q]), int(owners.ib[q + 1])
        tabs = []
        for k in m._mods:
            feat = getattr(m, k + "_feat")
            D = feat.shape[1]
            if real and D % W == 0 and (D // W) % 4 == 0:
                Dq = D // W
                mine = feat[i0:i1].to(dev)                                          # my item block, all columns
                cols = coll.all_to_all_rows(torch.cat([mine[:, p * Dq:(p + 1) * Dq] for p in range(W)]).contiguous(),
                                            [i1 - i0] * W, i_rows)                  # all items, my columns
                x0 = torch.cat([torch.zeros(U, Dq, dtype=torch.float32, device=dev), cols])
                S = self._horner_mean(x0)                                           # [N x Dq]
                recv = coll.all_to_all_rows(torch.cat([S[n] for n in nodes]), rows, [rows[q]] * W).view(W, rows[q], Dq)
                tabs.append(recv.permute(1, 0, 2).reshape(rows[q], D).contiguous())
            else:       # a width that does not split: the whole table on every rank (small shapes), own rows kept
                whole = feat[0:I] if hasattr(feat, "rows_read") else feat       # (a block-loaded table: the one block there is)
                x0 = torch.cat([torch.zeros(U, D, dtype=torch.float32, device=dev), whole.to(dev)])
                tabs.append(self._horner_mean(x0)[nodes[q]].contiguous())
        ones = torch.zeros(U + I, 4, dtype=torch.float32, device=dev)
        ones[U:, 0] = 1.0
        c_loc = self._horner_mean(ones)[nodes[q], 0].contiguous()
        return tabs, c_loc

    # ------------------------------------------------------------------ row-sharded constants: the lookup's device steps
    def cs_lookup_counts(self, acts):
        """int32 [W x W] on the device: rows of requester r's list that owner o holds."""
        return self.fshard.counts(acts.contiguous())

    @torch.no_grad()
    def cs_lookup_plan_counts(self, batches):
        """int64 [n_batches x W] on the host: per batch, how many of ITS active rows (unique users, positives, negatives) each
        rank owns -- the epoch-ahead form of cs_lookup_counts for this rank's own lists."""
        m, own = self.model, self.fshard.owners
        U, W = m.num_users, own.world
        dev = batches[0][0].device
        ub = torch.from_numpy(own.ub[1:].copy()).to(dev)
        ib = torch.from_numpy(own.ib[1:].copy()).to(dev)
        out = torch.zeros(len(batches), W, dtype=torch.int64, device=dev)
        by_size = {}
        for k, (u, p, n) in enumerate(batches):
            by_size.setdefault(int(u.numel()), []).append(k)
        for B, ks in by_size.items():
            u = torch.stack([batches[k][0] for k in ks]).long()
            it = torch.cat([torch.stack([batches[k][1] for k in ks]), torch.stack([batches[k][2] for k in ks])], dim=1).long()
            for ids, bounds in ((u, ub), (it, ib)):
                srt, _ = torch.sort(ids, dim=1)
                first = torch.ones_like(srt, dtype=torch.bool)
                first[:, 1:] = srt[:, 1:] != srt[:, :-1]
                owner = torch.bucketize(srt, bounds, right=True).clamp_(max=W - 1)
                flat = (torch.arange(len(ks), device=dev)[:, None] * W + owner)[first]
                cnt = torch.zeros(len(ks) * W, dtype=torch.int64, device=dev).scatter_add_(0, flat, torch.ones_like(flat))
                out[torch.tensor(ks, device=dev)] += cnt.view(len(ks), W)
        return out.cpu().numpy()

    def cs_lookup_pack(self, acts):
        """Owner side: my rows of every rank's list -> a flat uint8 send buffer (chunks in requester order)."""
        W, R = acts.shape
        buf = self._lookup_bufs.get(("send", W, R))
        if buf is None:
            buf = self._lookup_bufs[("send", W, R)] = torch.empty(W * R * self.lookup_row_bytes, dtype=torch.uint8, device=acts.device)
        self.fshard.pack(acts.contiguous(), buf)
        return buf

    def cs_lookup_recv(self, nbytes):
        R = self.model._plan_n
        buf = self._lookup_bufs.get(("recv", R))
        if buf is None:
            buf = self._lookup_bufs[("recv", R)] = torch.empty(R * self.lookup_row_bytes, dtype=torch.uint8, device=self.model._device())
        return buf[:nbytes]

    def cs_lookup_unpack(self, recv):
        """Requester side: the received chunks -> compact fp32 rows of the constants in active-row order."""
        R = self.model._plan_n
        self.fshard.unpack(self.model._ws["active_rows"][:R], recv, self.s_rows, self.c_rows)

    # ------------------------------------------------------------------ cached tables for predict()
    @torch.no_grad()
    def materialize_tables(self, ws):
        """ws['Out'] / ws['Y'] over all rows from the layer tables of the last forward and the projection weights it
        used (models/EliMRec.py:98-99: predict() reads what the last TRAINING forward computed)."""
        m = self.model
        L, U, d = m.n_layers, m.num_users, m.latent_dim
        N = U + m.num_items
        if self._x0_fwd is None:
            raise RuntimeError("no forward has run on the column-sharded engine yet")
        if self.wide:
            tabs = [self.x0w] + self.layers[1:]                  # every layer of the last forward, whole (x0w was built from its master)
        else:
            if self.xL is None:
                self.xL = self.layers[1].like() if L >= 2 else self.grad.like()
            tabs = [self._x0_fwd] + self.layers[1:]
            slab.hop(self.plan, self._srcs[L - 1], self.xL, gs=self.gs)

        def all_rows(out0, narrow):
            slab.rows(self.plan, self.ns, self.w, L, U, [t.data for t in tabs] + [self.xL.data], None, None, None, N, 1,
                      out0, narrow, False)
        if self.lookup and ((self.feature_shard == "row" and (self.world > 1 or self.lean)) or ws.get("fold") is None):
            def rows_of(node_ids, out0, narrow):               # (layer mean | shared part) of the listed rows, my columns
                n = int(node_ids.numel())
                if self.wide:
                    slab.wide_rows([t.data for t in tabs], N, self.ns, self.w, node_ids, n, out0, narrow)
                else:
                    slab.rows(self.plan, self.ns, self.w, L, U, [t.data for t in tabs] + [self.xL.data], None, node_ids.view(1, n), None, n, 1,
                              out0, narrow, False)
            return self._materialize_item_shard(ws, rows_of)
        if self.world == 1:
            all_rows(ws["Out"][:, :d], ws["Narrow"])
        else:
            loc = torch.empty(N, 2 * self.dl, dtype=torch.float32, device=ws["Out"].device)
            all_rows(loc[:, :self.dl], loc[:, self.dl:])
            parts = _all_gather_parts(loc, self.world, self.group)
            ws["Out"][:, :d].copy_(torch.cat([p[:, :self.dl] for p in parts], dim=1))
            ws["Narrow"].copy_(torch.cat([p[:, self.dl:] for p in parts], dim=1))
        m._full_tables(ws, ws["snap_views"])

    @torch.no_grad()
    def _materialize_item_shard(self, ws, rows_of):
        """Row-sharded constants: the cached tables are built ROW-sharded too -- every rank computes Out / Y for the nodes it
        owns (its users, its items) from its own rows of S_m / c, and evaluation runs item-sharded (shard_eval.py).
        rows_of(node ids, out0, narrow): (layer mean | shared part) of listed rows in MY columns. In chunks of
        `materialize_rows` (2^18) of every owner's rows (the transient stays a few GiB whatever N is): one all_to_all hands
        every owner its chunk's rows of every rank's columns (the column shards' transpose), the projections run on the
        chunk, and one all_gather at the end replicates the users' Y rows (every rank scores all users against its items)."""
        import os
        from .shard_eval import HipShardBackend, ItemShardScorer, collectives_for
        m, W, q, own = self.model, self.world, self.rank, self.fshard.owners
        d, dl, C, Cy = m.latent_dim, self.dl, m.C, m.Cy
        dev = m._device()
        coll = collectives_for(self.group, W)
        if getattr(self, "_own_nodes", None) is None:
            self._own_nodes = [torch.from_numpy(own.nodes(o)).to(device=dev, dtype=torch.int32) for o in range(W)]
        rows = [int(n.numel()) for n in self._own_nodes]
        mine = rows[q]
        nu = own.rows(q)[0]
        U, i_loc = m.num_users, mine - nu
        Yshard = torch.empty(U + i_loc, Cy, dtype=torch.float32, device=dev)     # [all users ; my items]: what the scorer reads
        Yu_loc, Yi = torch.empty(nu, Cy, dtype=torch.float32, device=dev), Yshard[U:]
        Wt = ws["snap_views"]                                       # the weights the last forward used (predict() is stale)
        wu, wi = m._fusion_weights(Wt)
        step = self.materialize_rows
        for r0 in range(0, max(rows), step):
            span = [(min(r0, n), min(r0 + step, n)) for n in rows]          # this chunk of every owner's local rows
            cnt = [b - a for a, b in span]
            ids = torch.cat([self._own_nodes[o][a:b] for o, (a, b) in enumerate(span)])
            loc = torch.empty(int(ids.numel()), 2 * dl, dtype=torch.float32, device=dev)
            if ids.numel():
                rows_of(ids, loc[:, :dl], loc[:, dl:])
            k = cnt[q]
            recv = coll.all_to_all_rows(loc, cnt, [k] * W).view(W, k, 2, dl)
            if k == 0:
                continue
            a, b = span[q]
            Out = torch.empty(k, C, dtype=torch.float32, device=dev)
            Nar = torch.empty(k, d, dtype=torch.float32, device=dev)
            ops.peer_cols_to_rows(recv.view(W, k, 2 * dl), Out[:, :d], Nar)
            # my rows of the constants, widened to fp32 (direct read of the local table, in its own order)
            S = torch.empty(k, self.fshard.sum_d, dtype=torch.float32, device=dev)
            c = torch.empty(k, dtype=torch.float32, device=dev)
            self.fshard.unpack(self._own_nodes[q][a:b].contiguous(), None, S, c, direct=True)
            problems, off = [], 0
            for j, (name, D) in enumerate(zip(m._mods, self.fshard.dims)):
                problems.append((S[:, off:off + D], Wt[name + "_dense.weight"], Wt[name + "_dense.bias"], Out[:, (j + 1) * d:(j + 2) * d], c, Nar))
                off += D
            ops.linear_fwd_batched(problems)
            ku = max(0, min(b, nu) - a)                             # user rows of this chunk come first
            for lo, hi, dst, wf, bf in ((0, ku, Yu_loc[a:a + ku], wu, "embedding_user_after_GCN.bias"),
                                        (ku, k, Yi[max(a, nu) - nu:b - nu], wi, "embedding_item_after_GCN.bias")):
                if hi <= lo:
                    continue
                head = [(Out[lo:hi], wf, Wt[bf], dst[:, :d])]
                for h, name in enumerate(m._mods):
                    blk = slice((h + 1) * d, (h + 2) * d)
                    head.append((Out[lo:hi, blk], Wt["s_dense_%s.weight" % name], Wt["s_dense_%s.bias" % name], dst[:, blk]))
                ops.linear_fwd_batched(head)
        Yshard[:U].copy_(coll.all_gather_rows(Yu_loc, [own.rows(o)[0] for o in range(W)]))           # every rank: all users' rows
        del Yu_loc
        i0, i1 = int(own.ib[q]), int(own.ib[q + 1])
        m._eval_shard = ItemShardScorer(HipShardBackend(m, Yshard, i0, i1), coll, own.ib)
        m._eval_shard_Y = Yshard
        m._publish_cache(Yshard, dirty=False)

    # ------------------------------------------------------------------ checkpoint / resume (full, rank-independent tensors)
    @torch.no_grad()
    def _gather_cols(self, flat):
        loc = slab.SlabTable(self.grad.n, self.ns, self.w, flat.device, data=flat).dense()
        if self.world == 1:
            return loc
        return torch.cat(_all_gather_parts(loc, self.world, self.group), dim=1)

    @torch.no_grad()
    def optimizer_state(self):
        """Adam state of the embeddings as row-major [N x d] tensors + the step count (all ranks must call it)."""
        return dict(step=self.step_count, exp_avg=self._gather_cols(self.m1).cpu(), exp_avg_sq=self._gather_cols(self.m2).cpu())

    @torch.no_grad()
    def load_optimizer_state(self, st):
        dev = self.grad.data.device
        self.step_count = int(st["step"])
        for flat, key in ((self.m1, "exp_avg"), (self.m2, "exp_avg_sq")):
            full = st[key].to(dev).float().contiguous()
            slab.SlabTable(self.grad.n, self.ns, self.w, dev, data=flat).from_rows(full, col0=self.col0)
