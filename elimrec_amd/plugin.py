"""The reference's own loop body on the column-shard engine.

    loss = model.bpr_loss(users, pos, neg)      # main.py:98   (models/EliMRec.py:115-142)
    opt.zero_grad()                             # main.py:99
    loss.backward(retain_graph=True)            # main.py:100
    opt.step()                                  # main.py:101
    ... loss.cpu().item()                       # main.py:102

The engine's training step (shard.py) is ONE enqueue: forward hops, head, cosine-BPR, head backward, adjoint hops with the
weight gradients behind their tiles and Adam as the last hop's epilogue -- there is no point in it where "the gradients" exist
as tensors and the update has not happened. So the four calls above are not four pieces of GPU work; they are four
statements of intent, and `StepController` runs the step when the intent is complete:

  * `bpr_loss` hands out the loss -- a 0-dim tensor in the engine's loss ring -- and notes the batch. Nothing is enqueued.
  * `loss.backward()` (no `gradient=`, no `inputs=`) notes that gradients are wanted.
  * `opt.step()` (`FusedAdam`) enqueues the whole step -- `ColumnShardTrainer.step`, the native program of program.py --
    which fills the loss. This is the fast path: the same kernels, in the same order, as `ColumnShardTrainer.step`.

Anything that LOOKS at an intermediate result in between makes it real first, launch by launch, with the bits of the one-enqueue
form (tests/test_api_gpu.py::test_plugin_loop_*):
  * reading the loss (any torch function on it: `.item()`, `.cpu()`, arithmetic, printing) before the step has run, reading
    `model.all_users` / `all_items` / `all_s_embs`, `predict()`, `evaluate()`: the forward half runs (trainer.forward_only);
  * reading a parameter's `.grad` after `backward()`, `backward(gradient=...)`, `torch.autograd.grad`, a second `bpr_loss`
    before the step: forward + backward half with the gradient table stored (trainer.backward_only(grads_only=True)); `.grad`
    then holds real tensors (the flat gradient views), which a following `opt.step()` -- FusedAdam or any torch optimizer --
    consumes as usual, modifications included;
  * reading the embedding parameters themselves (`model.embedding_user.weight`, `state_dict()`): the engine's master copy
    (slab-major, updated by the fused Adam) is written back to them first.
The embedding parameters the other way round -- `load_state_dict`, a foreign optimizer, in-place edits under no_grad -- are
noticed by their version counters and re-loaded into the master copy before the next step.
"""
import torch
from torch import nn

_GRAD = torch._C.TensorBase.grad            # the C-level getset descriptor behind Tensor.grad

# Tensor methods / attributes that read no element of the tensor: no reason to synchronise the master copy for them
_METADATA = frozenset(n for n in ("data_ptr", "size", "dim", "numel", "nelement", "element_size", "is_contiguous", "stride",
                                  "storage_offset", "untyped_storage", "is_floating_point", "is_complex", "type", "get_device",
                                  "__len__", "ndimension", "requires_grad_", "register_hook", "retain_grad", "is_shared",
                                  "_is_view", "has_names", "is_pinned", "__hash__", "__repr__", "__deepcopy__", "__reduce_ex__"))


class LazyGradParameter(nn.Parameter):
    """nn.Parameter of a model whose training step is deferred (StepController): `.grad` is produced on first read when a
    backward() was requested and has not run yet. isinstance(p, nn.Parameter) holds, state_dict keys are unchanged."""

    @property
    def grad(self):
        ctl = self.__dict__.get("_elimrec_ctl")
        if ctl is not None and ctl.grads_deferred():
            tr = getattr(ctl, "trainer", None)
            if tr is not None and tr.world > 1:
                # materialising is a collective (all-gather of the column slices, all-to-all + all-reduce of the backward half): a
                # read on ONE rank -- gradient-norm logging on rank 0 -- would hang the job. The same guard as sync_params.
                raise RuntimeError("reading .grad between backward() and the optimizer step materialises the gradients, which is a "
                                   "collective over %d ranks: call model.plugin.materialise_grads() on ALL ranks first" % tr.world)
            ctl.materialise_grads()
        return _GRAD.__get__(self)

    @grad.setter
    def grad(self, value):
        ctl = self.__dict__.get("_elimrec_ctl")
        if ctl is not None:
            if value is None:
                ctl.cancel_backward()           # zero_grad() behind a backward(): that backward is void, as with torch
            else:
                ctl.grads_set = True
        _GRAD.__set__(self, value)

    def _raw_grad(self):
        return _GRAD.__get__(self)


def _instances(args, cls, kwargs=None):
    """The `cls` instances among a torch function's arguments, one level of lists / tuples included (torch.cat([p, q]))."""
    for a in list(args) + (list(kwargs.values()) if kwargs else []):
        if isinstance(a, cls):
            yield a
        elif isinstance(a, (list, tuple)):
            for b in a:
                if isinstance(b, cls):
                    yield b


def _reads_values(func):
    """Does this torch function / Tensor method / attribute getter look at the tensor's elements?"""
    name = getattr(func, "__name__", "")
    if name == "__get__":                       # an attribute descriptor: .data and the transposes hand out the values
        return getattr(getattr(func, "__self__", None), "__name__", "") in ("data", "T", "mT", "H", "mH", "real", "imag")
    return name not in _METADATA and name != "__set__"


class EmbeddingParameter(LazyGradParameter):
    """... and whose VALUE lives in the engine's master copy between steps (embedding_user / embedding_item): any torch
    function that reads it writes the master copy back first."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if _reads_values(func):
            for a in _instances(args, EmbeddingParameter, kwargs):
                ctl = a.__dict__.get("_elimrec_ctl")
                if ctl is not None and ctl.master_newer:
                    ctl.sync_params(implicit=True)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


_HOST_READS = frozenset(("item", "cpu", "tolist", "__float__"))      # the loss taken to the host (main.py:102: loss.cpu().item())


class PendingLoss(torch.Tensor):
    """The 0-dim loss `bpr_loss` returns: a view of a slot the step will fill. `backward()` without arguments is taken as a
    request (no autograd engine); every other use makes the value real first. A caller that takes every step's loss to the
    host (`loss.cpu().item()`, main.py:102) is noticed: from the next step on the loss-summing launch publishes the value
    into coherent host memory and the read waits for that launch alone -- not for the adjoint hops and Adam behind it."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        reads = _reads_values(func) or func is torch.Tensor.backward
        name = getattr(func, "__name__", "")
        if name in _HOST_READS and len(args) == 1 and not kwargs and isinstance(args[0], PendingLoss):
            me = args[0]
            ctl = me.__dict__.get("_elimrec_ctl")
            if ctl is not None:
                ctl.host_read_gen = me.__dict__.get("_elimrec_gen")
                seq = me.__dict__.get("_elimrec_pub")
                if seq is not None:
                    val = ctl.engine.loss_publisher().wait(seq)
                    if val is not None:
                        return torch.tensor(val, dtype=torch.float32) if name == "cpu" else val
        for me in _instances(args, PendingLoss, kwargs):
            ctl = me.__dict__.get("_elimrec_ctl")
            if ctl is None:
                continue
            if func is torch.Tensor.backward and args and args[0] is me and len(args) == 1 and kwargs.get("gradient") is None \
                    and kwargs.get("inputs") is None and ctl.request_backward(me):
                return None
            if reads:
                ctl.realise_forward(me)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class _DeferredBprFn(torch.autograd.Function):
    """Gives the loss a grad_fn, for the uses the fast path does not cover: backward(gradient=...), arithmetic on the loss before
    backward, torch.autograd.backward. backward() here runs forward + backward halves with the gradients stored and assigns
    them to `.grad` itself (models/EliMRec.py's autograd would accumulate 18 separate tensors; these are views of one flat
    buffer)."""

    @staticmethod
    def forward(ctx, ctl, anchor):
        ctx.ctl, ctx.gen = ctl, ctl.gen
        return ctl.engine.new_loss_slot()

    @staticmethod
    def backward(ctx, grad_out):
        ctl = ctx.ctl
        pend = ctl.pending
        if pend is None or pend["gen"] != ctx.gen:
            raise RuntimeError("backward of a stale loss: another bpr_loss / optimizer step ran on this model since (the forward's "
                               "state lives in the engine's workspace; call backward before the next bpr_loss)")
        ctl.materialise_grads(scale=grad_out.detach().reshape(1).to(torch.float32))
        return None, None


class StepController(object):
    def __init__(self, model):
        self.model = model
        self.engine = self.trainer = self.opt = None
        self.pending = None
        self.gen = 0
        self.master_newer = False        # the engine's master copy holds newer embeddings than the model's parameters
        self.grads_set = False           # some .grad may be non-None (zero_grad has work to do)
        self._seen = None                # version counters of the embedding parameters when the master copy was last loaded
        self.fast_steps = self.slow_steps = 0
        self.host_read_gen = None        # generation of the last loss the caller took to the host (PendingLoss)
        self.published_steps = 0

    # ------------------------------------------------------------------ wiring
    def adopt(self, params):
        for p in params:
            if isinstance(p, LazyGradParameter):
                p.__dict__["_elimrec_ctl"] = self

    def register_optimizer(self, opt):
        """FusedAdam(model.parameters()) announces itself: ONE parameter group over all of the model's parameters is what
        the fused step covers (anything else keeps working through materialised gradients)."""
        groups = opt.param_groups
        mine = set(id(p) for p in self.model.parameters())
        if len(groups) == 1 and set(id(p) for p in groups[0]["params"]) == mine:
            self.opt = opt
            if self.trainer is not None:
                self.trainer.set_optimizer(opt)

    def _embeddings(self):
        m = self.model
        return m.embedding_user.weight, m.embedding_item.weight

    def _versions(self):
        return tuple(p._version for p in self._embeddings())

    def ensure_engine(self):
        """The engine and its trainer, built on first use (the model is on its device by then)."""
        if self.engine is None:
            from .shard import ColumnShardEngine, ColumnShardTrainer
            import torch.distributed as dist
            world = rank = None
            if dist.is_available() and dist.is_initialized():
                world, rank = dist.get_world_size(), dist.get_rank()
            eng = ColumnShardEngine(self.model)
            self.trainer = ColumnShardTrainer(eng, self.opt, world_size=world or 1, rank=rank or 0)
            self.engine = eng
            self._seen = self._versions()
        return self.engine

    def attach(self, engine, trainer):
        """A trainer the caller built itself (main.py / bench.py with explicit world, rank, feature shards)."""
        self.engine, self.trainer = engine, trainer
        if trainer.opt is not None and self.opt is None:
            self.register_optimizer(trainer.opt)
        self._seen = self._versions()

    # ------------------------------------------------------------------ the four statements
    def begin(self, users, pos, neg):
        """bpr_loss: note the batch, hand out the loss."""
        self.settle()
        eng = self.ensure_engine()
        m = self.model
        if self._versions() != self._seen:
            # the parameters were written from outside (load_state_dict, a foreign optimizer): they are the truth
            eng.load_from_model()
            self._seen = self._versions()
            self.master_newer = False
        users, pos, neg = m._index_tensors(users, pos, neg)
        if not (users.numel() == pos.numel() == neg.numel()):
            raise ValueError("bpr_loss: users, pos_items, neg_items must have the same length")
        publish = self.host_read_gen is not None and self.host_read_gen == self.gen     # the loss before this one went to the host
        self.gen += 1
        if torch.is_grad_enabled():
            out = _DeferredBprFn.apply(self, m.embedding_user_after_GCN.weight)
        else:
            out = eng.new_loss_slot()
        slot = out.detach()                                      # same memory, no graph: what the step's launches write
        handle = out.as_subclass(PendingLoss)
        handle.__dict__["_elimrec_ctl"] = self
        self.pending = dict(gen=self.gen, users=users, pos=pos, neg=neg, slot=slot,
                            versions=(users._version, pos._version, neg._version),
                            ctx=None, bwd=False, grads=False, grad_ok=bool(out.requires_grad), publish=publish,
                            hdict=handle.__dict__)
        handle.__dict__["_elimrec_gen"] = self.gen
        return handle

    def _mine(self, handle):
        p = self.pending
        return p is not None and handle.__dict__.get("_elimrec_gen") == p["gen"]

    def _check_batch(self, p):
        if (p["users"]._version, p["pos"]._version, p["neg"]._version) != p["versions"]:
            raise RuntimeError("the index tensors given to bpr_loss were modified in place before the step ran (the step is "
                               "enqueued when the optimizer steps or when its loss / gradients are first read)")

    def realise_forward(self, handle=None):
        """Make the pending loss (and the cached tables of its forward) real: the forward half, launch by launch."""
        p = self.pending
        if p is None or p["ctx"] is not None or (handle is not None and not self._mine(handle)):
            return
        self._check_batch(p)
        p["ctx"] = self.trainer.forward_only(p["users"], p["pos"], p["neg"], loss=p["slot"])

    def request_backward(self, handle):
        """loss.backward(): True if taken as a request (the loss of the pending step, differentiable)."""
        p = self.pending
        if not self._mine(handle) or not p["grad_ok"]:
            return False
        p["bwd"] = True
        return True

    def cancel_backward(self):
        p = self.pending
        if p is not None and p["bwd"] and not p["grads"]:
            p["bwd"] = False

    def any_raw_grad(self):
        """Is any parameter's .grad set? (autograd's AccumulateGrad writes .grad without passing the Python setter that
        maintains grads_set: a regulariser on a parameter outside bpr_loss)"""
        ps = self.__dict__.get("_params")
        if ps is None:
            ps = self._params = list(self.model.parameters())
        for q in ps:
            if (_GRAD.__get__(q) if isinstance(q, torch.Tensor) else None) is not None:
                return True
        return False

    def grads_deferred(self):
        p = self.pending
        return p is not None and p["bwd"] and not p["grads"]

    @torch.no_grad()
    def materialise_grads(self, scale=None):
        """Forward + backward halves with the gradient table stored; every parameter the loss reaches gets its `.grad`
        (views of the flat gradient buffer; the embeddings' [N x d] gradient written back row-major)."""
        p = self.pending
        if p is None:
            return
        if p["grads"]:
            return
        self.realise_forward()
        m, eng, tr = self.model, self.engine, self.trainer
        if scale is not None and tr.world > 1:
            scale = scale / tr.world
        # gradients of an EARLIER backward() that are still in .grad and live in the flat buffer this pass is about to write
        # (they were handed out as views of it): moved out first, so that this pass's gradients are ADDED to them below
        gv0 = (m._ws or {}).get("grad_views") or {}
        for name, prm in m.named_parameters():
            have = prm._raw_grad() if isinstance(prm, LazyGradParameter) else prm.grad
            view = gv0.get(name)
            if have is not None and view is not None and have.data_ptr() == view.data_ptr():
                _GRAD.__set__(prm, have.clone())
        tr.backward_only(p["ctx"], grads_only=True, scale=scale)
        p["bwd"] = p["grads"] = True
        ws = m._ws
        grads = dict(eng._grads)
        if not m._lean:
            gx = ws["gX0d"]
            if tr.world == 1:
                eng.grad.to_rows(gx, col0=0)
            else:
                from .shard import _all_gather_parts
                gx.copy_(torch.cat(_all_gather_parts(eng.grad.dense(), tr.world, tr.group), dim=1))
            gv = ws["grad_views"]
            grads["embedding_user.weight"], grads["embedding_item.weight"] = gv["embedding_user.weight"], gv["embedding_item.weight"]
        for name, prm in m.named_parameters():
            g = grads.get(name)
            if g is None or not prm.requires_grad:
                continue
            have = prm._raw_grad() if isinstance(prm, LazyGradParameter) else prm.grad
            if have is None:
                prm.grad = g
            elif have.data_ptr() != g.data_ptr():
                have.add_(g)

    def settle(self):
        """A new bpr_loss (or an evaluation) while a step is pending and was never stepped: its loss must still become real
        (somebody holds the tensor), a requested backward must still leave its gradients."""
        p = self.pending
        if p is None:
            return
        if p["bwd"] and not p["grads"]:
            self.materialise_grads()
        elif p["ctx"] is None:
            self.realise_forward()
        self.pending = None

    def step(self, opt):
        """FusedAdam.step(): True if the controller ran the update (a backward was requested for the pending loss)."""
        p = self.pending
        if p is None or not p["bwd"]:
            return False
        if opt is not self.opt:
            self.register_optimizer(opt)
            if opt is not self.opt:
                self.materialise_grads()                 # not the fused step's optimizer: ordinary gradients, ordinary update
                return False
        if self.trainer.opt is not opt:
            self.trainer.set_optimizer(opt)
        eng, tr = self.engine, self.trainer
        if not p["grads"] and (self.grads_set or self.any_raw_grad()):
            # gradients of an earlier backward() are still in .grad (bpr_loss(b1).backward(); bpr_loss(b2).backward(); step()):
            # torch's update consumes their SUM -- this batch's gradients are added to them, the update reads .grad
            self.materialise_grads()
        if p["grads"]:
            # the gradients were materialised (and may have been edited): the update reads them where they are
            self._update_from_grads(opt)
            self.slow_steps += 1
        elif p["ctx"] is not None:
            tr.backward_only(p["ctx"])                   # the forward half has run: the backward half with the fused Adam
            self.slow_steps += 1
        else:
            self._check_batch(p)
            tr.step(p["users"], p["pos"], p["neg"], loss=p["slot"], publish=p["publish"])
            if eng.last_pub_seq is not None:
                p["hdict"]["_elimrec_pub"] = eng.last_pub_seq
                self.published_steps += 1
            self.fast_steps += 1
        self.pending = None
        return True

    @torch.no_grad()
    def _update_from_grads(self, opt):
        m, eng, tr = self.model, self.engine, self.trainer
        ws = m._ws
        gv = ws["grad_views"]
        have = {}
        for name, prm in m.named_parameters():
            g = prm._raw_grad() if isinstance(prm, LazyGradParameter) else prm.grad
            if g is None:
                continue
            view = gv.get(name)
            if view is not None and g.data_ptr() != view.data_ptr():
                view.copy_(g)                            # a replaced .grad tensor: back into the buffer the launches read
            have[name] = view
        if not m._lean and have.get("embedding_user.weight") is not None:
            gx = ws["gX0d"]
            eng.grad.from_rows(gx, col0=eng.col0)
        eng._grads = {k: v for k, v in have.items() if not k.startswith(("embedding_user.", "embedding_item."))}
        eng._adam_in_hop = eng._tail_in_hop = False
        eng.cs_update()

    @torch.no_grad()
    def sync_params(self, implicit=False):
        """Master copy -> the model's embedding parameters. With several ranks this is a collective (every rank holds a column
        slice): state_dict() / sync_params() must then be called by ALL ranks, and a read of the parameters that would need
        it silently on one rank raises instead of hanging the job."""
        if self.engine is not None and self.master_newer:
            if implicit and self.trainer is not None and self.trainer.world > 1:
                raise RuntimeError("the embedding tables are column-sharded over %d ranks and the engine holds newer values than "
                                   "the model's parameters: call model.state_dict() / model.plugin.sync_params() on ALL ranks "
                                   "before reading embedding_user / embedding_item" % self.trainer.world)
            self.engine.sync_to_model()

    def synced(self):
        self.master_newer = False
        self._seen = self._versions()

    def params_changed(self):
        """The embedding parameters were written by something that does not bump their version counters."""
        self._seen = None
