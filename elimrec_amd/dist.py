"""Multi-GPU driver for the training step: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI), data-parallel over TRIPLETS with replicated tables.

Why replicas and not row shards at this shape: the whole propagated table is 115 MB at the Tiktok
shape (N=112 741 rows x 1 KiB) against 288 GB of HBM per GPU, and every hop of a row-sharded
propagation would move most of that table across xGMI (SURVEY.md §7 "xGMI volume") -- six times
per step. What actually differs between ranks is tiny: the gradient of the loss with respect to the
3B gathered head rows. So per step each rank
  1. runs the forward and the head's backward on its own B triplets (tables are bit-identical on every rank): loss,
     dOut rows of its active nodes, its share of the projection-weight gradients, all scaled by 1/world_size,
  2. all-gathers the [3B x 2d] source rows ([sum of dOut's column blocks | block 0]) + int32 node ids (3.1 MB per
     rank at B=2048) and all-reduces the span of the
     flat gradient buffer that holds the projection-weight gradients (0.3 MB),
  3. sums the gathered rows per node in rank order (elimrec_merge_rank_rows) and runs the SAME deterministic adjoint
     propagation + Adam.
(Engines without a sharded head backward -- the unfolded propagation paths -- all-gather node ids before the forward
and head-gradient rows after it, and run the whole backward on the gathered rows.)
Step 3 is bitwise identical on every rank (deterministic kernels, identical input order), so the
replicas never drift and no parameter/gradient all-reduce exists. The result equals one
single-GPU step with batch world_size*B (mean over the global batch).

The `engine` (EliMRec, or a CPU stand-in injected by tests/test_dist_cpu.py) provides
batch_keys / forward_local / backward_global / named_parameters.
"""

import torch
import torch.distributed as dist


class DataParallelTrainer(object):
    def __init__(self, engine, optimizer, world_size=1, rank=0, group=None, force_collectives=False):
        self.engine, self.opt, self.world, self.rank, self.group = engine, optimizer, int(world_size), int(rank), group
        # force_collectives: run the all-gather path even with one rank (exercises RCCL on a 1-GPU box)
        self.collectives = self.world > 1 or bool(force_collectives)
        self.profile_kernels = False
        self._events = []
        self._scale = None
        self._gather = None

    def step(self, users, pos, neg):
        """One training step on this rank's triplets; returns the (local) loss as a 0-dim tensor."""
        eng = self.engine
        if self.profile_kernels and getattr(eng, "_kernel_events", None) is None:
            eng._kernel_events = self._events
        if self.collectives and getattr(eng, "dp_shards_head", False):
            # forward and head backward on this rank's triplets only; the ranks exchange the dOut rows of their active
            # nodes (+ ids) and sum the projection-weight gradients; the adjoint propagation is replicated
            loss, _ = eng.forward_local(users, pos, neg, world_size=self.world)
            if self._scale is None:
                self._scale = torch.full((1,), 1.0 / self.world, dtype=torch.float32, device=loss.device)
            rows, keys, wgrads = eng.backward_local(self._scale)
            if self._gather is None or self._gather[0].shape[0] != self.world * rows.shape[0]:
                self._gather = (torch.empty(self.world * rows.shape[0], rows.shape[1], dtype=rows.dtype, device=rows.device),
                                torch.empty(self.world * keys.numel(), dtype=keys.dtype, device=keys.device))
            all_rows, all_keys = self._gather
            # in order (issued asynchronously, with the weight-gradient all-reduce under the merge and the adjoint propagation,
            # the extra stream hand-offs measured + 35 us per step on one GPU)
            dist.all_gather_into_tensor(all_rows, rows, group=self.group)
            dist.all_gather_into_tensor(all_keys, keys, group=self.group)
            dist.all_reduce(wgrads, op=dist.ReduceOp.SUM, group=self.group)
            grads = eng.backward_rows_global(all_rows, all_keys)
        elif self.collectives:
            keys = eng.batch_keys(users, pos, neg)
            if self._gather is None or self._gather[1].numel() != self.world * keys.numel():
                self._gather = None
                all_keys = torch.empty(self.world * keys.numel(), dtype=keys.dtype, device=keys.device)
            else:
                all_keys = self._gather[1]
            dist.all_gather_into_tensor(all_keys, keys, group=self.group)
            loss, grad_rows = eng.forward_local(users, pos, neg, all_keys=all_keys, rank=self.rank, world_size=self.world)
            if self._gather is None:
                self._gather = (torch.empty(self.world * grad_rows.shape[0], grad_rows.shape[1], dtype=grad_rows.dtype,
                                            device=grad_rows.device), all_keys)
                self._scale = torch.full((1,), 1.0 / self.world, dtype=torch.float32, device=grad_rows.device)
            all_rows = self._gather[0]
            dist.all_gather_into_tensor(all_rows, grad_rows, group=self.group)
            grads = eng.backward_global(all_rows, self._scale)
        else:
            loss, grad_rows = eng.forward_local(users, pos, neg)
            if self._scale is None:
                self._scale = torch.ones(1, dtype=torch.float32, device=grad_rows.device)
            grads = eng.backward_global(grad_rows, self._scale)
        for name, p in eng.named_parameters():
            p.grad = grads.get(name)          # None => the optimiser skips it (as torch does)
        self.opt.step()
        return loss

    def global_loss(self, loss):
        """Mean of the per-rank losses = loss of the global batch (all ranks hold equal B)."""
        if self.world > 1:
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
            loss /= self.world
        return loss

    def kernel_time_ms(self, name="spmm_hop"):
        """(total ms, launches) of the event-bracketed propagation kernels since profiling began."""
        total, launches = 0.0, 0
        for e0, e1, n in self._events:
            total += e0.elapsed_time(e1)
            launches += n
        return total, launches
