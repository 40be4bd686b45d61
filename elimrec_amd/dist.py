"""Data-parallel driver for the UNFOLDED row-major forms of the step (--propagation=full | bipartite, --head_rows=all,
layer_num < 2, an adjacency with a diagonal without --propagation=folded): one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI), data-parallel over TRIPLETS with replicated tables. The default form of the step runs on the
column-shard engine instead (shard.py, plugin.py); this driver is what main.py falls back to for the shapes the engine does
not take, and what bench.py's "reference-equivalent work" line times.

What differs between ranks is tiny: the gradient of the loss with respect to the 3B gathered head rows. Per step each rank
  1. all-gathers the int32 node ids of its triplet slots and runs the forward on its own B triplets (tables are
     bit-identical on every rank): loss and the [3B x Cy] head-gradient rows,
  2. all-gathers those rows,
  3. runs the whole backward on the gathered rows (scaled by 1/world_size) and the SAME deterministic Adam.
Step 3 is bitwise identical on every rank (deterministic kernels, identical input order), so the replicas never drift and
no parameter/gradient all-reduce exists. The result equals one single-GPU step with batch world_size*B (mean over the
global batch).

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
        if self.collectives:
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
