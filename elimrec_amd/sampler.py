"""Epoch-level pairwise sampler with the interface of the reference's `PairwiseSamplerV2`
(data/sampler.py:297-351): iterate -> (users, pos_items, neg_items) batches, `len()` = number of
batches. Sampling itself runs on the GPU (csrc/sampler.hip): the whole epoch of
`num_trainings` triplets is drawn in one launch and the batches are device-tensor views, so the
training loop does no per-batch host->device copy.

Contract (tests/test_hip_parity.py::test_sampler_contract_on_device): users uniform with replacement over users that have >= 1
training item; positives uniform over that user's training items; negatives uniform over the
catalogue minus the user's training items. The draws are i.i.d., so the reference's extra
`shuffle` pass is a distributional no-op and is not repeated. The random stream is Philox keyed
by (seed, epoch) -- the reference's libc rand() stream cannot be reproduced on a GPU and is
never seeded there either (random_choice.pyx:8).
"""
import numpy as np
import torch

from . import ops


class PairwiseSamplerV2(object):
    def __init__(self, dataset, neg_num=1, batch_size=1024, shuffle=True, drop_last=False, device=None, seed=2022,
                 shard=None):
        """shard = (rank, world): this process draws its 1/world share of the epoch's `num_trainings` triplets (rounded
        up, so every rank runs the same number of batches); give every rank its own seed."""
        if neg_num <= 0:
            raise ValueError("'neg_num' must be a positive integer.")
        if neg_num != 1:
            raise NotImplementedError("neg_num > 1 is not used by the EliMRec driver (main.py:59)")
        self.batch_size, self.drop_last, self.shuffle, self.neg_num = batch_size, drop_last, shuffle, neg_num
        self.item_num = dataset.num_items
        user_pos = dataset.get_user_train_dict()
        if not isinstance(user_pos, dict):
            raise TypeError("'user_pos_dict' must be a dict.")
        if not user_pos:
            raise ValueError("'user_pos_dict' cannot be empty.")
        self.num_trainings = sum(len(v) for v in user_pos.values())
        if shard is not None:
            rank, world = int(shard[0]), int(shard[1])
            if not (0 <= rank < world):
                raise ValueError("shard=(rank, world) with 0 <= rank < world")
            self.num_trainings = (self.num_trainings + world - 1) // world
        users = np.fromiter(user_pos.keys(), dtype=np.int32, count=len(user_pos))
        counts = np.fromiter((len(user_pos[u]) for u in users), dtype=np.int64, count=len(users))
        if int(counts.max()) >= self.item_num:
            raise ValueError("The number of 'exclusion' is greater than 'high'.")
        self._users = users
        self._ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self._items = np.concatenate([np.sort(np.asarray(user_pos[u], dtype=np.int32)) for u in users])
        self.device = device
        self.seed = int(seed)
        self.epoch = 0
        self._dev = None

    def __len__(self):
        n = self.num_trainings
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _to_device(self):
        device = self.device if self.device is not None else torch.device("cuda:0")
        if torch.device(device).type != "cuda":
            raise RuntimeError("PairwiseSamplerV2 samples on the GPU; device '%s' has no implementation" % device)
        if self._dev is None:
            self._dev = (torch.from_numpy(self._users).to(device), torch.from_numpy(self._ptr).to(device),
                         torch.from_numpy(self._items).to(device))
        return device

    def sample_epoch(self):
        """One epoch of triplets as three device int64 tensors of length num_trainings."""
        device = self._to_device()
        n = self.num_trainings
        u = torch.empty(n, dtype=torch.int64, device=device)
        p = torch.empty_like(u)
        q = torch.empty_like(u)
        ops.sample_triplets(self._dev[0], self._dev[1], self._dev[2], self.item_num, n, self.seed, self.epoch, u, p, q)
        self.epoch += 1
        return u, p, q

    def __iter__(self):
        u, p, q = self.sample_epoch()
        n, bs = self.num_trainings, self.batch_size
        for start in range(0, n, bs):
            stop = min(start + bs, n)
            if stop - start < bs and self.drop_last:
                return
            yield u[start:stop], p[start:stop], q[start:stop]
