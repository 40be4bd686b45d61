"""Shuffle + batch over parallel sequences with the reference's semantics
(util/data_iterator.py:44-58,86-110,145-152,158-210): `np.random.permutation(n)` when shuffling,
chunks of `batch_size`, a short final batch unless `drop_last`, columns returned as python lists
(a single sequence yields a flat list)."""
import numpy as np


class DataIterator(object):
    def __init__(self, *data, batch_size=1, shuffle=False, drop_last=False):
        data = list(data)
        for column in data:
            if len(column) != len(data[0]):
                raise ValueError("The length of the given data are not equal!")
        if not isinstance(batch_size, int) or isinstance(batch_size, bool) or batch_size <= 0:
            raise ValueError("batch_size should be a positive integeral value, but got batch_size={}".format(batch_size))
        if not isinstance(drop_last, bool):
            raise ValueError("drop_last should be a boolean value, but got drop_last={}".format(drop_last))
        self.data, self.batch_size, self.shuffle, self.drop_last = data, batch_size, shuffle, drop_last

    def _num_samples(self):
        return len(self.data[0]) if self.data else 0

    def __len__(self):
        n = self._num_samples()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self._num_samples()
        order = np.random.permutation(n).tolist() if self.shuffle else range(n)
        order = list(order)
        for start in range(0, n, self.batch_size):
            idx = order[start:start + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            columns = [[column[i] for i in idx] for column in self.data]
            yield columns[0] if len(columns) == 1 else columns
