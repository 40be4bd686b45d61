"""Micro-benchmark of the batch-row projection launches (small-M linear_fwd_batched)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elimrec_amd import ops

dev = torch.device("cuda:0")
N, R, d = 112741, 6144, 64
torch.manual_seed(0)
S = [torch.randn(N, 128, device=dev) for _ in range(3)]
W = [torch.randn(d, 128, device=dev) for _ in range(3)]
b = [torch.randn(d, device=dev) for _ in range(3)]
c = torch.randn(N, device=dev)
narrow = torch.randn(N, d, device=dev)
rows = torch.randint(0, N, (R,), device=dev, dtype=torch.int32)
out = torch.empty(R, 4 * d, device=dev)
Sc = [s[:R].contiguous() for s in S]


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = {
    "gather+rowscale+add": lambda: ops.linear_fwd_batched([(S[k], W[k], b[k], out[:, (k + 1) * d:(k + 2) * d], c, narrow, rows) for k in range(3)]),
    "gather only": lambda: ops.linear_fwd_batched([(S[k], W[k], b[k], out[:, (k + 1) * d:(k + 2) * d], None, None, rows) for k in range(3)]),
    "contiguous plain": lambda: ops.linear_fwd_batched([(Sc[k], W[k], b[k], out[:, (k + 1) * d:(k + 2) * d]) for k in range(3)]),
    "contiguous 1 problem": lambda: ops.linear_fwd_batched([(Sc[0], W[0], b[0], out[:, d:2 * d])]),
    "empty-ish (M=64)": lambda: ops.linear_fwd_batched([(Sc[0][:64], W[0], b[0], out[:64, d:2 * d])]),
}
for name, fn in cases.items():
    print("%-28s %7.2f us" % (name, timeit(fn)))
