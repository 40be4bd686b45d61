"""Micro-benchmark of elimrec_head_bwd_input at several active-row counts (dev tool)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import ops
dev = "cuda:0"
U, I, d, M, S = 36656, 76085, 64, 4, 3
C, Cy, N = d * M, (1 + S) * d, U + I
g = torch.Generator(device=dev).manual_seed(0)
Wu, Wi = torch.randn(d, C, device=dev, generator=g), torch.randn(d, C, device=dev, generator=g)
Wh = [torch.randn(d, d, device=dev, generator=g) for _ in range(S)]
G0 = torch.zeros(N, C, device=dev)
for R in (512, 2048, 6144, 16384, 49152):
    act = torch.sort(torch.randperm(N, device=dev, generator=g)[:R]).values.to(torch.int32)
    seg = torch.tensor([R, int((act < U).sum()), 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
    dY = torch.randn(R, Cy, device=dev, generator=g)
    for _ in range(3):
        ops.head_bwd_input(dY, act, seg, U, d, C, [1, 2, 3], Wu, Wi, Wh, 1.0, G0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.head_bwd_input(dY, act, seg, U, d, C, [1, 2, 3], Wu, Wi, Wh, 1.0, G0)
    e1.record(); torch.cuda.synchronize()
    print("R=%6d  %.1f us" % (R, e0.elapsed_time(e1) * 1e3 / 20))
