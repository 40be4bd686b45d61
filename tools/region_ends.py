"""Where a 20-step timed region's fixed cost sits: wall clock between the two synchronisations against the GPU's own time between an
event recorded in front of the first step and one behind the last (dev tool; the step_trace.py set-up)."""
import os, sys, time, torch
sys.argv = [sys.argv[0], "60"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_trace.py"))
tr, batches, nb = g["tr"], g["batches"], g["nb"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for K in (20, 20, 300, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(K):
        tr.step(*batches[i % nb])
    t_enq = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("K=%d: wall %.4f ms per step, GPU (event to event) %.4f ms per step; host done enqueueing after %.0f us; wall - GPU = %.0f us per region" % (
        K, (t1 - t0) * 1e3 / K, e0.elapsed_time(e1) / K, (t_enq - t0) * 1e6, (t1 - t0) * 1e6 - e0.elapsed_time(e1) * 1e3))
