import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0], "5"]
import runpy
g = runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "step_trace.py"))
tr, u, p, n, B = g["tr"], g["u"], g["p"], g["n"], g["B"]
torch.cuda.synchronize()
N = 240
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 2)]
t0 = time.perf_counter()
ev[0].record()
for i in range(N):
    j = (i + 5) % (u.numel() // B)
    tr.step(u[j * B:(j + 1) * B], p[j * B:(j + 1) * B], n[j * B:(j + 1) * B])
    ev[i + 1].record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("wall %.1f us per step; GPU-side per-step, means of 10 (us):" % (1e6 * dt / N), " ".join("%.0f" % (ev[i].elapsed_time(ev[i + 10]) * 100) for i in range(0, N, 10)))
