import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0], "5"]
import runpy
g = runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "step_trace.py"))
tr, u, p, n, B = g["tr"], g["u"], g["p"], g["n"], g["B"]
nb = u.numel() // B
def run(N, tag):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        j = i % nb
        tr.step(u[j * B:(j + 1) * B], p[j * B:(j + 1) * B], n[j * B:(j + 1) * B])
        ev[i + 1].record()
    torch.cuda.synchronize()
    print(tag, " ".join("%.0f" % (ev[i].elapsed_time(ev[i + 4]) * 250) for i in range(0, min(N, 40), 4)))
torch.cuda.synchronize()
run(300, "first 40 steps after start-up (means of 4):")
run(40, "40 steps right after a synchronize:")
time.sleep(0.05)
run(40, "40 steps after 50 ms of idle:")
time.sleep(0.002)
run(40, "40 steps after 2 ms of idle:")
