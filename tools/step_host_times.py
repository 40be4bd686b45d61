"""Host issue time of each of the first steps after a device synchronize (is the start of a timed region host-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "5"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_trace.py"))
tr, u, p, n, B = g["tr"], g["u"], g["p"], g["n"], g["B"]
for rep in range(2):
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for i in range(24):
        j = (i + 5) % (u.numel() // B)
        tr.step(u[j * B:(j + 1) * B], p[j * B:(j + 1) * B], n[j * B:(j + 1) * B])
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    print("host issue per step (us):", " ".join("%.0f" % (1e6 * (ts[i + 1] - ts[i])) for i in range(24)), "| total with sync %.0f" % (1e6 * (time.perf_counter() - ts[0]) / 24))
