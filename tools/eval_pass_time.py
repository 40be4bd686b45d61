"""Seconds of a full-catalogue TIE validation pass at the Tiktok shape (bench.py's eval line alone). Dev tool."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "2"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_trace.py"))
model, tr = g["model"], g["tr"]
tr.engine.sync_to_model()
model.predict_type = "TIE"
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t = time.perf_counter()
    model.evaluate()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("validation pass: %.4f s (best of 4 after the first: %s)" % (min(ts[1:]), " ".join("%.4f" % x for x in ts)))
