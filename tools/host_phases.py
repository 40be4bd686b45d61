"""Host time of each phase of the training step (wall time of the issuing calls, no device sync inside the step).
usage: [ELIMREC_SHARD_MULTI=1] host_phases.py [steps]"""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sys.argv = [sys.argv[0], "40"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_trace.py"))
tr, u, p, n, B = g["tr"], g["u"], g["p"], g["n"], g["B"]
acc = collections.OrderedDict()


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return w


for k in list(tr._ph):
    tr._ph[k] = timed(k, tr._ph[k])
for k in ("_all_gather", "_all_to_all", "_all_reduce_async"):
    setattr(tr, k, timed(k, getattr(tr, k)))
nb = u.numel() // B
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    j = i % nb
    tr.step(u[j * B:(j + 1) * B], p[j * B:(j + 1) * B], n[j * B:(j + 1) * B])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per step: issue %.1f us, with final sync %.1f us" % (1e6 * t_issue / K, 1e6 * t_all / K))
for k, v in acc.items():
    print("  %-20s %6.1f us" % (k, 1e6 * v / K))
print("  %-20s %6.1f us" % ("(rest of step)", 1e6 * (t_issue - sum(acc.values())) / K))
