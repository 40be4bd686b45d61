"""cProfile of the host side of the training step (which Python / ctypes / torch.distributed calls the issue time goes to).
usage: [ELIMREC_SHARD_MULTI=1] host_profile.py [steps]"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sys.argv = [sys.argv[0], "40"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_trace.py"))
tr, u, p, n, B = g["tr"], g["u"], g["p"], g["n"], g["B"]
nb = u.numel() // B
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    j = i % nb
    tr.step(u[j * B:(j + 1) * B], p[j * B:(j + 1) * B], n[j * B:(j + 1) * B])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("ROWS", "22")))
