"""Random small shapes / layer counts / batch sizes: one training step + predict vs the CPU oracle (dev tool; needs oracle/)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for case in range(n_cases):
    U, I = int(rs.randint(20, 1500)), int(rs.randint(30, 3000))
    E = int(rs.randint(max(U, I) * 2, max(U, I) * 12))
    dims = tuple(int(4 * rs.randint(1, 40)) for _ in range(3))
    recdim = int(rs.choice([16, 32, 64, 128]))
    L = int(rs.randint(1, 5))
    B = int(rs.randint(1, 1200))
    adj = str(rs.choice(["pre", "plain", "gcmc", "norm"]))
    try:
        model, om = T._full_shape_step(U, I, E, dims, recdim, B, "synthetic", extra_argv=["--layer_num=%d" % L, "--adj_type=%s" % adj])
        users = list(range(0, U, max(1, U // 30)))[:32]
        model.predict_type = om.predict_type = "TIE"
        err = np.abs(model.predict(users).numpy() - om.predict(users).numpy()).max()
        # 'plain' (unnormalised) adjacency lets the embeddings grow by ~degree per hop: fp32 summation-order noise of
        # the logits is then visible in the scores (the step itself is still checked to loss 1e-5 / gradients 1e-4 rel)
        ok = err < (1e-4 if adj == "plain" else 1e-5)
    except AssertionError as e:
        ok, err = False, str(e)[:80]
    bad += (not ok)
    print("%s U=%d I=%d E=%d dims=%s d=%d L=%d B=%d adj=%s lazy=%s err=%s" % ("ok " if ok else "BAD", U, I, E, dims, recdim, L, B, adj,
          getattr(model, "_lazy", None), err), flush=True)
print("failures:", bad)
