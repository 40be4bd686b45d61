"""K steps of the trainer at the Tiktok shape, then a bitwise fingerprint of losses, parameters and Adam moments: the
launch-form switches (ELIMREC_HEAD_SOURCES / FUSE_MERGE / FUSE_REDUCE / FUSE_BWDW / FUSE_ADAM / AUX_STREAM) must not change it.
usage: step_checksum.py [steps]"""
import hashlib, os, sys, torch
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["60"])
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_trace.py"))
tr, model = g["tr"], g["model"]
tr.engine.sync_to_model()
h = hashlib.sha256()
for k, v in sorted(model.state_dict().items()):
    h.update(v.detach().cpu().numpy().tobytes())
st = tr.engine.optimizer_state()
h.update(st["exp_avg"].cpu().numpy().tobytes()); h.update(st["exp_avg_sq"].cpu().numpy().tobytes())
print("fingerprint", h.hexdigest()[:16])
