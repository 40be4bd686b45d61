"""Smoke test of the REAL column-shard trainer with RCCL collectives when only one GPU is available: two processes,
both on cuda:0 (ELIMREC_SAME_GPU=1 makes bench.py map every rank to device 0). RCCL may refuse duplicate devices; then
this prints the error and exits 0 -- the multi-process path is covered by gloo on CPU (tests/test_dist_cpu.py). Dev tool."""
import os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
env = dict(os.environ, ELIMREC_SAME_GPU="1", NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
       "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"]
r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
print(r.stdout[-6000:])
print(r.stderr[-9000:])
