"""bench.py's headline region step by step: a fresh trainer, 5 warm-up steps, then 20 steps with an event behind each (dev tool)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
cfg, ds, model = bench.build(None, "cuda:0")
model = model.to("cuda:0")
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
B = 2048
u, p, n = PairwiseSamplerV2(ds, batch_size=B, device="cuda:0", seed=cfg["seed"]).sample_epoch()
W, K = int(os.environ.get("W", 5)), 20
batches = [(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B]) for i in range(W + K)]
tr.prestage(batches)
for i in range(W):
    tr.step(*batches[i])
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(K):
    tr.step(*batches[W + i])
    ev[i + 1].record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("wall %.4f ms per step; per step (us):" % (dt * 1e3 / K), " ".join("%.0f" % (ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(K)))
print("native:", tr._native_state()["native_steps"], "of", W + K, "steps; failed:", tr._native_state()["failed"])
