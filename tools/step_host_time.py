"""Host time to ISSUE one training step of the column-shard trainer (dev tool): after a device synchronise, K steps are
enqueued back to back and the clock stops when the last call returns -- with K small enough that the HIP queue never
fills, that is pure host time. ELIMREC_NATIVE_STEP=0 gives the Python-issued path for comparison."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2
if os.environ.get("ELIMREC_SHARD_MULTI") == "1":       # the multi-rank step over a one-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
cfg, ds, model = bench.build(None, "cuda:0")
model = model.to("cuda:0")
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
eng = ColumnShardEngine(model, feature_shard=os.environ.get("FSHARD", "replicated"))
tr = ColumnShardTrainer(eng, opt)
B = 2048
s = PairwiseSamplerV2(ds, batch_size=B, device="cuda:0", seed=1)
U, P, N = s.sample_epoch()
bs = [(U[i * B:(i + 1) * B], P[i * B:(i + 1) * B], N[i * B:(i + 1) * B]) for i in range(200)]
tr.plan_lookup(bs)
for b in bs[:20]:
    tr.step(*b)
torch.cuda.synchronize()
st = tr._native_state()
print("native:", st["on"], "failed:", st["failed"], "native steps so far:", st["native_steps"])
for K in (4, 8, 16, 32):
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in bs[20 + rep * K:20 + (rep + 1) * K]:
            tr.step(*b)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, (t1 - t0) / K)
    print("K=%2d: host issue %.1f us per step (GPU step %.1f us)" % (K, best * 1e6, (t2 - t0) / K * 1e6))
