"""Host (Python + ctypes) time per training step vs GPU time per step (dev tool)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from elimrec_amd import FusedAdam
from elimrec_amd.dist import DataParallelTrainer
cfg, ds, model = bench.build(None, "cuda:0")
model = model.to("cuda:0")
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
tr = DataParallelTrainer(model, opt)
g = torch.Generator(device="cuda:0").manual_seed(0)
u = torch.randint(0, ds.num_users, (2048,), device="cuda:0", generator=g)
p = torch.randint(0, ds.num_items, (2048,), device="cuda:0", generator=g)
n = torch.randint(0, ds.num_items, (2048,), device="cuda:0", generator=g)
for _ in range(5): tr.step(u, p, n)
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K): tr.step(u, p, n)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per step: %.3f ms; total per step: %.3f ms" % ((t1 - t0) * 1e3 / K, (t2 - t0) * 1e3 / K))
