"""Host vs total time per step through the reference-style API (bpr_loss / backward / optimizer.step). Dev tool."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from elimrec_amd import FusedAdam
cfg, ds, model = bench.build(None, "cuda:0")
model = model.to("cuda:0")
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
g = torch.Generator(device="cuda:0").manual_seed(0)
u = torch.randint(0, ds.num_users, (2048,), device="cuda:0", generator=g)
p = torch.randint(0, ds.num_items, (2048,), device="cuda:0", generator=g)
n = torch.randint(0, ds.num_items, (2048,), device="cuda:0", generator=g)
def step():
    loss = model.bpr_loss(u, p, n)
    opt.zero_grad()
    loss.backward(retain_graph=True)
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("autograd API: host enqueue per step %.3f ms; total per step %.3f ms" % ((t1 - t0) * 1e3 / K, (t2 - t0) * 1e3 / K))
