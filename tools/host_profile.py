"""cProfile of the host side of the training step (dev tool)."""
import cProfile, os, pstats, sys, torch
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
pr = cProfile.Profile()
pr.enable()
for _ in range(100): tr.step(u, p, n)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
