import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for _ in range(3): tr.step(u, p, n)
torch.cuda.synchronize()
for ptype in ("TIE", "TE"):
    model.predict_type = ptype
    t0 = time.time(); res, buf = model.evaluate(); torch.cuda.synchronize(); dt = time.time() - t0
    nu = len(model.valid_evaluator.evaluator.user_pos_test)
    print(ptype, "valid users", nu, "time %.2f s" % dt, "users/s %.0f" % (nu / dt), buf)
