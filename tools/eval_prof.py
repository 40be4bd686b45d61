"""Two validation passes of the device evaluator at the Tiktok shape, for rocprofv3 --kernel-trace --stats. Dev tool."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
cfg, ds, model = bench.build(None, "cuda:0")
model = model.to("cuda:0")
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
u, p, n = PairwiseSamplerV2(ds, batch_size=2048, device="cuda:0").sample_epoch()
for i in range(3): tr.step(u[i * 2048:(i + 1) * 2048], p[i * 2048:(i + 1) * 2048], n[i * 2048:(i + 1) * 2048])
torch.cuda.synchronize()
if os.environ.get("TIE_ORDER"):
    model.valid_evaluator.evaluator.tie_order = os.environ["TIE_ORDER"]
for k in range(3):
    t0 = time.time(); res, buf = model.evaluate(); torch.cuda.synchronize(); dt = time.time() - t0
    print("pass %d: %d users, %.4f s" % (k, len(model.valid_evaluator.evaluator.user_pos_test), dt))
