import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
cfg, ds, model = bench.build(None, "cuda:0")
model = model.to("cuda:0")
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
u, p, n = PairwiseSamplerV2(ds, batch_size=2048, device="cuda:0").sample_epoch()
for i in range(3): tr.step(u[i * 2048:(i + 1) * 2048], p[i * 2048:(i + 1) * 2048], n[i * 2048:(i + 1) * 2048])
model.predict_type = "TIE"
users = list(model.valid_evaluator.evaluator.user_pos_test.keys())[:512]
sc = torch.empty(len(users), model.num_items, device="cuda:0")
model.predict_device(users, scores=sc)
I = model.num_items
K = 10
pilot = sc[:, :2048]
thr = pilot.topk(K, dim=1).values[:, -1]
print("scores: min %.6f max %.6f; thr after pilot: mean %.6f" % (float(sc.min()), float(sc.max()), float(thr.mean())))
tiles = sc[:, 2048:2048 + 16384].reshape(len(users), -1, 16).max(dim=2).values
print("fraction of (user, tile) pairs of the next chunk with max >= thr: %.4f" % float((tiles >= thr[:, None]).float().mean()))
thr2 = sc[:, :18432].topk(K, dim=1).values[:, -1]
tiles2 = sc[:, 18432:18432 + 16384].reshape(len(users), -1, 16).max(dim=2).values
print("... of the chunk after: %.4f" % float((tiles2 >= thr2[:, None]).float().mean()))
print("distinct values among a user's scores: %d of %d" % (int(torch.unique(sc[0]).numel()), I))
print("top 12 of user 0:", sc[0].topk(12).values.tolist())
