"""Determinism soak of the device evaluator at the Tiktok shape: n validation passes in the reference tie order (ref_order_kernel:
one wave per user replaying libstdc++'s heap in LDS) after 2 and after 40 training steps -- every pass must return the first
pass's per-user metric rows bit for bit, and the scorers' range counter must stay zero. usage: eval_soak.py [passes]  (dev tool)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg, ds, model = bench.build(None, "cuda:0")
model = model.to("cuda:0")
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
u, p, q = PairwiseSamplerV2(ds, batch_size=2048, device="cuda:0").sample_epoch()
model.predict_type = "TIE"
ev = model.valid_evaluator.evaluator
users = list(ev.user_pos_test.keys())
done = 0
for steps in (2, 38):
    for i in range(done, done + steps):
        tr.step(u[i * 2048:(i + 1) * 2048], p[i * 2048:(i + 1) * 2048], q[i * 2048:(i + 1) * 2048])
    done += steps
    for order in ("reference", "id"):
        ev.tie_order = order
        first = ev.metric_rows(model, users, cached=True).clone()
        bad = 0
        for k in range(n):
            rows = ev.metric_rows(model, users, cached=True)
            bad += int(not torch.equal(rows, first))
        print("after %d steps, tie_order=%s: %d of %d passes differ from the first; range violations %d, scorer mismatch rows %d"
              % (done, order, bad, n, ev.range_violations, ev.scorer_mismatch_rows), flush=True)
