"""Validation-pass time of the device evaluator at the Tiktok shape: best of n passes per setting (tie order id / reference). Dev tool."""
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
model.predict_type = "TIE"
ev = model.valid_evaluator.evaluator
for order in ("id", "reference"):
    ev.tie_order = order
    ts = []
    for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); res, buf = model.evaluate(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("tie_order=%s: best %.5f s, median %.5f s  (%s)" % (order, min(ts), sorted(ts)[len(ts) // 2], buf.replace("\t", " ")[:60]))
from elimrec_amd import _lib
print("scorer cross-check: rows checked %d, mismatch rows %d, bf16x3 scorer on: %d" % (ev.scorer_checked_rows, ev.scorer_mismatch_rows, int(_lib.load().elimrec_score_get_bf16x3())))
