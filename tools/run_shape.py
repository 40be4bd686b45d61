"""Run a few training steps + one eval block at an arbitrary synthetic shape (dev tool).
usage: run_shape.py U I E d [D]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import Configurator, EliMRec, FusedAdam, Logger, PairwiseSamplerV2, SyntheticDataset, set_seed
from elimrec_amd.dist import DataParallelTrainer
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U, I, E, d = (int(x) for x in sys.argv[1:5])
D = int(sys.argv[5]) if len(sys.argv) > 5 else 128
os.chdir(ROOT)
cfg = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                   argv=["x", "--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=%d" % d, "--verbose=0"])
Logger.logger = Logger(show_in_console=False)
set_seed(1)
t0 = time.time()
ds = SyntheticDataset(U, I, E, feat_dims=(D, D, D), seed=0)
print("dataset %.1f s; train nnz %d" % (time.time() - t0, ds.train_matrix.nnz))
t0 = time.time()
model = EliMRec(cfg, ds).to("cuda:0")
print("model build %.1f s; params %.1f M" % (time.time() - t0, sum(p.numel() for p in model.parameters()) / 1e6))
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
tr = DataParallelTrainer(model, opt)
smp = PairwiseSamplerV2(ds, batch_size=2048, device="cuda:0")
u, p, n = smp.sample_epoch()
B = 2048
losses = []
for i in range(3): losses.append(tr.step(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B]))
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for i in range(3, 3 + K): losses.append(tr.step(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B]))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("step %.3f ms  (%.0f triplets/s); loss %.5f -> %.5f; mem %.1f GB" % (dt * 1e3, B / dt, float(losses[0]), float(losses[-1]),
      torch.cuda.max_memory_allocated() / 1e9))
model.predict_type = "TIE"
users = list(model.valid_evaluator.evaluator.user_pos_test.keys())[:128]
t0 = time.perf_counter()
rows = model.valid_evaluator.evaluator.evaluate_batch(model, users)
torch.cuda.synchronize()
print("eval block of 128 users: %.2f ms; recall@10 %.5f" % ((time.perf_counter() - t0) * 1e3, float(rows[:, 19].mean())))
