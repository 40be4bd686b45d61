"""Host-side cost of ColumnShardTrainer.step (cProfile) and whether the step is host- or GPU-bound. Dev tool."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, Configurator, EliMRec, FusedAdam, Logger, PairwiseSamplerV2, SyntheticDataset, set_seed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
dev = "cuda:0"
cfg = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                   argv=["x", "--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0"])
Logger.logger = Logger(show_in_console=False)
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(128, 128, 128), seed=0)
B = 2048
u, p, n = PairwiseSamplerV2(ds, batch_size=B, device=dev).sample_epoch()
set_seed(1)
model = EliMRec(cfg, ds).to(dev)
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
bt = [(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B]) for i in range(260)]
for i in range(10): tr.step(*bt[i])
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for i in range(10, 10 + K): tr.step(*bt[i])
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.3f ms/step; until GPU idle %.3f ms/step" % (t_host / K * 1e3, t_all / K * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(10, 60): tr.step(*bt[i])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
