"""Fused head forward alone, with phases switched off (ELIMREC_HEAD_DBG bits) -- where do its microseconds go. Dev tool."""
import os, subprocess, sys
if len(sys.argv) == 1:       # the ELIMREC_HEAD_DBG phase switches this tool used were removed from the kernel after the ablation
    for rows in ("16", "32"):
        subprocess.run([sys.executable, __file__, rows], env=dict(os.environ, ELIMREC_HEAD_ROWS=rows))
    sys.exit(0)
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import ColumnShardEngine, Configurator, EliMRec, FusedAdam, Logger, PairwiseSamplerV2, SyntheticDataset, set_seed
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
eng = ColumnShardEngine(model)
eng.cs_setup(1, 0, FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"]))
acts = eng.cs_plan(u[:B], p[:B], n[:B]).view(1, -1)
eng.cs_forward(acts)
ws = model._ws
def run():
    eng._head_forward_fused(ws, 3 * B, B)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("%s-row tiles: %.1f us per head forward + BPR head (+ loss sum)" % (sys.argv[1], e0.elapsed_time(e1) * 1e3 / 50))
