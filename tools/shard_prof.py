"""Steps of the slab trainer for rocprofv3 (--kernel-trace --stats). usage: shard_prof.py W [steps]  (W ranks emulated)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import ColumnShardEngine, Configurator, EliMRec, FusedAdam, Logger, PairwiseSamplerV2, SyntheticDataset, set_seed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
dev = "cuda:0"
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                   argv=["x", "--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0"])
Logger.logger = Logger(show_in_console=False)
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(128, 128, 128), seed=0)
B = 2048
u, p, n = PairwiseSamplerV2(ds, batch_size=B, device=dev).sample_epoch()
engines = []
for q in range(W):
    set_seed(1)
    model = EliMRec(cfg, ds).to(dev)
    eng = ColumnShardEngine(model)
    eng.cs_setup(W, q, FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"]))
    engines.append(eng)
scale = torch.full((1,), 1.0 / W, device=dev)
for i in range(K):
    bt = [(u[(i * W + q) * B:(i * W + q + 1) * B], p[(i * W + q) * B:(i * W + q + 1) * B], n[(i * W + q) * B:(i * W + q + 1) * B]) for q in range(W)]
    acts = torch.stack([e.cs_plan(*b).clone() for e, b in zip(engines, bt)])
    sends = [e.cs_forward(acts) for e in engines]
    sends = [None if s is None else s.clone() for s in sends]
    s2s, wgs = [], []
    for q, e in enumerate(engines):
        e.cs_head(None if W == 1 else torch.stack([sends[q2][q] for q2 in range(W)]))
        s2, wg = e.cs_backward_local(scale)
        s2s.append(s2.clone()); wgs.append(wg)
    for q, e in enumerate(engines):
        e.cs_backward_hops(torch.stack([s2s[q2][q] for q2 in range(W)]), acts)
        e.cs_update()
torch.cuda.synchronize()
print("done", W, K)
