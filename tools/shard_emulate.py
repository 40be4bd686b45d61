"""Per-rank GPU time of a column-sharded step, W ranks emulated on one GPU (collectives done by hand, excluded where
possible): total time of the W ranks' kernels / W. Dev tool. usage: shard_emulate.py [W ...]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, Configurator, EliMRec, FusedAdam, Logger, PairwiseSamplerV2, SyntheticDataset, set_seed
from elimrec_amd.dist import DataParallelTrainer
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
dev = "cuda:0"
d = int(os.environ.get("RECDIM", "64"))
cfg = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                   argv=["x", "--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=%d" % d, "--verbose=0"])
Logger.logger = Logger(show_in_console=False)
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(128, 128, 128), seed=0)
B = 2048
smp = PairwiseSamplerV2(ds, batch_size=B, device=dev)
u, p, n = smp.sample_epoch()

def make(world, rank):
    set_seed(1)
    model = EliMRec(cfg, ds).to(dev)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    eng = ColumnShardEngine(model)
    eng.cs_setup(world, rank, opt)
    return eng

def emulated_step(engines, batches, regions):
    W = len(engines)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    marks = [ev() for _ in range(6)]
    marks[0].record()
    acts = torch.stack([e.cs_plan(*b).clone() for e, b in zip(engines, batches)])
    marks[1].record()
    sends = [e.cs_forward(acts) for e in engines]
    marks[2].record()
    sends = [None if s is None else s.clone() for s in sends]
    scale = torch.full((1,), 1.0 / W, device=dev)
    recvs = [None if W == 1 else torch.stack([sends[q2][q] for q2 in range(W)]) for q in range(W)]
    torch.cuda.synchronize()
    t_head = 0.0
    sends2, wgs = [], []
    e0, e1 = ev(), ev()
    e0.record()
    for q, e in enumerate(engines):
        e.cs_head(recvs[q])
        s2, wg = e.cs_backward_local(scale)
        sends2.append(s2); wgs.append(wg)
    e1.record()
    sends2 = [s.clone() for s in sends2]
    recv2 = [torch.stack([sends2[q2][q] for q2 in range(W)]) for q in range(W)]
    e2, e3 = ev(), ev()
    e2.record()
    for q, e in enumerate(engines):
        e.cs_backward_hops(recv2[q], acts)
    e3.record()
    e4 = ev()
    for q, e in enumerate(engines):
        e.cs_update()
    e4.record()
    torch.cuda.synchronize()
    regions["plan"] += marks[0].elapsed_time(marks[1]) / W
    regions["fwd hops+rows"] += marks[1].elapsed_time(marks[2]) / W
    regions["head fwd+bwd"] += e0.elapsed_time(e1) / W
    regions["merge+adjoint hops"] += e2.elapsed_time(e3) / W
    regions["adam"] += e3.elapsed_time(e4) / W

worlds = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
for W in worlds:
    engines = [make(W, q) for q in range(W)]
    K = 10
    regions = None
    for i in range(3 + K):
        if i == 3:
            regions = {k: 0.0 for k in ("plan", "fwd hops+rows", "head fwd+bwd", "merge+adjoint hops", "adam")}
        batches = [(u[(i * W + q) * B:(i * W + q + 1) * B], p[(i * W + q) * B:(i * W + q + 1) * B], n[(i * W + q) * B:(i * W + q + 1) * B]) for q in range(W)]
        emulated_step(engines, batches, regions if regions is not None else {k: 0.0 for k in ("plan", "fwd hops+rows", "head fwd+bwd", "merge+adjoint hops", "adam")})
    tot = sum(regions.values()) / K
    print("W=%d dl=%d (ns=%d w=%d gs=%d): per-rank %.3f ms/step  [%s]" % (W, engines[0].dl, engines[0].ns, engines[0].w, engines[0].gs, tot,
          ", ".join("%s %.0f us" % (k, 1e3 * v / K) for k, v in regions.items())))
    del engines
    torch.cuda.empty_cache()
# reference points: the row-major trainer and the slab trainer at world 1, wall clock
for kind in ("rows", "slab"):
    set_seed(1)
    model = EliMRec(cfg, ds).to(dev)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    tr = DataParallelTrainer(model, opt) if kind == "rows" else ColumnShardTrainer(ColumnShardEngine(model), opt)
    for i in range(5): tr.step(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 50
    for i in range(5, 5 + K): loss = tr.step(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B])
    torch.cuda.synchronize()
    print("%s trainer, world 1: %.3f ms/step wall, loss %.5f" % (kind, (time.perf_counter() - t0) / K * 1e3, float(loss)))
