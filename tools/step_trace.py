"""Steps of the real trainer for rocprofv3 --kernel-trace (bench.py's training loop without the extras). usage: step_trace.py [steps]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, Configurator, EliMRec, FusedAdam, Logger, PairwiseSamplerV2, SyntheticDataset, set_seed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
dev = "cuda:0"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                   argv=["x", "--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=64", "--verbose=0"])
Logger.logger = Logger(show_in_console=False)
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(128, 128, 128), seed=0)
B = int(os.environ.get("B", 2048))
u, p, n = PairwiseSamplerV2(ds, batch_size=B, device=dev).sample_epoch()
if os.environ.get("ELIMREC_SHARD_MULTI") == "1":       # the multi-rank step over a one-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
set_seed(1)
model = EliMRec(cfg, ds).to(dev)
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
eng = ColumnShardEngine(model, feature_shard=os.environ.get("FEATURE_SHARD") or None)
tr = ColumnShardTrainer(eng, opt)
if tr.lookup and tr.multi:                               # row-sharded constants: the split sizes planned ahead, as main.py does per epoch
    pass
nb = int(u.numel()) // B                                 # whole batches of the sampled epoch; more steps go round again
batches = [(u[j * B:(j + 1) * B], p[j * B:(j + 1) * B], n[j * B:(j + 1) * B]) for j in range(nb)]
if tr.lookup and tr.multi:                               # row-sharded constants: the split sizes planned ahead, as main.py does per epoch
    tr.plan_lookup(batches)
if os.environ.get("PRESTAGE", "1") == "1":               # the epoch's triplets are resident: planners may run ahead (as main.py / bench.py say)
    tr.prestage(batches)
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K):
    j = i % nb
    tr.step(*batches[j])
torch.cuda.synchronize()
print("done: %.4f ms per step over %d steps (the first ones traced / warm-up included); native steps %d, failed: %s" % (
    1e3 * (time.perf_counter() - t0) / K, K, tr._native_state()["native_steps"], tr._native_state()["failed"]))
