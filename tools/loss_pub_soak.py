"""Soak of the published loss (loss publication): 4000 steps of the
reference loop body with the loss read on a random 60% of steps, kept for a much
later read on 10% (the slot ring has wrapped by then) and not read on the rest.
Checks every read is finite, the late reads equal the device tensors and no step
left the native program.  GPU only."""
import os, sys, time, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench, torch
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2
device = torch.device("cuda", 0)
cfg, ds, model = bench.build(None, device)
model = model.to(device)
B = 2048
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
U_, P_, N_ = PairwiseSamplerV2(ds, batch_size=B, device=device, seed=1).sample_epoch()
nb = U_.numel() // B
batches = [(U_[i*B:(i+1)*B], P_[i*B:(i+1)*B], N_[i*B:(i+1)*B]) for i in range(nb)]
random.seed(0)
held, vals = [], []
t0 = time.time()
for k in range(4000):
    u, p, n = batches[k % nb]
    loss = model.bpr_loss(u, p, n); opt.zero_grad(); loss.backward(retain_graph=True); opt.step()
    mode = random.random()
    if mode < 0.6:
        vals.append((k, loss.cpu().item()))
    elif mode < 0.7:
        held.append((k, loss))                     # read much later (ring wrapped)
torch.cuda.synchronize()
late = [(k, h.item()) for k, h in held[:200]]
ctl = model.plugin
print("steps 4000 in %.1f s, published %d, fast %d slow %d; finite %s; late reads %d finite %s" % (
    time.time() - t0, ctl.published_steps, ctl.fast_steps, ctl.slow_steps, all(v == v for _, v in vals), len(late), all(v == v for _, v in late)))
# late reads equal the device tensors
print("late reads equal device values:", all(abs(h.detach().cpu().item() - v) == 0 for (k, h), (_, v) in zip(held[:200], late)))
