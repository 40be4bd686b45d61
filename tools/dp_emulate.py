"""Per-step GPU time of what ONE rank of a W-rank data-parallel job computes, emulated on one GPU:
local forward on B triplets + backward/Adam on W*3B gathered gradient rows (comm excluded). Dev tool."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from elimrec_amd import FusedAdam, PairwiseSamplerV2
cfg, ds, model = bench.build(None, "cuda:0")
model = model.to("cuda:0")
opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
B = 2048
smp = PairwiseSamplerV2(ds, batch_size=B, device="cuda:0", seed=1)
U_, P_, N_ = smp.sample_epoch()
for W in (1, 2, 4, 8):
    scale = torch.full((1,), 1.0 / W, device="cuda:0")
    def step(i):
        gk = model.batch_keys(U_[i * B:(i + 1) * B], P_[i * B:(i + 1) * B], N_[i * B:(i + 1) * B])
        # stand-in for the all-gathers: W-1 other ranks' keys / rows (different nodes; values are irrelevant for timing)
        all_keys = torch.cat([gk] + [(gk + 977 * (r + 1)) % (ds.num_users + ds.num_items) for r in range(W - 1)]).to(torch.int32) if W > 1 else gk
        loss, gr = model.forward_local(U_[i * B:(i + 1) * B], P_[i * B:(i + 1) * B], N_[i * B:(i + 1) * B], all_keys=all_keys,
                                       rank=0, world_size=W)
        all_rows = gr.repeat(W, 1) if W > 1 else gr
        grads = model.backward_global(all_rows, scale)
        for name, p in model.named_parameters():
            p.grad = grads.get(name)
        opt.step()
    for i in range(3): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 30
    for i in range(K): step(3 + i)
    torch.cuda.synchronize()
    print("W=%d  %.3f ms/step (compute only, incl. the stand-in concat)" % (W, (time.perf_counter() - t0) * 1e3 / K))
