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
for W in [int(x) for x in os.environ.get("DP_WORLDS", "1,2,4,8").split(",")]:
    scale = torch.full((1,), 1.0 / W, device="cuda:0")
    gathered_rows = torch.randn(W * 3 * B, 2 * model.latent_dim, device="cuda:0") * 1e-4
    gathered_keys = torch.zeros(W * 3 * B, dtype=torch.int32, device="cuda:0")
    filled = [False]
    def step(i):
        sl = slice(i * B, (i + 1) * B)
        loss, _ = model.forward_local(U_[sl], P_[sl], N_[sl], world_size=W)
        if W == 1:
            grads = model.backward_global(model._ws["grad_rows"], scale)
        else:
            rows, keys, wg = model.backward_local(scale)
            # stand-in for the collectives: W-1 other ranks' rows / ids (different nodes; values irrelevant for timing)
            all_rows = gathered_rows; all_rows[:rows.shape[0]].copy_(rows)
            all_keys = gathered_keys
            R = keys.numel()
            all_keys[:R].copy_(keys)
            if not filled[0]:                         # other ranks: the same node ids shifted, sorted, negative padding
                valid = keys[keys >= 0]
                for r in range(1, W):
                    kr = torch.sort((valid + 977 * r) % (ds.num_users + ds.num_items))[0].to(torch.int32)
                    all_keys[r * R:r * R + kr.numel()].copy_(kr)
                    all_keys[r * R + kr.numel():(r + 1) * R].fill_(-1)
                filled[0] = True
            grads = model.backward_rows_global(all_rows, all_keys)
    for i in range(3): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 30
    for i in range(K): step(3 + i)
    torch.cuda.synchronize()
    print("W=%d  %.3f ms/step (compute only, incl. the stand-in concat)" % (W, (time.perf_counter() - t0) * 1e3 / K))
