import os, sys, time, torch
sys.path.insert(0, ".")
from elimrec_amd import ops
U, I, d, S, K = 2000, 76085, int(os.environ.get("RD", "128")), 3, 10
Y = torch.randn(U + I, (1 + S) * d, device="cuda") * 0.2
users = torch.arange(1024, device="cuda")
ws = torch.empty(ops.score_workspace(1024, U, I, S, K), dtype=torch.uint8, device="cuda")
idx = torch.empty(1024, K, dtype=torch.int32, device="cuda")
for _ in range(3): ops.score_topk(Y, U, I, users, d, S, 7, "rubi", "TIE", ws, K=K, topk_idx=idx)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): ops.score_topk(Y, U, I, users, d, S, 7, "rubi", "TIE", ws, K=K, topk_idx=idx)
torch.cuda.synchronize(); print("recdim %d T16=%s: %.2f ms per 1024-user block" % (d, os.environ.get("ELIMREC_SCORE_T16", "1"), (time.perf_counter() - t) * 100))
