"""Scores of the evaluator's two math modes side by side (EXACT = IEEE division + libm expf, FAST = v_exp / v_rcp + Newton
step): largest absolute / relative difference and the rows whose top-K lists differ. Dev tool."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from elimrec_amd import _lib, ops
lib = _lib.load(); DEV = "cuda:0"
U, I, d, S, K = 130, 40000, 64, 3, 20
g = torch.Generator().manual_seed(9)
for scale in (0.4, 1.5):
    Y = (torch.randn(U + I, (1 + S) * d, generator=g) * scale).to(DEV)
    users = torch.arange(0, 128).to(DEV)
    ws = torch.empty(ops.score_workspace(128, U, I, S, K), dtype=torch.uint8, device=DEV)
    for mode in ("rubi", "hm", "sum"):
        for ptype in ("normal", "TE", "TIE"):
            out = {}
            for fast in (0, 1):
                lib.elimrec_score_set_math(fast)
                sc = torch.empty(128, I, device=DEV)
                idx = torch.empty(128, K, dtype=torch.int32, device=DEV)
                ops.score_topk(Y, U, I, users, d, S, 0b111, mode, ptype, ws, scores=sc, K=K, topk_idx=idx)
                out[fast] = (sc.double().cpu().numpy(), idx.cpu().numpy())
            e, f = out[0][0], out[1][0]
            print(scale, mode, ptype, "max abs %.2e  max rel %.2e  rows with different top-K %d" % (np.abs(e - f).max(), (np.abs(e - f) / np.maximum(np.abs(e), 1e-30)).max(), int((out[0][1] != out[1][1]).any(1).sum())))
lib.elimrec_score_set_math(1)
