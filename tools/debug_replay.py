import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import load_golden, build_model_from_fixture
from elimrec_amd import FusedAdam
DEV = torch.device("cuda:0")
g = load_golden("ml3")
def run(adam_fast):
    model, cfg = build_model_from_fixture(g, DEV)
    model._use_replay = adam_fast
    opt = FusedAdam(model.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    out = []
    for t in range(1, int(g["steps"]) + 1):
        u, p, n = (torch.from_numpy(g["step%d/%s" % (t, k)]).to(DEV) for k in ("users", "pos", "neg"))
        loss = model.bpr_loss(u, p, n)
        opt.zero_grad()
        loss.backward(retain_graph=True)
        gr = {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}
        opt.step()
        out.append((loss.item(), gr, {k: v.clone() for k, v in model.state_dict().items()}))
    return out
a = run(True)
os.environ["X"]="1"
b = run(False)
for t, (x, y) in enumerate(zip(a, b)):
    print("step", t + 1, "loss", x[0], y[0], "want", float(g["step%d/loss" % (t + 1)]))
    for k in x[1]:
        d = (x[1][k] - y[1][k]).abs().max().item()
        if d > 0: print("   grad diff", k, d)
    for k in x[2]:
        d = (x[2][k] - y[2][k]).abs().max().item()
        if d > 0: print("   param diff", k, d)
