#!/usr/bin/env python3
"""BASELINE.md section 3 calibration, build container only (needs /root/reference): times the REFERENCE's own training
step (main.py:98-101 around models/EliMRec.py:115-142, imported through the shims of tests/golden/make_golden.py) and
the oracle's step (oracle/elimrec_oracle.py) on the same Tiktok-shape synthetic data, same threads, and prints the
ratio. The oracle is accepted as "the reference CPU path" on the GPU box if it is within +-20 % here.
    python tools/calibrate_cpu_baseline.py [steps]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
U, I, E, D, d, B = 36656, 76085, 720829, 128, 64, 2048
from elimrec_amd import SyntheticDataset
ds = SyntheticDataset(U, I, E, feat_dims=(D, D, D), seed=0)
w = mg.prepare_copy()
mg.install_shims()
dd = os.path.join(w, "dataset")
os.makedirs(dd, exist_ok=True)
for split, mat in (("train", ds.train_matrix), ("valid", ds.valid_matrix), ("test", ds.test_matrix)):
    coo = mat.tocoo()
    np.savetxt(os.path.join(dd, "movielens.%s" % split), np.stack([coo.row, coo.col], 1), fmt="%d", delimiter=",")
for key, fn in (("v", "FeatureVideo_normal"), ("a", "FeatureAudio_avg_normal"), ("t", "FeatureText_stl_normal")):
    np.save(os.path.join(dd, "movielens_%s.npy" % fn), getattr(ds, key + "_feat").numpy())
os.chdir(w)
sys.path.insert(0, w)
sys.argv = ["main.py", "--recommender=EliMRec", "--data.input.dataset=movielens", "--alpha=0.5", "--loss=bpr_loss",
            "--batch_size=%d" % B, "--verbose=0", "--save_flag=False", "--recdim=%d" % d]
import torch
torch.set_num_threads(os.cpu_count())
from util.configurator import Configurator
from util.tool import set_seed
args = Configurator("./NeuRec.properties", default_section="hyperparameters")
set_seed(args["seed"])
import main, tqdm
main.tqdm = tqdm.tqdm
t0 = time.time()
net = main.Net(args)
rec, opt = net.recommender, net.opt
print("reference built in %.1f s: U %d I %d, %d params" % (time.time() - t0, net.dataset.num_users, net.dataset.num_items,
                                                           sum(p.numel() for p in rec.parameters())))
rs = np.random.RandomState(1)
nU, nI = net.dataset.num_users, net.dataset.num_items
def batch():
    return (torch.from_numpy(rs.randint(0, nU, B)).long(), torch.from_numpy(rs.randint(0, nI, B)).long(),
            torch.from_numpy(rs.randint(0, nI, B)).long())
def ref_step(u, p, n):
    loss = rec.bpr_loss(u, p, n)
    opt.zero_grad()
    loss.backward(retain_graph=True)
    opt.step()
    return loss.cpu().item()
ref_step(*batch())
tr = []
for _ in range(steps):
    t0 = time.time(); ref_step(*batch()); tr.append(time.time() - t0)
print("reference step: %s s" % ["%.2f" % t for t in tr])
# the oracle on the reference's own tensors
from oracle import elimrec_oracle as eo
adj = eo.build_adj(*net.dataset.get_train_interactions(), nU, nI, args["adj_type"])
feats = {"v": rec.v_feat, "a": rec.a_feat, "t": rec.t_feat.detach()}
om = eo.OracleEliMRec(nU, nI, d, args["layer_num"], adj, feats, {k: v.detach().numpy() for k, v in rec.state_dict().items()}, 0.5)
oopt = eo.OracleAdam(om.params, lr=args["lr"], weight_decay=args["weight_decay"])
eo.train_step(om, oopt, *batch())
to = []
for _ in range(steps):
    t0 = time.time(); eo.train_step(om, oopt, *batch()); to.append(time.time() - t0)
print("oracle step:    %s s" % ["%.2f" % t for t in to])
print("RESULT threads %d torch %s: reference %.2f s/step, oracle %.2f s/step, oracle/reference = %.2f"
      % (torch.get_num_threads(), torch.__version__, np.mean(tr), np.mean(to), np.mean(to) / np.mean(tr)))
