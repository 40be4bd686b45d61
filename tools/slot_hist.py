"""How many slots of a batch list the same node (the member lists the head backward walks): the bench's data set and sampler.
usage (GPU box): python tools/slot_hist.py   (dev tool)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import PairwiseSamplerV2, SyntheticDataset
ds = SyntheticDataset(36656, 76085, 720829, feat_dims=(4, 4, 4), seed=0)
u, p, n = PairwiseSamplerV2(ds, batch_size=2048, device="cuda:0").sample_epoch()
for b in range(3):
    sl = slice(b * 2048, (b + 1) * 2048)
    keys = torch.cat([u[sl], 36656 + p[sl], 36656 + n[sl]])
    cnt = torch.bincount(keys).float()
    cnt = cnt[cnt > 0]
    top = torch.sort(cnt, descending=True)[0][:8].int().tolist()
    print("batch %d: %d active rows of 6144 slots; most-listed rows %s; rows listed more than 4 times: %d" % (b, cnt.numel(), top, int((cnt > 4).sum())))
