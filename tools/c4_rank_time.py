"""BASELINE.json configs[3] (Tiktok shape x16 items, |I| = 1 217 360, recdim 128, B = 2048 per rank): GPU time per step of
ONE rank of a W-rank column-sharded job on one MI355X. W = 1 is the real trainer; for W > 1 rank 0's engine runs every
kernel of its step at the real sizes, with its own send buffers fed back as the peers' (the values are then meaningless,
the work is not): what a rank computes between the collectives. usage: c4_rank_time.py [W ...]  (dev tool)"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, Configurator, EliMRec, FusedAdam, Logger, PairwiseSamplerV2, SyntheticDataset, set_seed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
dev = "cuda:0"
small = os.environ.get("SMALL") == "1"
U, I, E, d = (36656, 76085, 720829, 64) if small else (36656, 1217360, 16 * 720829, 128)
cfg = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                   argv=["x", "--data.input.dataset=synthetic", "--alpha=0.5", "--loss=bpr_loss", "--recdim=%d" % d, "--verbose=0"])
Logger.logger = Logger(show_in_console=False)
t0 = time.time()
ds = SyntheticDataset(U, I, E, feat_dims=(128, 128, 128), seed=0)
B = 2048
u, p, n = PairwiseSamplerV2(ds, batch_size=B, device=dev).sample_epoch()
print("data set %.0f s: U=%d I=%d train nnz=%d, recdim %d" % (time.time() - t0, U, I, ds.train_matrix.nnz, d), flush=True)
K = 30


class _Done:
    def wait(self):
        pass


for W in [int(x) for x in sys.argv[1:]] or [1, 8]:
    set_seed(1)
    t0 = time.time()
    model = EliMRec(cfg, ds).to(dev)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    eng = ColumnShardEngine(model)
    if W == 1:
        tr = ColumnShardTrainer(eng, opt)
        step = lambda i: tr.step(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B])
    else:
        eng.cs_setup(W, 0, opt)
        eng.multi_aux = True                              # the trainer's second stream (planner, row bitmap, source bits under the hops)
        scale = torch.full((1,), 1.0 / W, device=dev)

        def step(i):
            act = eng.cs_plan(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B])
            ps = eng.plan_stream()
            if ps is not None:                            # the planner ran on the second stream: the copy below reads its list
                torch.cuda.current_stream().wait_stream(ps)
            acts = act.view(1, -1).expand(W, -1).contiguous()
            eng.cs_gathered_ids(acts, _Done())            # second stream, as the trainer does behind the id exchange: row bitmap + source bits
            eng.cs_forward_hops()
            if not eng._long_wanted_only():
                eng.cs_forward_long()                     # every split row, ahead of the waits (graphs with few split rows)
            send = eng.cs_forward_rows(acts)
            eng.cs_head(send)                         # own slices in place of the peers'
            s2, wg = eng.cs_backward_local(scale)
            eng.cs_backward_hops(s2, acts)
            eng.cs_update()
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    print("W=%d set-up %.0f s (dl=%d, slabs %dx%d in %d groups)" % (W, time.time() - t0, eng.dl, eng.ns, eng.w, eng.gs), flush=True)
    t1 = time.perf_counter()
    for i in range(4, 4 + K):
        step(i)
    torch.cuda.synchronize()
    print("W=%d: %.3f ms of GPU work per rank-step (wall clock over %d steps, collectives excluded)" % (W, (time.perf_counter() - t1) / K * 1e3, K), flush=True)
    del model, opt, eng
    torch.cuda.empty_cache()
