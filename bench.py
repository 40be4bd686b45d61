#!/usr/bin/env python3
"""BPR triplets/s of the EliMRec training step on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of B triplets already resident in HBM:
forward (table assembly + feature projections + L-hop propagation + head Linears + cosine-BPR),
backward (deterministic scatter-add, head/propagation/projection gradients) and the dense Adam
update -- the tables are re-propagated every step, as the reference does (main.py:98-101).

Workload at N=1: BASELINE.json configs[1], synthetic Tiktok shape (|U|=36 656, |I|=76 085,
720 829 interactions, 128-d V/A/T features, recdim 64, 3 layers, B=2048), fp32.
N>1: one process per GPU (torch.distributed / RCCL), data-parallel over triplets with replicated
tables -- each rank takes its own B triplets, the row-sparse head gradients are all-gathered, and
every rank applies the identical update (elimrec_amd/dist.py). Weak scaling: per-GPU batch fixed.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(name="tiktok-shape-synthetic", num_users=36656, num_items=76085, num_interactions=720829,
                feat_dims=(128, 128, 128), recdim=64, layer_num=3, batch_size=2048, alpha=0.5)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes(model, B):
    """SURVEY.md §8(d) closed form (fp32: s = s_f = 4)."""
    s = 4
    U, I, d, M, L = model.num_users, model.num_items, model.latent_dim, model.M, model.n_layers
    N = U + I
    nnz = int(model.adj_val.numel())
    T = N * M * d * s
    G = N * d * s
    Ccsr = 8 * nnz + 4 * (N + 1)
    F = I * sum(getattr(model, m + "_feat").shape[1] for m in model._mods) * s
    P = sum(p.numel() for p in model.parameters())
    step = 2 * F + 2 * (M - 1) * I * d * s + 2 * (L * (2 * T + Ccsr) + (L + 2) * T) + 2 * (T + G) + 4 * (M - 1) * G \
        + 9 * B * M * d * s + 28 * P
    # one propagation hop as the kernels run it: all M tables side by side (bipartite / full paths), or -- with
    # the constant feature tables folded into GEMM operands -- only the id table: d columns
    Th = G if getattr(model, "_folded", False) else T
    hop = (L * (2 * Th + Ccsr) + (L + 2) * Th) / max(L, 1)    # layer-mean traffic included
    return step, hop


def pmc_traffic_per_hop():
    """HBM bytes per propagation hop from the committed PMC passes (profiles/r01_h_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs of this same command,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950). None if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_h_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["propagation_hop_traffic_bytes"]
    except Exception:
        return None


def build(args, device):
    import torch
    from elimrec_amd import Configurator, EliMRec, FusedAdam, SyntheticDataset, set_seed
    w = WORKLOAD
    cfg = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                       argv=["bench.py", "--data.input.dataset=synthetic", "--alpha=%r" % w["alpha"], "--loss=bpr_loss",
                             "--recdim=%d" % w["recdim"], "--layer_num=%d" % w["layer_num"],
                             "--batch_size=%d" % w["batch_size"], "--verbose=0"])
    os.chdir(ROOT)
    set_seed(cfg["seed"])
    ds = SyntheticDataset(w["num_users"], w["num_items"], w["num_interactions"], feat_dims=w["feat_dims"], seed=0)
    from elimrec_amd import Logger
    Logger.logger = Logger(show_in_console=False)
    model = EliMRec(cfg, ds)
    return cfg, ds, model


def cpu_baseline(ds, model_cpu_state, cfg, batches, thread_counts=(8, 16, 32)):
    """The oracle (CPU restatement of the reference step, pinned to the reference by
    tests/test_oracle_golden.py) timed on this box's host cores on a bounded sample: one warm-up
    step, then one full training step per candidate thread count; the fastest is reported
    (torch.sparse.mm, 84 % of the reference's step, stops scaling long before 256 threads)."""
    import torch
    from oracle import elimrec_oracle as eo
    ncpu = os.cpu_count() or 1
    tu, ti = ds.get_train_interactions()
    adj = eo.build_adj(tu, ti, ds.num_users, ds.num_items, cfg["adj_type"])
    feats = {m: eo.OracleEliMRec.normalize_features(getattr(ds, m + "_feat")) for m in ("v", "a", "t")}
    om = eo.OracleEliMRec(ds.num_users, ds.num_items, cfg["recdim"], cfg["layer_num"], adj, feats, model_cpu_state,
                          cfg["alpha"])
    opt = eo.OracleAdam(om.params, lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    B = len(batches[0][0])
    counts = sorted(set(min(c, ncpu) for c in thread_counts))
    torch.set_num_threads(counts[0])
    eo.train_step(om, opt, *batches[0])            # warm-up (allocations, thread pool)
    best = None
    timings = {}
    for k, c in enumerate(counts):
        torch.set_num_threads(c)
        t0 = time.time()
        eo.train_step(om, opt, *batches[1 + k % (len(batches) - 1)])
        dt = time.time() - t0
        timings[c] = round(dt, 3)
        if best is None or dt < best[1]:
            best = (c, dt)
    return dict(value=B / best[1], unit="triplets/s", cores=best[0], kind="port", host_cpus=ncpu,
                sample="1 warm-up + 1 full training step (B=%d, same workload) per thread count %s; seconds per step: %s"
                       % (B, counts, timings), ms_per_step=1e3 * best[1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-eval", action="store_true", help="skip the secondary evaluator timing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs a torch.distributed.run launch with %d ranks (WORLD_SIZE=%d)"
                         % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_coll = os.environ.get("ELIMREC_FORCE_COLLECTIVES", "0") == "1" and "RANK" in os.environ
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg, ds, model = build(args, device)
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    from elimrec_amd import FusedAdam, PairwiseSamplerV2
    from elimrec_amd.dist import DataParallelTrainer
    B = WORKLOAD["batch_size"]
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    trainer = DataParallelTrainer(model, opt, world_size=world, rank=rank, force_collectives=force_coll)

    # triplets for every step, sampled on the device and resident in HBM before the timed region
    total = args.warmup + args.steps
    sampler = PairwiseSamplerV2(ds, batch_size=B, device=device, seed=cfg["seed"] + rank)
    pools = [sampler.sample_epoch()]
    while sum(p[0].numel() for p in pools) < total * B:
        pools.append(sampler.sample_epoch())
    U_, P_, N_ = (torch.cat([p[i] for p in pools]) for i in range(3))
    batches = [(U_[i * B:(i + 1) * B], P_[i * B:(i + 1) * B], N_[i * B:(i + 1) * B]) for i in range(total)]

    def sync():
        if world > 1 or force_coll:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.step(*batches[i])
    sync()
    trainer.profile_kernels = True            # HIP events around the dominant kernel, on the launch stream
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        loss = trainer.step(*batches[i])
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    hop_ms, hop_launches = trainer.kernel_time_ms("propagation_hop")
    final_loss = float(loss.item())

    if rank == 0:
        step_bytes, hop_bytes = algorithmic_bytes(model, B)
        achieved = hop_bytes / (hop_ms * 1e-3 / hop_launches) / 1e9 if hop_launches else None
        out = {
            "metric": "BPR triplets/sec (Tiktok-shape, d=128x3)", "value": B * world * args.steps / dt,
            "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD["name"], "num_users": ds.num_users, "num_items": ds.num_items,
                       "train_interactions": int(ds.train_matrix.nnz), "feat_dims": list(WORKLOAD["feat_dims"]),
                       "recdim": WORKLOAD["recdim"], "layer_num": WORKLOAD["layer_num"], "batch_per_gpu": B,
                       "global_batch": B * world, "parallelism": "dp%d-replicated-tables" % world,
                       "propagation": "folded" if getattr(model, "_folded", False) else
                                      ("bipartite" if getattr(model, "_bipartite", False) else "full"),
                       "head_rows": "batch" if getattr(model, "_lazy", False) else "all",
                       "final_loss": final_loss},
            "step_algorithmic_GB": step_bytes / 1e9,
            "step_achieved_GBps": step_bytes / (dt / args.steps) / 1e9,
            "roofline": {"bound": "hbm",
                         "kernel": ("propagation hop of the d-column table [E_u;E_i] = ONE half_hop_kernel<16> launch over the "
                                    "full adjacency; 2L hops per step; feature tables folded into GEMM operands"
                                    if getattr(model, "_folded", False) else
                                    "propagation hop = half_hop_kernel<64> (C columns) + half_hop_kernel<16> (d columns); "
                                    "2L hops per step"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": pmc_traffic_per_hop(),
                         "algorithmic_bytes_per_launch": hop_bytes,
                         "avg_launch_us": 1e3 * hop_ms / hop_launches if hop_launches else None,
                         "launches_timed": hop_launches},
        }
        if world == 1 and not args.no_eval:
            # secondary metric of SURVEY 8(d): full-catalogue TIE top-K validation pass on the device evaluator
            # (after the timed region; the first pass also materialises the full cached tables and the user blocks)
            model.predict_type = "TIE"
            secs = []
            for _ in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.evaluate()
                torch.cuda.synchronize()
                secs.append(time.perf_counter() - t1)
            n_eval = len(model.valid_evaluator.evaluator.user_pos_test)
            topks = cfg["topks"]
            out["eval"] = {"what": "full-catalogue TIE top-%d validation pass" % (max(topks) if isinstance(topks, (list, tuple)) else int(topks)),
                           "users": n_eval, "seconds_first": secs[0], "seconds": secs[1], "users_per_s": n_eval / secs[1]}
        if not args.no_cpu_baseline and world == 1:     # the host baseline is timed on rank 0 at N=1 only
            cpu_batches = [tuple(x.cpu() for x in b) for b in batches[:5]]
            out["cpu_baseline"] = cpu_baseline(ds, {k: v.cpu().numpy() for k, v in init_state.items()}, cfg, cpu_batches)
        print(json.dumps(out))
    if world > 1 or force_coll:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
