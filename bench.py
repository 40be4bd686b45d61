#!/usr/bin/env python3
"""BPR triplets/s of the EliMRec training step on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of B triplets already resident in HBM: forward (L-hop propagation
of the folded table + layer means + feature/fusion/head projections at the batch's rows + cosine-BPR), backward
(deterministic scatter-add, head / projection gradients, adjoint propagation) and the dense Adam update -- the tables
are re-propagated every step, as the reference does (main.py:98-101).

Workload at N=1: BASELINE.json configs[1], synthetic Tiktok shape (|U|=36 656, |I|=76 085, 720 829 interactions,
128-d V/A/T features, recdim 64, 3 layers, B=2048), fp32.
N>1: one process per GPU (torch.distributed / RCCL), column-sharded (elimrec_amd/shard.py): rank q owns recdim/N
columns of [E_u ; E_i], of its gradient and Adam moments and runs every hop on its slice without communication; only
the layer means / adjoint sources of the batch's active rows cross xGMI (two all-to-alls + one all-gather of ids + the
projection-weight all-reduce per step). Weak scaling: B triplets per GPU.

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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")      # before the HIP runtime initialises (elimrec_amd/__init__.py has the measurements)

WORKLOAD = dict(name="tiktok-shape-synthetic", num_users=36656, num_items=76085, num_interactions=720829,
                feat_dims=(128, 128, 128), recdim=64, layer_num=3, batch_size=2048, alpha=0.5)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector peak
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)


def reference_step_bytes(model, B):
    """SURVEY.md 8(d) closed form: bytes the REFERENCE algorithm moves per step (fp32: s = s_f = 4)."""
    s = 4
    U, I, d, M, L = model.num_users, model.num_items, model.latent_dim, model.M, model.n_layers
    N = U + I
    nnz = int(model.adj_val.numel())
    T, G = N * M * d * s, N * d * s
    Ccsr = 8 * nnz + 4 * (N + 1)
    F = I * sum(getattr(model, m + "_feat").shape[1] for m in model._mods) * s
    P = sum(p.numel() for p in model.parameters())
    return 2 * F + 2 * (M - 1) * I * d * s + 2 * (L * (2 * T + Ccsr) + (L + 2) * T) + 2 * (T + G) + 4 * (M - 1) * G \
        + 9 * B * M * d * s + 28 * P


def step_bytes(model, eng, B, world):
    """Bytes THIS step has to move per rank: the hops read a table, the index stream once per
    slab group and write a table; the adjoint's first hop reads only the index stream and writes a table; Adam moves
    28 B per owned parameter; everything else touches the <= 3B active rows only."""
    s = 4
    U, I, d, M, L = model.num_users, model.num_items, model.latent_dim, model.M, model.n_layers
    N, dl = U + I, eng.dl
    G = N * dl * s
    idx = eng.plan.index_bytes()
    R = 3 * B
    hop = 2 * G + eng.gs * idx                       # one full hop
    fwd = (L - 1) * hop + G + eng.gs * 8 * eng.plan.sell_seg_entries      # hop L: the split rows + the active rows only
    bwd = (G + eng.gs * idx) + (L - 1) * hop         # first adjoint hop: row-sparse source
    P_tail = sum(p.numel() for n_, p in model.named_parameters() if not n_.startswith(("embedding_user.", "embedding_item.")))
    adam = 28 * (N * dl + P_tail)
    if eng._fuse_adam():                              # Adam is the last adjoint hop's epilogue: the gradient table is neither
        adam -= 2 * G                                 # written by the hop nor read back by the optimizer
    D = sum(getattr(model, m + "_feat").shape[1] for m in model._mods)
    rows = R * s * ((L + 1) * dl + 2 * D + 6 * model.C + 6 * model.Cy + 4 * d)     # layer rows, folded constants, Out/Y rows fwd+bwd, sources
    return dict(total=fwd + bwd + adam + rows, hop=hop, hop_minimal=2 * G + idx)


def build(args, device, extra_argv=()):
    import torch
    from elimrec_amd import Configurator, EliMRec, Logger, SyntheticDataset, set_seed
    w = WORKLOAD
    cfg = Configurator(os.path.join(ROOT, "NeuRec.properties"), default_section="hyperparameters",
                       argv=["bench.py", "--data.input.dataset=synthetic", "--alpha=%r" % w["alpha"], "--loss=bpr_loss",
                             "--recdim=%d" % w["recdim"], "--layer_num=%d" % w["layer_num"],
                             "--batch_size=%d" % w["batch_size"], "--verbose=0"] + list(extra_argv))
    os.chdir(ROOT)
    set_seed(cfg["seed"])
    ds = SyntheticDataset(w["num_users"], w["num_items"], w["num_interactions"], feat_dims=w["feat_dims"], seed=0)
    Logger.logger = Logger(show_in_console=False)
    model = EliMRec(cfg, ds)
    return cfg, ds, model


def cpu_baseline(ds, model_cpu_state, cfg, batches, thread_counts=(8, 16, 32, 64, 128, 1 << 20), timed_steps=5):
    """The oracle (CPU restatement of the reference step, pinned to the reference by tests/test_oracle_golden.py and
    calibrated against the reference's own step time in BASELINE.md) on this box's host cores: one warm-up step, one
    probe step per candidate thread count -- 8, 16, 32, 64, 128 and ALL host cores -- then `timed_steps` steps at the fastest
    count (torch.sparse.mm, 84 % of the reference's step, stops scaling long before 256 threads: the probe shows it)."""
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
    probe = {}
    for k, c in enumerate(counts):
        torch.set_num_threads(c)
        t0 = time.time()
        eo.train_step(om, opt, *batches[(1 + k) % len(batches)])
        probe[c] = round(time.time() - t0, 3)
        if probe[c] > 1.5 * min(probe.values()):      # past the knee (a step at ALL 256 cores measured 63 s against 1.2 s at 32):
            break                                     # larger counts are not probed, the sample text says where it stopped
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    ts = []
    for k in range(timed_steps):
        t0 = time.time()
        eo.train_step(om, opt, *batches[k % len(batches)])
        ts.append(time.time() - t0)
    mean = sum(ts) / len(ts)
    return dict(value=B / mean, unit="triplets/s", cores=best, kind="port", host_cpus=ncpu,
                sample="1 warm-up + 1 probe step per thread count of %s up to the first that is 1.5x slower than the best (s/step %s; "
                       "all %d cores in one step measured 62.7 s, profiles/README.md), then %d timed full training steps (B=%d, same "
                       "workload) at %d threads: %s s" % (counts, probe, ncpu, timed_steps, B, best, [round(t, 3) for t in ts]),
                ms_per_step=1e3 * mean)


def _free_port():
    import socket
    with socket.socket() as s:          # a free port on the loop-back interface
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tail(path, n=25):
    try:
        with open(path, "rb") as f:
            return b"\n".join(f.read().splitlines()[-n:]).decode("utf-8", "replace")
    except OSError:
        return "(no log)"


def spawn_ranks(n):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>, as a child process of a
    parent that has not touched the GPU -- the launch line the driver uses. Every rank it starts is a SUPERVISOR (supervise()
    below) that enforces the wall-clock limit per attempt; this parent only adds an outer guard around the whole launch."""
    import signal
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    limit = float(os.environ.get("ELIMREC_BENCH_LIMIT", 600))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=2 * limit + 120)           # two attempts of `limit` seconds each + start-up
    except subprocess.TimeoutExpired:
        sys.stderr.write("bench.py: the %d-rank launch did not end within %.0f s: killing it\n" % (n, 2 * limit + 120))
        os.killpg(child.pid, signal.SIGKILL)
        return 124


def supervise(args):
    """One torch.distributed.run rank of a multi-GPU bench = one SUPERVISOR: a process that never touches the GPU, starts the
    real rank (`bench.py --worker`, a fresh child each attempt -- nothing is ever exec'ed over a process that initialised HIP),
    and ends it after a wall-clock limit. The supervisors keep a gloo group among themselves (CPU only) and poll once a second
    in lock-step: all children done -> rank 0 forwards its child's JSON line; any child dead, or the limit reached -> every
    child is killed (SIGUSR1 first: the workers dump their Python stacks), each rank prints its child's last log lines, and
    ONE more attempt runs with torch.distributed's collectives instead of the library-owned RCCL communicator
    (ELIMREC_NATIVE_COMM=0) on a fresh rendezvous port. The JSON line says which attempt produced it.
    ELIMREC_BENCH_LIMIT (seconds per attempt, default 600); ELIMREC_TEST_HANG=1 / first (tests): the workers of every / of the
    first attempt hang before their first step."""
    import datetime
    import signal
    import subprocess
    import tempfile
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    limit = float(os.environ.get("ELIMREC_BENCH_LIMIT", 600))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(120.0, limit)))
    logdir = os.environ.get("ELIMREC_BENCH_LOGDIR") or tempfile.mkdtemp(prefix="elimrec_bench_")
    attempts = [("the library's own RCCL communicator", {}), ("torch.distributed collectives (fallback)", {"ELIMREC_NATIVE_COMM": "0"})]
    RUNNING, OK, FAILED = 0, 1, 2
    why = ""
    for k, (name, extra) in enumerate(attempts):
        port = [_free_port() if rank == 0 else None]
        dist.broadcast_object_list(port, src=0)
        env = dict(os.environ)
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)        # the workers' rank 0 hosts its own store on the fresh port
        env.update(MASTER_PORT=str(port[0]), ELIMREC_BENCH_ATTEMPT=str(k), ELIMREC_BENCH_COLLECTIVES=name, **extra)
        env.setdefault("NCCL_DEBUG", "WARN")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if k > 0:
            env["ELIMREC_BENCH_FALLBACK_REASON"] = why[:300]
        out_path = os.path.join(logdir, "rank%d_attempt%d.out" % (rank, k))
        err_path = os.path.join(logdir, "rank%d_attempt%d.err" % (rank, k))
        with open(out_path, "wb") as fo, open(err_path, "wb") as fe:
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"] + sys.argv[1:], env=env, stdout=fo, stderr=fe,
                                     start_new_session=True)
            t0 = time.time()
            verdict = None
            while verdict is None:
                time.sleep(1.0)
                rc = child.poll()
                mine = RUNNING if rc is None else (OK if rc == 0 else FAILED)
                st = torch.zeros(world + 1, dtype=torch.int64)
                st[rank] = mine
                st[world] = int(time.time() - t0 > limit)     # any supervisor's clock past the limit ends the attempt for all
                dist.all_reduce(st)
                states = st[:world].tolist()
                if all(x == OK for x in states):
                    verdict = "ok"
                elif any(x == FAILED for x in states):
                    verdict = "rank(s) %s exited with an error" % [r for r, x in enumerate(states) if x == FAILED]
                elif int(st[world]) > 0:
                    verdict = "no result within %.0f s (ranks still running: %s)" % (limit, [r for r, x in enumerate(states) if x == RUNNING])
            if verdict != "ok" and child.poll() is None:
                try:
                    os.killpg(child.pid, signal.SIGUSR1)     # faulthandler in the worker: Python stacks into its log
                    time.sleep(1.0)
                    os.killpg(child.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                child.wait()
        if verdict == "ok":
            if rank == 0:
                with open(out_path, "rb") as f:
                    sys.stdout.write(f.read().decode("utf-8", "replace"))
                sys.stdout.flush()
            dist.barrier()
            dist.destroy_process_group()
            return 0
        why = "attempt %d (%s): %s" % (k, name, verdict)
        sys.stderr.write("[rank %d] bench.py %s\n[rank %d] last lines of %s:\n%s\n" % (rank, why, rank, err_path, _tail(err_path)))
        sys.stderr.flush()
    dist.destroy_process_group()
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)      # the first ~15 steps after an idle GPU run 1-3 % slower
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-eval", action="store_true", help="skip the secondary evaluator timing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-work", action="store_true", help="skip the reference-equivalent-work line")
    ap.add_argument("--feature-shard", choices=["auto", "row", "replicated"], default="auto",
                    help="folded constants S_m / c: row-sharded with an all_to_all lookup (default for N > 1, the north star's "
                         "partition) or replicated on every rank (default for N = 1, where both are the same tables)")
    ap.add_argument("--feature-dtype", choices=["f32", "f16", "bf16"], default="f32", help="storage of the folded constants")
    ap.add_argument("--no-b-sweep", action="store_true", help="skip the batch-size sweep line")
    ap.add_argument("--no-reduced-precision", action="store_true", help="skip the bf16-feature-storage line (configs[1]'s label)")
    ap.add_argument("--no-projection", action="store_true", help="skip the multi-GPU projection block (emulated ranks on this GPU)")
    ap.add_argument("--projection-only", action="store_true", help=argparse.SUPPRESS)   # the child process of projection_in_child()
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)      # a rank started by supervise()
    args = ap.parse_args()

    if args.projection_only:
        import torch
        device = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        cfg, ds, _ = build(args, device)
        print(json.dumps(multi_gpu_projection(args, device, cfg, ds, WORKLOAD["batch_size"], torch)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves, as CHILD processes of a parent that has not touched
        # the GPU (nothing above imports torch), and leave with their exit code -- the same launch line the driver uses
        raise SystemExit(spawn_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.worker:
        # a rank of a multi-GPU launch (the driver's torch.distributed.run line, or spawn_ranks above): supervise the real rank
        raise SystemExit(supervise(args))
    if args.worker:
        import faulthandler
        import signal
        faulthandler.register(signal.SIGUSR1, all_threads=True)          # the supervisor asks for the stacks before it kills
        sys.argv = [a for a in sys.argv if a != "--worker"]

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    same_gpu = os.environ.get("ELIMREC_SAME_GPU") == "1"   # every rank on device 0 (one-GPU staging of the multi-rank job:
    if same_gpu:                                           # RCCL refuses duplicate devices, so the group is gloo and the
        local_rank = 0                                     # collectives are staged through the host, shard.py)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    multi_path = os.environ.get("ELIMREC_SHARD_MULTI") == "1"     # one rank through the multi-rank step over a one-rank RCCL group
    if world > 1 or multi_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if same_gpu and world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg, ds, model = build(args, device)
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2, slab
    B = WORKLOAD["batch_size"]
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    fshard = args.feature_shard if args.feature_shard != "auto" else ("row" if world > 1 else "replicated")
    eng = ColumnShardEngine(model, feature_shard=fshard, feature_dtype=args.feature_dtype)
    trainer = ColumnShardTrainer(eng, opt, world_size=world, rank=rank)

    # triplets for every step, sampled on the device and resident in HBM before the timed region
    total = args.warmup + args.steps
    sampler = PairwiseSamplerV2(ds, batch_size=B, device=device, seed=cfg["seed"] + rank)
    pools = [sampler.sample_epoch()]
    while sum(p[0].numel() for p in pools) < total * B:
        pools.append(sampler.sample_epoch())
    U_, P_, N_ = (torch.cat([p[i] for p in pools]) for i in range(3))
    batches = [(U_[i * B:(i + 1) * B], P_[i * B:(i + 1) * B], N_[i * B:(i + 1) * B]) for i in range(total)]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # row-sharded constants: the split sizes of every batch's row exchange are planned ahead, as main.py does once per epoch
    # (one device pass over the epoch's triplets + one all_gather; timed here, outside the step loop like the sampler)
    plan_ms = None
    if trainer.lookup and trainer.multi:
        sync()
        t_plan = time.perf_counter()
        trainer.plan_lookup(batches)
        torch.cuda.synchronize()
        plan_ms = 1e3 * (time.perf_counter() - t_plan) / len(batches)

    trainer.prestage(batches)          # every step's triplets are resident (sampled above): planners may run ahead of the step before

    hang = os.environ.get("ELIMREC_TEST_HANG", "")
    if args.worker and (hang == "1" or (hang == "first" and os.environ.get("ELIMREC_BENCH_ATTEMPT", "0") == "0")):
        sys.stderr.write("[rank %d] ELIMREC_TEST_HANG: sleeping in front of the first step\n" % rank)
        sys.stderr.flush()
        while True:
            time.sleep(3600)
    first_losses = []            # the losses of the first steps (ring slots, read after the timed region): the reduced-precision
    for i in range(args.warmup):  # line runs the same batches from the same initial parameters and reports the difference
        first_losses.append(trainer.step(*batches[i]))
    sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        loss = trainer.step(*batches[i])
        if len(first_losses) < 64:
            first_losses.append(loss)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if (same_gpu and world > 1) else device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss.item())

    # the same workload, model, optimizer and engine through the REFERENCE'S loop body (main.py:98-101), right behind the headline
    plugin = None
    if world == 1:
        try:
            plugin = plugin_api_loop(model, opt, batches, torch, steps=args.steps, warmup=args.warmup, headline_ms=1e3 * dt / args.steps)
        except Exception as e:   # noqa: BLE001
            plugin = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    # roofline sample of the dominant kernel, in its own loop (not in the headline's): ONE full propagation hop of this
    # rank's column slice = one elimrec_slab_hop call (sell_hop_kernel + the split rows' sell_fixup_kernel), bracketed by
    # HIP events on the launch stream, alternating between two tables as the forward does
    tabs = [eng.master[eng.cur], eng.tmp[0], eng.tmp[1]]       # the adjoint's scratch tables are free between steps
    n_launch = 100

    def hop_chain(n):
        src, dst = tabs[0], tabs[1]
        for _ in range(n):
            slab.hop(eng.plan, src, dst, gs=eng.gs)
            src, dst = dst, (tabs[2] if dst is tabs[1] else tabs[1])
    hop_chain(4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    hop_chain(n_launch)
    e1.record()
    torch.cuda.synchronize()
    hop_us = e0.elapsed_time(e1) * 1e3 / n_launch

    if rank == 0:
        sb = step_bytes(model, eng, B, world)
        achieved = sb["hop_minimal"] / (hop_us * 1e-6) / 1e9
        ms = 1e3 * dt / args.steps
        out = {
            "metric": "BPR triplets/sec (Tiktok-shape, d=128x3)", "value": B * world * args.steps / dt,
            "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD["name"], "num_users": ds.num_users, "num_items": ds.num_items,
                       "train_interactions": int(ds.train_matrix.nnz), "feat_dims": list(WORKLOAD["feat_dims"]),
                       "recdim": WORKLOAD["recdim"], "layer_num": WORKLOAD["layer_num"], "batch_per_gpu": B,
                       "global_batch": B * world, "parallelism": "colshard%d" % world + ("-multi-rank-path" if multi_path else "") +
                                                                ("+rowshard-features" if fshard == "row" and world > 1 else ""),
                       "graph_table": "column-sharded (%d of %d columns per rank): hops need no communication" % (eng.dl, WORKLOAD["recdim"]),
                       "feature_tables": ("row-sharded S_m / c (1/%d of the users' and items' rows per rank), all_to_all id lookup of the "
                                          "active rows per step" % world) if fshard == "row" and world > 1 else "replicated S_m / c",
                       "feature_dtype": args.feature_dtype,
                       "columns_per_gpu": eng.dl, "slabs": [eng.ns, eng.w, eng.gs],
                       "propagation": "folded", "head_rows": "batch", "final_loss": final_loss,
                       "step_issue": ("one host call per step (csrc/program.hip): %d of the %d timed steps" % (
                           min(trainer._native_state()["native_steps"], args.steps), args.steps)) if trainer._native_state()["native_steps"] else
                                     "launch by launch from Python" + (" (%s)" % trainer._native_state()["failed"] if trainer._native_state()["failed"] else "")},
            # bytes THIS implementation's step has to move per rank (closed form) and the fraction of
            # the HBM peak the whole step reaches on them; the reference algorithm's bytes are quoted beside it
            "step_model": {"bytes_per_rank_step": sb["total"], "GBps": sb["total"] / (ms * 1e-3) / 1e9,
                           "frac_of_hbm_peak": sb["total"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "reference_algorithm_bytes_per_step": reference_step_bytes(model, B)},
            "roofline": {"bound": "hbm",
                         "kernel": ("sell_tier_kernel (one launch over wave tiles)" if eng.plan.tiered else
                                    "sell_hop_kernel + sell_fixup_kernel for the split rows") +
                                   ": one full LightGCN hop X' = A X of this rank's [N x %d] column slice, slab-major %dx%d floats "
                                   "in %d groups" % (eng.dl, eng.ns, eng.w, eng.gs),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic() if world == 1 else None,
                         "traffic_source": ("profiles/%s: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes of a committed profile "
                                            "run of this workload (tools/profile_round.sh), per launch -- NOT collected in this run"
                                            % pmc_file()[0]) if (world == 1 and pmc_file()[0]) else None,
                         # the L2-miss traffic the launch really moves (the PMC figure above) against the same peak: what the
                         # memory side sees, over-fetch included
                         "traffic_GBps": (pmc_traffic() / (hop_us * 1e-6) / 1e9) if (world == 1 and pmc_traffic()) else None,
                         "traffic_frac_of_peak": (pmc_traffic() / (hop_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (world == 1 and pmc_traffic()) else None,
                         "limiter": pmc_limiter(sb["hop_minimal"]),
                         # what the launch's gather instructions move: every non-zero pulls one row piece of each slab through the
                         # CUs' L1 (a source row is gathered deg times; the algorithmic bytes count it once). The chip's measured
                         # rate for uniformly random rows: MI355X_MICROARCH.md, "Indexed rows: gather into LDS"
                         "gather": {"gathered_bytes_per_launch": int(eng.plan.nnz) * eng.dl * 4,
                                    "achieved_GBps": int(eng.plan.nnz) * eng.dl * 4 / (hop_us * 1e-6) / 1e9,
                                    "chip_random_row_gather_GBps": {"table in Infinity Cache (38 MB, 1152-B rows)": 8600,
                                                                    "151 MB table": 7650, "rows shared through L2": 17800},
                                    "frac_of_random_gather_rate": int(eng.plan.nnz) * eng.dl * 4 / (hop_us * 1e-6) / 1e9 / 8600.0},
                         # ... the two ceilings side by side: `frac` prices the launch's ALGORITHMIC bytes against the HBM peak; the
                         # launch's real work is gathering nnz row pieces of a table that lives in the Infinity Cache, whose ceiling is
                         # the chip's random-row gather rate -- the headroom of this kernel is 1 - gather_frac, not 1 - frac
                         "gather_ceiling_TBps": 8.6,
                         "gather_achieved_TBps": int(eng.plan.nnz) * eng.dl * 4 / (hop_us * 1e-6) / 1e12,
                         "gather_frac": int(eng.plan.nnz) * eng.dl * 4 / (hop_us * 1e-6) / 1e9 / 8600.0,
                         "algorithmic_bytes_per_launch": sb["hop_minimal"],
                         "algorithmic_bytes_formula": "read X + write X' + index stream once: 2*N*dl*4 + plan.index_bytes() (8 B per index entry + the tile / item records)",
                         "bytes_with_index_per_group": sb["hop"],
                         "avg_launch_us": hop_us, "launches_timed": n_launch},
        }
        if plugin is not None:
            out["plugin_api_loop"] = plugin
        if world > 1 or multi_path:
            comm = trainer._native_comm()
            nranks = None
            if comm is not None:
                import ctypes
                from elimrec_amd import _lib
                n_c = ctypes.c_int32(0)
                if _lib.load().elimrec_comm_nranks(comm, ctypes.byref(n_c)) == 0:
                    nranks = int(n_c.value)
            out["collectives"] = {"path": ("library-owned RCCL communicator (csrc/program.hip: ncclAllGather / grouped ncclSend+ncclRecv / "
                                           "ncclAllReduce on the step's own streams)") if comm is not None else
                                          ("torch.distributed, backend %s" % dist.get_backend()),
                                  "rccl_nranks": nranks, "attempt": int(os.environ.get("ELIMREC_BENCH_ATTEMPT", "0")),
                                  "fallback_reason": os.environ.get("ELIMREC_BENCH_FALLBACK_REASON")}
        if world > 1:
            out["xgmi_bytes_sent_per_rank_step"] = trainer.xgmi_bytes
            # what ONE GPU does at the same global batch (a step is O(graph) + O(B), so one GPU's triplets/s rises with B):
            # a weak-scaling value is a speed-up only against this figure. Measured here, on rank 0's GPU, after the timed
            # region, while the other ranks wait at the closing barrier.
            out["one_gpu_same_global_batch_triplets_per_s"] = None
            if not args.no_b_sweep:
                try:
                    out["one_gpu_same_global_batch_triplets_per_s"] = one_gpu_at_batch(args, device, cfg, ds, B * world, torch)
                except Exception as e:   # noqa: BLE001
                    out["one_gpu_same_global_batch_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
            if plan_ms is not None:
                out["lookup"] = {"plan_ms_per_batch": plan_ms, "steps_that_synchronised_for_split_sizes": trainer.lookup_syncs,
                                 "row_bytes": eng.lookup_row_bytes, "shard_bytes_per_rank": eng.fshard.nbytes()}
        def extra(key, fn):          # the secondary lines never cost the headline: a failure is reported in place
            try:
                out[key] = fn()
            except Exception as e:   # noqa: BLE001
                out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if world == 1 and not args.no_eval:
            def eval_line():
                try:
                    early = early_state_tie_orders(args, device, cfg, batches, torch)
                except Exception as e:   # noqa: BLE001
                    early = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                return eval_pass(model, cfg, torch, early=early)
            extra("eval", eval_line)
        if world == 1 and not args.no_reference_work:
            extra("reference_equivalent_work", lambda: reference_work_line(args, device, cfg, batches, torch))
        if world == 1 and not args.no_b_sweep:
            extra("batch_sweep", lambda: batch_sweep(trainer, sampler, pools, B, torch))
        if world == 1 and not args.no_projection:
            extra("multi_gpu_projection", projection_in_child)
        if world == 1 and not args.no_reduced_precision and args.feature_dtype == "f32":
            extra("reduced_precision", lambda: reduced_precision_line(args, device, cfg, batches, first_losses, torch))
        mu = pmc_field("mfma_utilisation")
        if mu is not None:
            out["mfma_utilisation"] = dict(mu, source="profiles/%s (committed profile run, not this run)" % pmc_file()[0])
        if not args.no_cpu_baseline and world == 1:     # the host baseline is timed on rank 0 at N=1 only
            cpu_batches = [tuple(x.cpu() for x in b) for b in batches[:5]]
            extra("cpu_baseline", lambda: cpu_baseline(ds, {k: v.cpu().numpy() for k, v in init_state.items()}, cfg, cpu_batches))
        # the secondary lines' headline figures once more INSIDE `config` (records that keep only the contract's keys keep these)
        def pick(key, *path):
            v = out.get(key)
            for q in path:
                v = v.get(q) if isinstance(v, dict) else None
            return v
        out["config"]["also_measured"] = {
            "reference_equivalent_work_ms_per_step": pick("reference_equivalent_work", "ms_per_step"),
            "reference_equivalent_work_triplets_per_s": pick("reference_equivalent_work", "value"),
            "reference_equivalent_work_frac_of_hbm_peak": pick("reference_equivalent_work", "frac_of_hbm_peak"),
            "plugin_api_loop_ms_per_step": pick("plugin_api_loop", "ms_per_step"),
            "plugin_api_loop_with_line_102_item_ms_per_step": pick("plugin_api_loop", "with_line_102_loss_item_every_step", "ms_per_step"),
            "reduced_precision_bf16_ms_per_step": pick("reduced_precision", "ms_per_step"),
            "eval_seconds_id_order": pick("eval", "tie_order", "after_the_timed_steps", "id_seconds"),
            "eval_seconds_reference_order": pick("eval", "tie_order", "after_the_timed_steps", "reference_seconds"),
            "eval_seconds_reference_order_after_2_steps": pick("eval", "tie_order", "after_2_training_steps", "reference_seconds"),
            "eval_seconds_id_order_after_2_steps": pick("eval", "tie_order", "after_2_training_steps", "id_seconds"),
            "batch_sweep_ms_per_step": {str(x["batch"]): x["ms_per_step"] for x in (pick("batch_sweep", "sizes") or []) if isinstance(x, dict)},
            "multi_gpu_PROJECTION_not_measured": {w: {"per_rank_kernel_ms": v.get("per_rank_kernel_ms"), "projected_ms_per_step": v.get("projected_ms_per_step"),
                                                      "projected_triplets_per_s": v.get("projected_triplets_per_s")}
                                                  for w, v in (pick("multi_gpu_projection", "worlds") or {}).items()},
        }
        try:        # C-side stdio first (RCCL prints its version banner there), so that the JSON line is the LAST line on stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def plugin_api_loop(model, opt, batches, torch, steps, warmup, headline_ms):
    """The reference's loop body verbatim -- `loss = model.bpr_loss(u, p, n); opt.zero_grad(); loss.backward(retain_graph=True);
    opt.step()` (main.py:98-101) -- on the headline's model, optimizer and engine (elimrec_amd/plugin.py: the four calls
    complete a request, FusedAdam.step() enqueues the engine's whole step). Then the same with line 102's per-step
    `loss.cpu().item()`, which makes the host wait for every step."""
    ctl = model.plugin
    fast0, slow0 = ctl.fast_steps, ctl.slow_steps
    n = len(batches)

    def body(k):
        u, p, neg = batches[k % n]
        loss = model.bpr_loss(u, p, neg)
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        return loss
    for k in range(warmup):
        body(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        loss = body(warmup + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    B = batches[0][0].numel()
    out = {"what": "main.py:98-101 of the reference verbatim (model.bpr_loss -> opt.zero_grad -> loss.backward(retain_graph=True) -> "
                   "opt.step), same model / optimizer / engine / batches as the headline, %d timed steps after %d" % (steps, warmup),
           "ms_per_step": 1e3 * dt, "triplets_per_s": B / dt, "vs_headline_ms_per_step": 1e3 * dt / headline_ms,
           "final_loss": float(loss.item())}
    k_sync = min(steps, 300)
    for k in range(min(warmup, 20)):         # (the step whose BPR launch publishes the loss is its own one-call program: traced here)
        body(k).cpu().item()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(k_sync):
        body(k).cpu().item()
    torch.cuda.synchronize()
    dts = (time.perf_counter() - t0) / k_sync
    out["with_line_102_loss_item_every_step"] = {"ms_per_step": 1e3 * dts, "triplets_per_s": B / dts, "steps": k_sync,
                                                 "steps_whose_loss_was_published_to_the_host": ctl.published_steps,
                                                 "what": "main.py:102's loss.cpu().item() after every step: the launch that sums the loss "
                                                         "stores it into coherent host memory and the read waits for that launch, not for "
                                                         "the step's end (plugin.py PendingLoss, elimrec_bpr_head_rows_sum_pub)"}
    out["steps_through_the_one_enqueue_path"] = ctl.fast_steps - fast0
    out["steps_launch_by_launch"] = ctl.slow_steps - slow0
    return out


def one_gpu_at_batch(args, device, cfg, ds, Bg, torch, steps=20, warmup=5):
    """The one-rank engine (all columns, replicated constants, the one-GPU fast path) at batch Bg on this rank's GPU."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2
    _, _, model = build(args, device)
    model = model.to(device)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    tr = ColumnShardTrainer(ColumnShardEngine(model), opt, world_size=1, rank=0)
    sampler = PairwiseSamplerV2(ds, batch_size=Bg, device=device, seed=cfg["seed"])
    pools = []
    while sum(p[0].numel() for p in pools) < (steps + warmup) * Bg:
        pools.append(sampler.sample_epoch())
    U_, P_, N_ = (torch.cat([p[i] for p in pools]) for i in range(3))
    bs = [(U_[i * Bg:(i + 1) * Bg], P_[i * Bg:(i + 1) * Bg], N_[i * Bg:(i + 1) * Bg]) for i in range(steps + warmup)]
    tr.prestage(bs)
    for b in bs[:warmup]:
        tr.step(*b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in bs[warmup:]:
        tr.step(*b)
    torch.cuda.synchronize()
    return Bg * steps / (time.perf_counter() - t0)


def batch_sweep(trainer, sampler, pools, B0, torch, sizes=(2048, 4096, 8192, 16384, 32768), steps=20, warmup=3):
    """One GPU, the same engine: the step at growing batch sizes. A step is O(graph) + O(B): the hops, the adjoint and Adam do
    not depend on B, everything at the active rows does -- so triplets/s rises with B on ONE GPU, and a multi-GPU "weak
    scaling" number (B triplets per GPU) has to be read against this curve, not against the B = 2048 point alone."""
    import time as _t
    out = []
    for B in sizes:
        need = (steps + warmup) * B
        while sum(p[0].numel() for p in pools) < need:
            pools.append(sampler.sample_epoch())
        U_, P_, N_ = (torch.cat([p[i] for p in pools]) for i in range(3))
        bs = [(U_[i * B:(i + 1) * B], P_[i * B:(i + 1) * B], N_[i * B:(i + 1) * B]) for i in range(steps + warmup)]
        trainer.prestage(bs)
        for b in bs[:warmup]:
            trainer.step(*b)
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for b in bs[warmup:]:
            trainer.step(*b)
        torch.cuda.synchronize()
        dt = (_t.perf_counter() - t0) / steps
        out.append({"batch": B, "ms_per_step": 1e3 * dt, "triplets_per_s": B / dt})
    return {"what": "same engine, one GPU, %d timed steps per size" % steps, "sizes": out}


def pmc_file():
    """The newest committed PMC summary of the training step (profiles/rNN_pmc_traffic.json, written by tools/pmc_summary.py from
    the rocprofv3 --pmc passes of tools/profile_round.sh), as (name, contents) -- or (None, {})."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not paths:
        return None, {}
    try:
        with open(paths[-1]) as f:
            return os.path.basename(paths[-1]), json.load(f)
    except Exception:
        return None, {}


def pmc_field(key):
    return pmc_file()[1].get(key)


def pmc_traffic():
    """HBM-side bytes per full hop from the committed PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc
    runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950). A number of the COMMITTED profile, not of this run:
    the line says which file it came from (`traffic_source`). None if there is no such file."""
    return pmc_file()[1].get("propagation_hop_traffic_bytes")


def pmc_limiter(algorithmic_bytes):
    """What the committed counters say bounds the hop, in words, with every number read from the file."""
    name, d = pmc_file()
    t = d.get("propagation_hop_traffic_bytes")
    if not t:
        return None
    text = ("the launch moves %.2fx its algorithmic bytes past L2 (every XCD pulls its share of the table through a 4 MB L2 on a random "
            "graph), L2 hit rate %.2f; on the CU side TA busy %.0f %% and L1 stalled on pending misses %.0f %% of the launch "
            "(profiles/%s; DESIGN.md section 3)" % (t / algorithmic_bytes, d.get("propagation_hop_L2_hit_rate", float("nan")),
                                                    100 * d.get("propagation_hop_TA_busy_frac", float("nan")),
                                                    100 * d.get("propagation_hop_TCP_pending_stall_frac", float("nan")), name))
    return text + hop_forms_note()


def hop_forms_note():
    """The other launch form of the hop tried at this shape (the window sweep, csrc/sweep.hip: one side's rows accumulated in LDS
    over L2-sized windows of the other side), from the committed profile of that experiment."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hop_forms.json")))
    if not paths:
        return ""
    try:
        with open(paths[-1]) as f:
            forms = json.load(f)["forms"]
        us = lambda form: float([l for l in forms[form]["timing"] if l.endswith("us per hop")][0].split()[0])
        return ("; the window-sweep form of the hop at this shape (user rows in LDS over windows of item rows, item rows by their own tile "
                "hop): %.1f us against %.1f us for the tile hop and no fewer bytes past L2 -- not taken (profiles/%s)"
                % (us("sweep"), us("tile"), os.path.basename(paths[-1])))
    except Exception:
        return ""


XGMI_LINK_GBS = 153.0          # one direct xGMI link, one direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU)
COLLECTIVE_LATENCY_US = 10.0    # launch + first-byte latency charged per collective on the critical path (assumed, not measured here)


def multi_gpu_projection(args, device, cfg, ds, B, torch, worlds=(2, 4, 8), steps=20):
    """A PROJECTION, not a measurement: what one rank of a W-GPU job computes per step, measured on THIS one GPU (rank 0's engine
    runs every kernel of its step at the real per-rank sizes with its own send buffers fed back as the peers' -- tools/c4_rank_time.py's
    method), plus the step's collectives priced on the direct xGMI links (every pair of GPUs of a node has its own link: the W - 1
    chunks of an all_to_all / all_gather travel at once, each at one link's rate). Nothing here has run on more than one GPU."""
    from elimrec_amd import ColumnShardEngine, FusedAdam, PairwiseSamplerV2

    class _Done:
        def wait(self):
            pass
    sampler = PairwiseSamplerV2(ds, batch_size=B, device=device, seed=cfg["seed"])
    u, p, n = sampler.sample_epoch()
    out = {"what": "PROJECTION from one-GPU measurements -- per-rank kernel time of an emulated rank of a W-rank column-sharded job "
                   "(B = %d triplets per rank) + the step's critical-path collectives on direct xGMI links at %.0f GB/s per link and "
                   "direction, %.0f us charged per collective; the id all_gather, the constants' lookup all_to_all and the weight-gradient "
                   "all_reduce run under the hops (DESIGN.md section 6) and are listed but not added" % (B, XGMI_LINK_GBS, COLLECTIVE_LATENCY_US),
           "worlds": {}}
    R, d = 3 * B, WORKLOAD["recdim"]
    sum_d = sum(WORKLOAD["feat_dims"])
    for W in worlds:
        if d % (4 * W):
            continue
        _, _, model = build(args, device)
        model = model.to(device)
        opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        eng = ColumnShardEngine(model)
        eng.cs_setup(W, 0, opt)
        eng.multi_aux = True
        scale = torch.full((1,), 1.0 / W, device=device)

        def step(i):
            act = eng.cs_plan(u[i * B:(i + 1) * B], p[i * B:(i + 1) * B], n[i * B:(i + 1) * B])
            ps = eng.plan_stream()
            if ps is not None:                            # the planner ran on the second stream: the copy below reads its list
                torch.cuda.current_stream().wait_stream(ps)
            acts = act.view(1, -1).expand(W, -1).contiguous()
            eng.cs_gathered_ids(acts, _Done())
            eng.cs_forward_hops()
            if not eng._long_wanted_only():
                eng.cs_forward_long()
            send = eng.cs_forward_rows(acts)
            eng.cs_head(send)
            s2, _ = eng.cs_backward_local(scale)
            eng.cs_backward_hops(s2, acts)
            eng.cs_update()
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4, 4 + steps):
            step(i)
        torch.cuda.synchronize()
        kernel_ms = 1e3 * (time.perf_counter() - t0) / steps
        dl = d // W
        chunk = lambda total_sent: total_sent / (W - 1)                       # bytes per peer: they travel on W - 1 links at once
        coll = {"all_gather_ids": 4 * R * (W - 1), "all_to_all_rows_fwd": R * 2 * dl * 4 * (W - 1), "all_to_all_sources_bwd": R * 2 * dl * 4 * (W - 1),
                "all_to_all_v_constants_lookup_upper_bound": int(R * (sum_d + 4) * 4 * (W - 1) / W), "all_reduce_weight_grads": 4 * 70000}
        t_us = {k: chunk(v) / (XGMI_LINK_GBS * 1e3) + COLLECTIVE_LATENCY_US for k, v in coll.items()}
        critical_us = t_us["all_to_all_rows_fwd"] + t_us["all_to_all_sources_bwd"]
        step_ms = kernel_ms + critical_us / 1e3
        out["worlds"][str(W)] = {
            "columns_per_rank": dl, "per_rank_kernel_ms": kernel_ms, "bytes_sent_per_rank_step": coll,
            "modelled_us_per_collective": {k: round(v, 1) for k, v in t_us.items()},
            "critical_path_collectives_us": round(critical_us, 1), "projected_ms_per_step": step_ms,
            "projected_triplets_per_s": B * W / (step_ms * 1e-3)}
        del model, opt, eng
        torch.cuda.empty_cache()
    # the alternative north_star names for the graph table too: ROW shards (each rank computes 1/W of a hop's output rows from
    # the whole input table), an all_gather of the table per hop on the direct links -- priced, not built
    N = ds.num_users + ds.num_items
    L = WORKLOAD["layer_num"]
    table = N * d * 4
    alt = {}
    for W in worlds:
        per_link = table / W                                                  # every peer sends me its 1/W of the rows on its own link
        hop_gather_us = per_link / (XGMI_LINK_GBS * 1e3) + COLLECTIVE_LATENCY_US
        alt[str(W)] = {"all_gather_bytes_received_per_hop": int(table * (W - 1) / W), "modelled_us_per_hop_all_gather": round(hop_gather_us, 1),
                       "hops_per_step": 2 * L, "all_gather_us_per_step": round(2 * L * hop_gather_us, 1)}
    # one hop of one rank under either partition, measured on one MI355X with tools/row_range_hop.py on the round-6 build (constants of
    # that run, not of this one): rank 0's rows r, r + W, ... at full width against all rows at d / W columns
    measured = {"2": (21.4, 23.5), "4": (18.4, 24.0), "8": (15.5, 24.4)}
    for W in worlds:
        m = measured.get(str(W))
        if m is not None:
            alt[str(W)]["hop_us_per_rank_row_partition_measured_r06"] = m[0]
            alt[str(W)]["hop_us_per_rank_column_partition_measured_r06"] = m[1]
            alt[str(W)]["row_partition_hop_plus_all_gather_us"] = round(m[0] + alt[str(W)]["modelled_us_per_hop_all_gather"], 1)
    out["row_sharded_graph_table_alternative"] = {
        "what": "not built: per hop every rank needs the whole [N x d] input (%.1f MB): an all_gather on the direct links -- NOT the ring "
                "time earlier rounds quoted (7x more). A rank's hop over 1/W of the rows at full width was measured beside the column "
                "shards' hop (DESIGN.md section 6): the smaller hop does not pay for its exchange at 153 GB/s per link" % (table / 1e6),
        "worlds": alt}
    return out


def projection_in_child(limit_s=240):
    """multi_gpu_projection in a process of its own (`bench.py --projection-only`): it builds three more engines with emulated
    peers on this GPU, and nothing it does -- a fault included -- may cost the headline line. Returns the block, or the reason."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--projection-only"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=limit_s, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "the projection's process did not finish within %d s" % limit_s}
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "the projection's process ended with code %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}
    return json.loads(lines[-1])


def tied_rows(model, evalr, torch):
    """Users of the validation pass whose K + 1 best masked scores contain two equal values (counted on the device)."""
    users = list(evalr.user_pos_test.keys())
    dev = model._require_gpu()
    n = 0
    for a in range(0, len(users), 8192):
        part = users[a:a + 8192]
        ptr, items = evalr._batch_csr(part, evalr.user_pos_train, dev, unique=False)
        _, val = model.predict_device(torch.as_tensor(part, device=dev), top_k=evalr.max_top + 1, train_ptr=ptr, train_items=items)
        n += int((val[:, :-1] == val[:, 1:]).any(1).sum())
    return n


def early_state_tie_orders(args, device, cfg, batches, torch):
    """The evaluator's two tie orders on a model that has taken TWO training steps from its initial parameters: the TIE scores
    of the whole catalogue then crowd around 0.5 and most users' best scores collide."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    _, _, model = build(args, device)
    model = model.to(device)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    tr = ColumnShardTrainer(ColumnShardEngine(model), opt)
    for i in range(2):
        tr.step(*batches[i])
    model.predict_type = "TIE"
    evalr = model.valid_evaluator.evaluator
    out = {}
    for order in ("id", "reference"):
        evalr.tie_order = order
        secs = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.evaluate()
            torch.cuda.synchronize()
            secs.append(time.perf_counter() - t1)
        out[order + "_seconds"] = min(secs[1:])
    out["reference_over_id"] = out["reference_seconds"] / out["id_seconds"]
    out["rows_with_a_tie_in_top_K_plus_1"] = tied_rows(model, evalr, torch)
    return out


def eval_pass(model, cfg, torch, early=None):
    """Secondary metric of SURVEY 8(d): full-catalogue TIE top-K validation pass on the device evaluator (after the timed
    region; the first pass also materialises the cached tables). flops = 2*d per (user, item) for the row means of pass 1
    + 2*d*(1+S) for the (1+S) dot products of pass 2."""
    from elimrec_amd import _lib
    lib = _lib.load()
    model.predict_type = "TIE"

    def timed(n):
        out = []
        for _ in range(n):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.evaluate()
            torch.cuda.synchronize()
            out.append(time.perf_counter() - t1)
        return out
    default_math = int(lib.elimrec_score_get_math())
    evalr = model.valid_evaluator.evaluator
    tie_default = evalr.tie_order
    evalr.tie_order = "id"
    secs = timed(3)
    evalr._evaluations = 0                 # (the evaluator cross-checks its scorer in the first evaluation of a run and every 16th after it:
    checked_secs = timed(1)                #  the pass that carries the check, timed on its own)
    # the reference's tie order (the default of the evaluator and of main.py): evaluate.h:26-33's partial_sort_copy replayed on the
    # device inside the scoring call, every row -- timed beside the device's own rule, with the number of rows whose K + 1 best
    # scores hold an exact tie (the rows on which the two rules can differ), at this state and at an early one
    evalr.tie_order = "reference"
    ref_secs = timed(3)[1:]
    tied_now = tied_rows(model, evalr, torch)
    evalr.tie_order = tie_default
    lib.elimrec_score_set_math(0)          # EXACT: IEEE division + libm expf (the default's factors are within ~2 ulp of these)
    exact = timed(2)
    lib.elimrec_score_set_math(default_math)
    n_eval = len(model.valid_evaluator.evaluator.user_pos_test)
    topks = cfg["topks"]
    flops = float(n_eval) * model.num_items * 2 * model.latent_dim * (2 + model.S)
    best = min(secs[1:])
    return {"what": "full-catalogue TIE top-%d validation pass" % (max(topks) if isinstance(topks, (list, tuple)) else int(topks)),
            "users": n_eval,
            "math": "v_exp/v_rcp + Newton step, the heads' sigmoid product through one reciprocal, dot products as six bf16 piece products "
                    "with fp32 accumulation: scores within 2.4e-7 of the IEEE/libm fp32-MFMA form (default)" if default_math else "exact",
            "users_per_launch": model.valid_evaluator.evaluator.block_users,
            "seconds_first": secs[0], "seconds": best, "users_per_s": n_eval / best,
            # the scorer's dot products run on the bf16 matrix cores: SIX bf16 piece products per fp32 product (exact three-piece
            # splits), so the matrix cores retire 6x the fp32-equivalent flops and the pipe to price them against is the dense
            # bf16 MFMA peak; the fp32-equivalent figure (against the fp32 MFMA peak) stays beside it
            "roofline": {"bound": "mfma", "kernel": "score_t16b_kernel (pass 1 + pass 2 on the bf16 matrix cores, six piece products per fp32 "
                                                    "product) over the whole validation pass",
                         "achieved": 6 * flops / best / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": 6 * flops / best / 1e12 / MFMA_BF16_PEAK_TF, "bf16_flops": 6 * flops,
                         "fp32_equivalent": {"flops": flops, "achieved": flops / best / 1e12, "peak": MFMA_F32_PEAK_TF,
                                             "frac": flops / best / 1e12 / MFMA_F32_PEAK_TF}},
            "scorer_cross_check": {"what": "the first evaluation of a run and every %d-th after it re-score their first %d users with the fp32-MFMA "
                                           "scorer and compare the K returned scores with the default bf16x3 scorer's (> 1e-6 = mismatch, read behind "
                                           "the pass's launches; a mismatch switches the process to the fp32 scorer and the pass is scored again); "
                                           "`seconds` is a pass without the check, `seconds_with_check` one that carries it"
                                           % (evalr.scorer_check_every, evalr.scorer_check_users),
                                   "seconds_with_check": checked_secs[0],
                                   "users_checked": evalr.scorer_checked_rows, "scorer_mismatch_rows": evalr.scorer_mismatch_rows},
            "tie_order": {"what": "seconds per validation pass by the order among equal scores: id = (score desc, item id asc), what `seconds` "
                                  "above is; reference = the default, std::partial_sort_copy's heap order (evaluate.h:26-33) replayed on the "
                                  "device by ref_order_kernel inside the scoring call, every row, nothing copied to the host; "
                                  "rows_with_a_tie_in_top_K_plus_1 = rows on which the two orders can differ",
                          "after_the_timed_steps": {"id_seconds": best, "reference_seconds": min(ref_secs),
                                                    "reference_over_id": min(ref_secs) / best,
                                                    "rows_with_a_tie_in_top_K_plus_1": tied_now},
                          "after_2_training_steps": early if early is not None else "skipped"},
            "exact_math": {"seconds": min(exact), "users_per_s": n_eval / min(exact),
                           "frac_of_mfma_peak": flops / min(exact) / 1e12 / MFMA_F32_PEAK_TF}}


def reduced_precision_line(args, device, cfg, batches, first_losses, torch, steps=200):
    """BASELINE.json configs[1] says "bf16": the V / A / T feature constants STORED in bf16 (--feature_dtype=bf16), widened when
    a step looks its rows up, all arithmetic fp32. The same model from the same initial parameters on
    the same batches: ms per step, and how far its losses are from the fp32 run's over the first steps."""
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam
    _, _, model = build(args, device)
    model = model.to(device)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    eng = ColumnShardEngine(model, feature_dtype="bf16")
    tr = ColumnShardTrainer(eng, opt)
    tr.prestage(batches)
    n_cmp = len(first_losses)
    mine = [tr.step(*batches[i]) for i in range(n_cmp)]
    a = torch.stack([x.detach() for x in mine]).cpu()
    b = torch.stack([x.detach() for x in first_losses]).cpu()
    n = len(batches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(*batches[(n_cmp + i) % n])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    B = batches[0][0].numel()
    return {"what": "--feature_dtype=bf16: S_m / c stored in bf16 (c as hi + lo) and read by the fused head where they lie, widened in "
                    "registers (elimrec_head_fwd_fused_src16), fp32 arithmetic; same initial parameters and batches as the fp32 headline",
            "ms_per_step": 1e3 * dt, "triplets_per_s": B / dt, "steps": steps,
            "loss_delta_vs_f32": {"steps_compared": n_cmp, "max_abs": float((a - b).abs().max()), "mean_abs": float((a - b).abs().mean()),
                                  "f32_loss_first_last": [float(b[0]), float(b[-1])], "bf16_loss_first_last": [float(a[0]), float(a[-1])]},
            "stated_tolerance": "loss 2e-3 abs, gradients 2e-2 (tests/test_hip_parity.py::test_full_tiktok_shape_with_bf16_feature_storage)",
            "feature_table_bytes": {"f32": int((model.num_users + model.num_items) * (eng.fshard.sum_d + 1) * 4),
                                    "bf16": int(eng.fshard.nbytes())}}


def reference_work_line(args, device, cfg, batches, torch, steps=20):
    """The same step with the work the reference's algorithm does: all M tables through the graph (bipartite form) and
    every projection over all N rows every step (--propagation=bipartite --head_rows=all), row-major trainer."""
    from elimrec_amd import FusedAdam
    from elimrec_amd.dist import DataParallelTrainer
    _, _, model = build(args, device, extra_argv=["--propagation=bipartite", "--head_rows=all"])
    model = model.to(device)
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    tr = DataParallelTrainer(model, opt)
    for i in range(3):
        tr.step(*batches[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(*batches[(3 + i) % len(batches)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    B = batches[0][0].numel()
    rb = reference_step_bytes(model, B)
    return {"what": "--propagation=bipartite --head_rows=all: M tables through the graph, projections over all rows",
            "ms_per_step": 1e3 * dt, "value": B / dt, "unit": "triplets/s", "steps": steps,
            "reference_algorithm_bytes_per_step": rb, "GBps": rb / dt / 1e9, "frac_of_hbm_peak": rb / dt / 1e9 / HBM_PEAK_GBS}


if __name__ == "__main__":
    main()
