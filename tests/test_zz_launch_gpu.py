"""The two launchers as a user starts them from a bare shell -- `python bench.py --gpus N` and `python main.py --gpus=N` start
their own ranks as child processes -- staged on ONE GPU (ELIMREC_SAME_GPU=1: gloo group, collectives through the host). They
are subprocess runs, kept in a file of their own that sorts last so that a failure here hides no other test; the launcher the
driver's scaling run depends on (bench.py) goes first. Needs a GPU: `-m gpu`."""
import json
import os
import re
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _bare_env():
    env = dict(os.environ, ELIMREC_SAME_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return env


def test_bench_starts_its_own_ranks_staged_on_one_gpu():
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): it starts its two ranks as child processes before touching
    the GPU, and rank 0 prints ONE JSON line last with n_gpus = 2, the hybrid partition and the per-rank xGMI bytes."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                          "--no-b-sweep"], cwd=ROOT, env=_bare_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2 and line["unit"] == "triplets/s" and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "colshard2+rowshard-features" and line["config"]["global_batch"] == 2 * line["config"]["batch_per_gpu"]
    assert line["value"] > 0 and abs(line["value"] - line["config"]["global_batch"] * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    assert set(line["xgmi_bytes_sent_per_rank_step"]) >= {"all_gather", "all_to_all_fwd", "all_to_all_bwd", "all_reduce", "all_to_all_lookup"}
    # a scaling line is read against ONE GPU at the same global batch
    assert line["one_gpu_same_global_batch_triplets_per_s"] is None or line["one_gpu_same_global_batch_triplets_per_s"] > 0


def test_bench_launch_survives_a_hang():
    """The first multi-GPU run of a path that has only ever run on one GPU must fail loudly and cheaply (or recover), never sit
    out the driver's limit. Every rank of a launch is a supervisor that ends its worker after ELIMREC_BENCH_LIMIT seconds:
    ELIMREC_TEST_HANG=first -- the first attempt's workers hang in front of their first step -- the supervisors kill them,
    print the ranks' last log lines and run ONE more attempt with torch.distributed's collectives: exit 0 and a JSON line that
    says so. ELIMREC_TEST_HANG=1 -- every attempt hangs: non-zero exit inside twice the limit (+ start-up), diagnostics on stderr."""
    import time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-b-sweep"]
    env = dict(_bare_env(), ELIMREC_BENCH_LIMIT="45", ELIMREC_TEST_HANG="first")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["collectives"]["attempt"] == 1 and "no result within" in line["collectives"]["fallback_reason"]
    assert "ELIMREC_TEST_HANG" in out.stderr and "attempt 0" in out.stderr
    env["ELIMREC_TEST_HANG"] = "1"
    env["ELIMREC_BENCH_LIMIT"] = "25"
    t0 = time.time()
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and time.time() - t0 < 2 * 25 + 120
    assert "attempt 1" in out.stderr and not any(l.startswith("{") for l in out.stdout.splitlines())


# a "[TIE]\tR\tP\tNDCG" line of a TEST pass (main.py:141-143's format), whatever else shares the pipe's line with it
_TIE = re.compile(r"\[TIE\]\t(\d\S*)\t(\d\S*)\t(\d\S*)")


def test_driver_with_four_ranks_staged_on_one_gpu(tmp_path):
    """`main.py --gpus=4` end to end as the user runs it: column-sharded training with row-sharded constants, item-sharded
    validation and test, the best checkpoint written by rank 0 -- and every rank reports the same results. The four ranks share
    one pipe: every Logger line is one write tagged `[rank k]`, and the results are taken by pattern, not by whole lines.
    (The first run of this kind found a check-then-mkdir race in getFileName.)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--gpus=4", "--data.input.dataset=synthetic", "--alpha=0.5",
                          "--loss=bpr_loss", "--feature_shard=row", "--synthetic_shape=[600,1400,12000]", "--synthetic_dims=[16,8,12]",
                          "--batch_size=512", "--num_epoch=4", "--test_step=2", "--verbose=1", "--path=%s" % str(tmp_path / "ck")],
                         cwd=ROOT, env=_bare_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    per_rank = {}
    for l in out.stdout.splitlines():
        rank = re.search(r"\[rank (\d+)\]", l)
        for m in _TIE.finditer(l):
            per_rank.setdefault(rank.group(1) if rank else "?", []).append(m.groups())
    assert set(per_rank) == {"0", "1", "2", "3"}, (sorted(per_rank), out.stdout[-1500:])
    finals = {r: v[-1] for r, v in per_rank.items()}
    assert len(set(finals.values())) == 1, finals                 # the four ranks' final test lines agree
    assert len({len(v) for v in per_rank.values()}) == 1, {r: len(v) for r, v in per_rank.items()}
    assert any(f.endswith(".pth.tar") for f in os.listdir(tmp_path / "ck"))
