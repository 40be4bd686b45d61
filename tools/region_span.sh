#!/bin/bash
# On the GPU box: the 20-step timed region of bench.py under --kernel-trace: GPU span of the region against the wall clock the line reports.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/region; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o r -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-work --no-b-sweep --no-projection --no-reduced-precision --no-eval > $O/run.log 2>/dev/null < /dev/null
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$O/run.log" <<'PY'
import csv, json, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
adam = [i for i, r in enumerate(rows) if "sell_tier_adam_kernel" in r["Kernel_Name"]]
# runs of consecutive steps: an idle gap of more than 1 ms before a step's first kernel starts a new region
ends = [int(rows[i]["End_Timestamp"]) for i in adam]
iv = [(ends[k] - ends[k - 1]) / 1e3 for k in range(1, len(ends))]
print("ms_per_step reported (traced run): %.4f" % line["ms_per_step"])
print("step-to-step intervals (Adam hop end to Adam hop end, us):")
print(" ".join("%.0f" % x for x in iv))
PY
rm -f $f
