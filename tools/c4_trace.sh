#!/bin/bash
# On the GPU box: kernel statistics of the configs[3] step on one GPU (tools/c4_rank_time.py 1) -> gpurun_out/c4_step_kernel_stats.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4_trace; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c -- python3 $R/tools/c4_rank_time.py 1 > $O/run.log 2>&1 < /dev/null
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/c4_step_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-86s calls %4s avg %8.1f us  %5.1f %%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep "ms of GPU work" $O/run.log
t=$(find $O -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $t 3 | cut -c1-150 | tee $R/gpurun_out/c4_step_timeline.txt
find $O -name "*kernel_trace.csv" -delete
