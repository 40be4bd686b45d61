#!/bin/bash
# On the GPU box: kernel statistics of ONE rank's work of a W-rank configs[3] job (tools/c4_rank_time.py W; W env, default 8)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; W=${W:-8}; O=$R/gpurun_out/c4_rank_trace; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c -- python3 $R/tools/c4_rank_time.py $W > $O/run.log 2>&1 < /dev/null
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/c4_rank_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("%-86s calls %4s avg %8.1f us  %5.1f %%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep "ms of GPU work" $O/run.log
find $O -name "*kernel_trace.csv" -delete
