#!/bin/bash
# On the GPU box: counters of the C4-shape hop, tile hop + window sweep (FETCH_SIZE, WRITE_SIZE, L2 hits in separate passes).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_c4; mkdir -p $O
export SHAPE=c4 SWEEP=1 ELIMREC_SWEEP_WINDOW=${WINDOW:-32768}
H="python3 $R/tools/hop_only.py 128 4"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o h -- $H > $O/hop.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $H > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $H > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -o p -- $H > /dev/null 2>&1 < /dev/null
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_c4"
for sub, names in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] == names[0]: cnt[k] += 1
    for k in acc:
        if "sweep" in k or "sell_tier" in k:
            print(sub, k, {n: acc[k][n] / max(cnt[k], 1) for n in names}, "launches", cnt[k])
PY
cat $O/stats/*/*kernel_stats.csv 2>/dev/null | head -8
