#!/bin/bash
# On the GPU box: one validation pass of tools/eval_prof.py under --kernel-trace: span, kernel time, the largest gaps between launches.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eval_gaps; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o e -- python3 $R/tools/eval_prof.py > $O/run.log 2>&1 < /dev/null
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last pass: from the last split3_items launch that follows a rank_metrics launch
# a pass starts with the scorer's cross-check of its first users (score_t16_kernel<1, ...>: once per pass)
marks = [i for i, r in enumerate(rows) if "score_t16_kernel<1" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy, last_end, gaps = 0, t0, []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > last_end:
        gaps.append(((s - last_end) / 1e3, r["Kernel_Name"].split("(")[0][-50:]))
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
print("pass: %d launches, span %.2f ms, GPU busy %.2f ms, idle %.2f ms" % (len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
for g, name in sorted(gaps, reverse=True)[:12]:
    print("  gap %8.1f us before %s" % (g, name))
import collections
acc = collections.OrderedDict()
for r in seg:
    k = r["Kernel_Name"].split("(")[0][-60:]
    e = acc.setdefault(k, [0, 0]); e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-60s x%3d %8.1f us" % (k, c, t / 1e3))
PY
rm -f $f
