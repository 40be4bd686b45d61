cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/evp -o e -- python3 $GRAFT_REPO_ROOT/tools/eval_prof.py 2>&1 < /dev/null | grep pass
f=$(find $GRAFT_REPO_ROOT/gpurun_out/evp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-72s calls %5s total %8.2f ms avg %8.1f us" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
