# per-rank kernel time of a W-rank column-sharded step emulated on one GPU (rocprofv3 sums): usage shard_prof_run.sh W
cd /tmp; export TMPDIR=/tmp
W=${1:-8}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sp_$W -o s -- python3 $GRAFT_REPO_ROOT/tools/shard_prof.py $W 20 > /dev/null 2>&1 < /dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/sp_$W -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" $W <<'PY'
import csv, sys
W=int(sys.argv[2]); rows=[r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"])>=20*W]
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("W=%d: %.1f us of kernels per rank-step (all kernels launched >= once per rank-step, torch copies included)" % (W, tot/1e3/(20*W)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
    print("   %-70s %5.1f us x %.1f" % (r["Name"][:70], float(r["AverageNs"])/1e3, int(r["Calls"])/(20.0*W)))
PY
