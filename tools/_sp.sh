cd /tmp; export TMPDIR=/tmp
for W in 8; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sp_$W -o s -- python3 $GRAFT_REPO_ROOT/tools/shard_prof.py $W 20 > /dev/null 2>&1 < /dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/sp_$W -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" $W <<'PY'
import csv, sys
W=int(sys.argv[2]); rows=[r for r in csv.DictReader(open(sys.argv[1])) if "elimrec" in r["Name"] and int(r["Calls"])>=20*W]
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("W=%d: elimrec kernels, %.1f us per rank-step (profiled sum over %d kernel kinds)" % (W, tot/1e3/(20*W), len(rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("   %-66s %5.1f us x %.1f" % (r["Name"][:66], float(r["AverageNs"])/1e3, int(r["Calls"])/(20.0*W)))
PY
done
