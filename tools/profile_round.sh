# Round profiles (run on the GPU box through gpurun; outputs under gpurun_out/$ROUND (default r05), summaries are then copied to
# profiles/ by tools/profile_collect.py).
# Kernel stats and PMC counters in SEPARATE rocprofv3 runs (no --pmc together with trace domains).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND:-r05}; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-work --no-b-sweep --no-projection --no-reduced-precision"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- $B > $O/bench_stats.log 2>&1 < /dev/null
P="python3 $R/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-reference-work --no-eval --no-b-sweep --no-projection --no-reduced-precision"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -o p -- $P > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ta -o p -- $P > /dev/null 2>&1 < /dev/null
# evaluator: kernel stats + MFMA / VALU activity of the scorer
E="python3 $R/tools/eval_prof.py"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval_stats -o e -- $E > $O/eval_stats.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/eval_pmc -o p -- $E > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/eval_fetch -o p -- $E > /dev/null 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/eval_write -o p -- $E > /dev/null 2>&1 < /dev/null
# C4 shape (BASELINE configs[3]: |I| = 1.2 M, recdim 128): HBM traffic of a full hop
export SHAPE=c4
H="python3 $R/tools/hop_only.py 128 12"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -o h -- $H > $O/c4_hop.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4_fetch -o p -- $H > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4_write -o p -- $H > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/c4_l2 -o p -- $H > /dev/null 2>&1 < /dev/null
# ... and the same hop with the user rows by the window sweep (csrc/sweep.hip) + the item rows by a tile plan of their own
export SWEEP=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4s_stats -o h -- $H > $O/c4s_hop.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4s_fetch -o p -- $H > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4s_write -o p -- $H > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/c4s_l2 -o p -- $H > /dev/null 2>&1 < /dev/null
unset SWEEP SHAPE
# the multi-rank step over a one-rank RCCL communicator (the library's own RCCL calls, issued from the step's program)
export ELIMREC_SHARD_MULTI=1 FEATURE_SHARD=row
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/multi_stats -o m -- python3 $R/tools/step_trace.py 60 > $O/multi.log 2>&1 < /dev/null
T=$(find $O/multi_stats -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python3 $R/tools/timeline.py $T 3 > $O/multi_timeline.txt 2>&1       # (no trace file: the run died; keep the old timeline)
unset ELIMREC_SHARD_MULTI FEATURE_SHARD
# the one-rank step's timeline (the headline path)
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/step_trace -o s -- python3 $R/tools/step_trace.py 80 > $O/step_trace.log 2>&1 < /dev/null
T=$(find $O/step_trace -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python3 $R/tools/timeline.py $T 3 > $O/step_timeline.txt 2>&1
[ -n "$T" ] && python3 $R/tools/step_table.py $T $O/step_table.json 20 > $O/step_table.log 2>&1
# the bench run's kernels with the names that cover several launch shapes split by grid size (full hops / long-rows hop / the roofline loop)
T=$(find $O/stats -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python3 - "$T" "$O/bench_kernels_by_grid.csv" <<'PY'
import collections, csv, sys
def _grid(r):
    if r.get("Grid_Size"):
        return str(r["Grid_Size"])
    try:
        return str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    except (KeyError, ValueError):
        return "?"
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0], _grid(r), r.get("Queue_Id", "?"))
    e = acc.setdefault(k, [0, 0])
    e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Grid_Size", "Queue_Id", "Calls", "TotalDurationNs", "AverageNs"])
    for (n, g, q), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, g, q, c, t, round(t / c)])
PY
# the step at B = 32768 (65 k active rows): timeline of one step, counters of its O(B) kernels
B=32768 bash $R/tools/bigb_trace.sh > /dev/null 2>&1
cp $R/gpurun_out/bigb_timeline_32768.txt $O/step_timeline_B32768.txt
B=32768 bash $R/tools/bigb_pmc.sh > $O/bigb_counters.txt 2>&1
rm -rf $R/gpurun_out/bigb_trace_32768 $R/gpurun_out/r05_bigb
find $O -name "*kernel_trace.csv" -delete      # large; the stats csv is what gets committed
ls -R $O | head -50
