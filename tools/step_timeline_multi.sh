cd /tmp; export TMPDIR=/tmp
export ELIMREC_SHARD_MULTI=1
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tlm -o t -- python3 $GRAFT_REPO_ROOT/tools/step_trace.py 30 > /dev/null 2>&1 < /dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/tlm -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/timeline.py $f 3
