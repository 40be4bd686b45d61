cd /tmp; export TMPDIR=/tmp
ELIMREC_AUX_STREAM=${AUX:-1} timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl -o t -- python3 $GRAFT_REPO_ROOT/tools/step_trace.py 30 > /dev/null 2>&1 < /dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/timeline.py $f 3
