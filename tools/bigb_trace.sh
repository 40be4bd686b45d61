#!/bin/bash
# On the GPU box: kernel timeline of one step at a large batch (B env, default 32768) -> gpurun_out/bigb_timeline_<B>.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export B=${B:-32768}; O=$R/gpurun_out/bigb_trace_$B; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/tools/step_trace.py 16 > $O/run.log 2>&1 < /dev/null
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $f 3 | tee $R/gpurun_out/bigb_timeline_$B.txt
