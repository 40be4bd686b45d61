#!/bin/bash
# On the GPU box: kernel timeline of one training step (one rank, the headline path; FEATURE_SHARD / ELIMREC_SHARD_MULTI select others).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r05_step}; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O -o s -- python3 $R/tools/step_trace.py 80 > $O/run.log 2>&1 < /dev/null
python3 $R/tools/timeline.py $(find $O -name "*kernel_trace.csv" | head -1) 3 | tee $O/timeline.txt | tail -24
find $O -name "*kernel_trace.csv" -delete
