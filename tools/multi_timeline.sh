#!/bin/bash
# On the GPU box: kernel timeline of one step of the multi-rank path over a one-rank RCCL group, row-sharded constants.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_multi; rm -rf $O; mkdir -p $O
export ELIMREC_SHARD_MULTI=1 FEATURE_SHARD=${FEATURE_SHARD:-row}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o m -- python3 $R/tools/step_trace.py 80 > $O/run.log 2>&1 < /dev/null
tail -2 $O/run.log
python3 $R/tools/timeline.py $(find $O -name "*kernel_trace.csv" | head -1) 3 | tee $O/timeline.txt | tail -45
find $O -name "*kernel_trace.csv" -delete
