cd /tmp; export TMPDIR=/tmp
for d in 0 31 15 6 9; do
  ELIMREC_HEAD_DBG=$d timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hb_$d -o hb -- python3 $GRAFT_REPO_ROOT/tools/bench_head.py $d > /dev/null 2>&1 < /dev/null
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/hb_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && echo "dbg $d: $(grep -E 'head_fwd_fused|pack_head|bpr_head_rows' $f < /dev/null | cut -d, -f1,2,4 | tr '\n' ' ')"
done
