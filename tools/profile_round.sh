# Round profiles (run on the GPU box through gpurun; outputs under gpurun_out/r02, summaries are then copied to profiles/).
# Kernel stats and PMC counters in SEPARATE rocprofv3 runs (no --pmc together with trace domains).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
B="python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-work --no-bf16"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- $B > $O/bench_stats.log 2>&1 < /dev/null
P="python3 $R/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-reference-work --no-bf16 --no-eval"
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
find $O -name "*kernel_trace.csv" -delete      # large; the stats csv is what gets committed
ls -R $O | head -50
