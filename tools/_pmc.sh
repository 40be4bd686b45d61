cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)\s*:\s*\S+|SQ_[A-Z_0-9]+|TA_[A-Z_0-9]+|TCP_[A-Z_0-9]+" | sort -u | tr '\n' ' ' | head -c 6000 > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_hop_$i -o p -- python3 $GRAFT_REPO_ROOT/tools/hop_only.py 64 6 > /dev/null 2>&1 < /dev/null
done
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out"
for f in sorted(glob.glob(root+"/pmc_hop_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        if "sell_tier" in r["Kernel_Name"] or "sell_hop" in r["Kernel_Name"]:
            a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,(n,v) in acc.items(): print("%-40s per launch %.4g  (%d rows)"%(k, v/max(n,1)*1.0, n))
PY
