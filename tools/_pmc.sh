cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ev_$i -o p -- python3 $GRAFT_REPO_ROOT/tools/eval_prof.py > /dev/null 2>&1 < /dev/null
done
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out"
for f in sorted(glob.glob(root+"/pmc_ev_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        for tag in ("score_t16_kernel<2", "score_t16_kernel<1"):
            if tag in r["Kernel_Name"]:
                a=acc[(tag, r["Counter_Name"])]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for (tag,k),(n,v) in sorted(acc.items()): print("%-26s %-34s per launch %.4g  (%d)"%(tag, k, v/max(n,1), n))
PY
