cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/coexec; mkdir -p $O
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -o p -- python3 $R/tools/eval_prof.py > /dev/null 2>&1 < /dev/null
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/coexec/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if "score_t16" in n: acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(n[-40:], {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
