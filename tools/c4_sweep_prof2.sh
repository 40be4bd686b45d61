#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_c4b; mkdir -p $O
export SHAPE=c4 SWEEP=1 ELIMREC_SWEEP_WINDOW=${WINDOW:-16384}
H="python3 $R/tools/hop_only.py 128 4"
timeout 600 rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ta -o p -- $H > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $O/sq -o p -- $H > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $O/tcp -o p -- $H > /dev/null 2>&1 < /dev/null
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_c4b"
for sub in ("ta", "sq", "tcp"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        if "sweep" in k or "sell_tier" in k:
            print(sub, k, {n: round(acc[k][n] / max(cnt[k][n], 1)) for n in acc[k]})
PY
