#!/bin/bash
# On the GPU box: the two forms of one propagation hop at the Tiktok shape (configs[1]) side by side -- the wave-tile hop
# (sell_tier_kernel, the shipped form) and the window sweep of the user rows + tile hop of the item rows (csrc/sweep.hip, the
# form tables beyond the Infinity Cache take) -- timing and counters (FETCH_SIZE, WRITE_SIZE, L2 hits, TA / L1 stalls, each
# in its own rocprofv3 --pmc pass). Writes gpurun_out/r05_hop_forms.json (copied to profiles/ by hand).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_hopforms; mkdir -p $O
export ELIMREC_SWEEP_WINDOW=${WINDOW:-16384}
for FORM in tile sweep; do
  if [ $FORM = sweep ]; then export SWEEP=1; else unset SWEEP; fi
  H="python3 $R/tools/hop_only.py 64 4"
  timeout 300 python3 $R/tools/hop_only.py 64 12 > $O/$FORM.time.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$FORM/stats -o h -- $H > /dev/null 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$FORM/fetch -o p -- $H > /dev/null 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$FORM/write -o p -- $H > /dev/null 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$FORM/l2 -o p -- $H > /dev/null 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/$FORM/ta -o p -- $H > /dev/null 2>&1 < /dev/null
done
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob, os, collections, json, re
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_hopforms"
doc = {"what": "one LightGCN hop of the [N x 64] table at the Tiktok shape (configs[1]), two launch forms, one MI355X; counters per launch, "
               "each from its own rocprofv3 --pmc pass; FETCH_SIZE doubled (gfx950), KiB -> bytes", "forms": {}}
for form in ("tile", "sweep"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("fetch", "write", "l2", "ta"):
        for f in glob.glob(O + "/%s/%s/**/*counter_collection.csv" % (form, sub), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if "sell_tier_kernel" in k or "sweep" in k:
                    per[k + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    log = open(O + "/%s.time.log" % form).read()
    entry = {"timing": [l.strip() for l in log.splitlines() if " us" in l], "kernels": {}}
    for k, c in per.items():
        mean = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else None
        e = {"launches_counted": len(c.get("FETCH_SIZE", [])), "fetch_MB_x2": None if mean("FETCH_SIZE") is None else round(2 * mean("FETCH_SIZE") * 1024 / 1e6, 2),
             "write_MB": None if mean("WRITE_SIZE") is None else round(mean("WRITE_SIZE") * 1024 / 1e6, 2)}
        if mean("TCC_HIT_sum") is not None:
            e["L2_hit_rate"] = round(mean("TCC_HIT_sum") / (mean("TCC_HIT_sum") + mean("TCC_MISS_sum")), 3)
        if mean("GRBM_GUI_ACTIVE"):
            e["TA_busy_frac"] = round(mean("TA_BUSY_avr") / (mean("GRBM_GUI_ACTIVE") / 8), 3) if mean("TA_BUSY_avr") else None
            e["TCP_pending_stall_frac"] = round(mean("TCP_PENDING_STALL_CYCLES_sum") / 256 / (mean("GRBM_GUI_ACTIVE") / 8), 3) if mean("TCP_PENDING_STALL_CYCLES_sum") else None
        entry["kernels"][k] = e
    doc["forms"][form] = entry
json.dump(doc, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_hop_forms.json", "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
