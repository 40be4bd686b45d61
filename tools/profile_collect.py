"""gpurun_out/$ROUND/* (tools/profile_round.sh; ROUND defaults to r05) -> profiles/$ROUND_*: kernel-stat CSVs copied, PMC passes
summarised. Usage: python tools/profile_collect.py"""
import collections
import csv
import glob
import json
import shutil
import subprocess
import sys

import os
RND = os.environ.get("ROUND", "r05")
O = "gpurun_out/" + RND


def read(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
        name = r["Kernel_Name"].split("(")[0]
        if "sell_tier_kernel" in name or "half_hop" in name:
            name += " grid=%s" % r["Grid_Size"]
        out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


subprocess.check_call([sys.executable, "tools/pmc_summary.py", "--fetch", O + "/fetch", "--write", O + "/write", "--mfma", O + "/mfma",
                       "--steps", "10", "--out", "/tmp/%s_pmc.json" % RND], stdout=subprocess.DEVNULL)
doc = json.load(open("/tmp/%s_pmc.json" % RND))
l2, ta = read(O + "/l2"), read(O + "/ta")
hop = doc["propagation_hop_kernel"]
h, t = l2[hop], ta[hop]
avg = lambda v: sum(v) / len(v)
gui = avg(t["GRBM_GUI_ACTIVE"])
doc["propagation_hop_L2_hit_rate"] = round(sum(h["TCC_HIT_sum"]) / (sum(h["TCC_HIT_sum"]) + sum(h["TCC_MISS_sum"])), 4)
doc["propagation_hop_counters_per_launch"] = {"TA_BUSY_avr_cycles": round(avg(t["TA_BUSY_avr"])),
                                              "TCP_PENDING_STALL_CYCLES_sum": round(avg(t["TCP_PENDING_STALL_CYCLES_sum"])),
                                              "GRBM_GUI_ACTIVE": round(gui)}
doc["propagation_hop_TA_busy_frac"] = round(avg(t["TA_BUSY_avr"]) / (gui / 8.0), 3)
doc["propagation_hop_TCP_pending_stall_frac"] = round(avg(t["TCP_PENDING_STALL_CYCLES_sum"]) / 256.0 / (gui / 8.0), 3)
doc["note"] += (" TCC_HIT/TCC_MISS, TA_BUSY_avr, TCP_PENDING_STALL_CYCLES_sum from two more passes of the same command. Busy fractions: "
                "counter / (GRBM_GUI_ACTIVE / 8 XCDs); TCP stall cycles are summed over the 256 CUs.")
# MFMA utilisation of the kernels that hold the reference's dense contractions (K1 feature projections, K5 fusion
# Linears, K8 single-modal heads -- fused in head_fwd16 / head_bwd_input16; their weight gradients ride in the adjoint hops' tails)
mf = read(O + "/mfma")
util = {}
for name, c in mf.items():
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) > 0:
        util[name.replace("void ", "").replace("elimrec::", "")] = round(sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(c["GRBM_GUI_ACTIVE"]) / 8 * 1024), 4)
doc["mfma_utilisation"] = {"what": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) per kernel, over its launches; K1 + K5 + K8 forward = "
                                   "head_fwd16_kernel, their input gradients = head_bwd_input16_kernel, their weight gradients = the tail workgroups of "
                                   "sell_tier_bwdw_kernel<8, true> (partial products) -- all latency-bound at <= 3B active rows, not MFMA-bound",
                           "kernels": util}
json.dump(doc, open("profiles/%s_pmc_traffic.json" % RND, "w"), indent=1)
# C4 shape: one full hop of the [N x 128] table
import os
c4 = {}
for tag, label in (("c4", "tile hop over all rows, slab groups side by side (blockIdx % gs)"),
                   ("c4s", "window sweep over the user rows (sweep_rows_kernel) + tile hop over the item rows: both launches of a hop")):
    if not os.path.isdir(O + "/%s_fetch" % tag):
        continue
    f, w = read(O + "/%s_fetch" % tag), read(O + "/%s_write" % tag)
    rows = list(csv.DictReader(open(O + "/%s_stats/h_kernel_stats.csv" % tag)))
    per = lambda c, key: sum(c[key]) * 1024 / len(c[key])
    if tag == "c4s":
        # the hop's two launches; the item-rows launch is the sell_tier_kernel grid that is not the reference run's whole-plan hop
        sweep = [k for k in f if "sweep_rows_kernel" in k][0]
        grids = sorted((k for k in f if "sell_tier_kernel" in k), key=lambda k: len(f[k]["FETCH_SIZE"]))
        items = grids[-1]                                   # launched with every hop (the whole-plan hop runs once, as the reference)
        parts = {"window sweep over the user rows": sweep, "tile hop over the item rows": items}
        fetch = sum(2 * per(f[k], "FETCH_SIZE") for k in parts.values())
        write = sum(per(w[k], "WRITE_SIZE") for k in parts.values())
        tr = list(csv.DictReader(open(glob.glob(O + "/c4s_stats/*kernel_stats.csv")[0])))
        avg_ns = float([r for r in tr if "sweep_rows_kernel" in r["Name"]][0]["AverageNs"])
        log = open(O + "/c4s_hop.log").read()
        c4[label] = {"us_per_hop_both_launches": float([l for l in log.splitlines() if "us per hop" in l][-1].split()[0]),
                     "sweep_launch_us": round(avg_ns / 1e3, 1), "fetch_MB_x2": round(fetch / 1e6, 1), "write_MB": round(write / 1e6, 1),
                     "traffic_MB": round((fetch + write) / 1e6, 1),
                     "per_launch": {n: {"fetch_MB_x2": round(2 * per(f[k], "FETCH_SIZE") / 1e6, 1), "write_MB": round(per(w[k], "WRITE_SIZE") / 1e6, 1)}
                                    for n, k in parts.items()}}
        if os.path.isdir(O + "/c4s_l2"):
            l = read(O + "/c4s_l2")
            c4[label]["L2_hit_rate"] = {n: round(sum(l[k]["TCC_HIT_sum"]) / (sum(l[k]["TCC_HIT_sum"]) + sum(l[k]["TCC_MISS_sum"])), 3) for n, k in parts.items()}
        continue
    hopk = sorted((k for k in f if "sell_tier_kernel" in k), key=lambda k: -len(f[k]["FETCH_SIZE"]))[0]
    avg_ns = float([r for r in rows if "sell_tier_kernel" in r["Name"]][0]["AverageNs"])
    fetch = 2 * sum(f[hopk]["FETCH_SIZE"]) * 1024 / len(f[hopk]["FETCH_SIZE"])
    write = sum(w[hopk]["WRITE_SIZE"]) * 1024 / len(w[hopk]["WRITE_SIZE"])
    c4[label] = {"avg_launch_us": round(avg_ns / 1e3, 1), "fetch_MB_x2": round(fetch / 1e6, 1), "write_MB": round(write / 1e6, 1),
                 "traffic_MB": round((fetch + write) / 1e6, 1)}
    if os.path.isdir(O + "/%s_l2" % tag):
        l = read(O + "/%s_l2" % tag)[hopk]
        c4[label]["L2_hit_rate"] = round(sum(l["TCC_HIT_sum"]) / (sum(l["TCC_HIT_sum"]) + sum(l["TCC_MISS_sum"])), 3)
if c4:
    N, dcol = 36656 + 1217360, 128
    json.dump({"note": "tools/hop_only.py 128 12 with SHAPE=c4 (BASELINE configs[3]: |U| = 36 656, |I| = 1 217 360, recdim 128) under rocprofv3: kernel "
                       "stats, FETCH_SIZE (x2, gfx950) and WRITE_SIZE in separate passes. Algorithmic bytes of a hop = read X + write X' = 2 x N x 128 x 4 "
                       "+ the index stream.", "table_MB": round(N * dcol * 4 / 1e6, 1), "algorithmic_MB_without_index": round(2 * N * dcol * 4 / 1e6, 1),
               "variants": c4}, open("profiles/%s_c4_hop_traffic.json" % RND, "w"), indent=1)
    print(json.dumps(c4, indent=1))
if os.path.exists(O + "/multi_stats/m_kernel_stats.csv"):
    shutil.copy(O + "/multi_stats/m_kernel_stats.csv", "profiles/%s_multi_rank_path_kernel_stats.csv" % RND)
for src, dst in (("multi_timeline.txt", "multi_rank_path_timeline.txt"), ("step_timeline.txt", "step_timeline.txt"),
                 ("step_timeline_B32768.txt", "step_timeline_B32768.txt"), ("bigb_counters.txt", "bigb_counters.txt")):
    if os.path.exists(O + "/" + src) and any(k in open(O + "/" + src).read() for k in ("step span", "mfma_busy")):
        shutil.copy(O + "/" + src, "profiles/%s_%s" % (RND, dst))
shutil.copy(O + "/stats/b_kernel_stats.csv", "profiles/%s_bench_kernel_stats.csv" % RND)
if os.path.exists(O + "/bench_kernels_by_grid.csv"):       # the same run with kernels that cover several launch shapes split by grid size
    shutil.copy(O + "/bench_kernels_by_grid.csv", "profiles/%s_bench_kernels_by_grid.csv" % RND)
if os.path.exists(O + "/step_table.json"):
    # the step's table (tools/step_table.py over the step_trace run) + per-launch counter bytes of the PMC passes (kernel + grid match)
    # + the algorithmic bytes of the hop launches
    tab = json.load(open(O + "/step_table.json"))
    fetch, write = read(O + "/fetch"), read(O + "/write")
    strip = lambda k: k.replace("void ", "").replace("elimrec::", "")
    byname = {}
    for k in fetch:
        base, _, grid = k.partition(" grid=")
        byname.setdefault(strip(base), []).append((grid, k))
    for row in tab["kernels"]:
        cands = byname.get(row["kernel"], [])
        hit = [k for g, k in cands if g == row["grid"]] or ([k for g, k in cands] if len(cands) == 1 else [])
        if hit:
            f, w = fetch[hit[0]].get("FETCH_SIZE", []), write.get(hit[0], {}).get("WRITE_SIZE", [])
            if f:
                row["counter_MB_per_launch"] = round((2 * sum(f) / len(f) + (sum(w) / len(w) if w else 0)) * 1024 / 1e6, 2)
    tab["note"] = ("counter_MB_per_launch = FETCH_SIZE x 2 + WRITE_SIZE of the same kernel and grid in the PMC passes of bench.py (separate "
                   "rocprofv3 --pmc runs); the trace itself stretches a step by ~3 % against the untraced bench figure")
    json.dump(tab, open("profiles/%s_step_table.json" % RND, "w"), indent=1)
shutil.copy(O + "/eval_stats/e_kernel_stats.csv", "profiles/%s_eval_kernel_stats.csv" % RND)
ev, ef, ew = read(O + "/eval_pmc"), read(O + "/eval_fetch"), read(O + "/eval_write")
out = {}
for name in ev:
    if "score_t16" in name or "topk" in name or "rank_metrics" in name:
        c = ev[name]
        n = len(c["SQ_BUSY_CYCLES"])
        row = {"launches": n}
        for k, v in c.items():
            row[k + "_per_launch"] = round(sum(v) / n)
        if name in ef:
            row["fetch_MB_per_launch_x2"] = round(2 * sum(ef[name]["FETCH_SIZE"]) * 1024 / 1e6 / len(ef[name]["FETCH_SIZE"]), 2)
        if name in ew:
            row["write_MB_per_launch"] = round(sum(ew[name]["WRITE_SIZE"]) * 1024 / 1e6 / len(ew[name]["WRITE_SIZE"]), 2)
        dur = row["SQ_BUSY_CYCLES_per_launch"] / 32.0
        row["mfma_busy_frac"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / 1024.0 / dur, 3)
        row["valu_issue_busy_frac"] = round(4 * row["SQ_ACTIVE_INST_VALU_per_launch"] / 1024.0 / dur, 3)
        out[name] = row
json.dump({"note": "rocprofv3 --pmc passes over tools/eval_prof.py (3 TIE validation passes, 8192 users per launch, default math, catalogue in "
                   "a 2048-item pilot chunk + 16384-item chunks, scores stored only in tiles that reach the running K-th best). mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 SEs); "
                   "valu_issue_busy_frac = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / 1024 / the same duration. FETCH_SIZE doubled (gfx950).",
           "kernels": out}, open("profiles/%s_eval_pmc.json" % RND, "w"), indent=1)
for k in ("propagation_hop_kernel", "propagation_hop_traffic_bytes", "propagation_hop_L2_hit_rate", "propagation_hop_TA_busy_frac",
          "propagation_hop_TCP_pending_stall_frac"):
    print(k, doc.get(k))
for k, v in out.items():
    print(k[:60], {a: b for a, b in v.items() if "frac" in a})
