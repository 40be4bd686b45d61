"""Summarise rocprofv3 --pmc passes of `bench.py --steps K --warmup W --no-cpu-baseline` into the JSON bench.py reads
(profiles/<round>_pmc_traffic.json). Usage:
    python tools/pmc_summary.py --fetch DIR --write DIR --mfma DIR --steps 10 --out profiles/r01_g_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (its calibration:
wide coalesced streaming reads; the hop's 128-B row-piece gathers are the same 16 B/lane loads). MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import argparse, collections, csv, glob, json, os


def read(dirname):
    path = glob.glob(os.path.join(dirname, "*counter_collection.csv"))[0]
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            if "half_hop_kernel" in name or "sell_tier_kernel" in name or "sell_hop" in name:   # full / long-rows-only hops share kernels: split by grid
                name += " grid=%s" % r["Grid_Size"]
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True); ap.add_argument("--write", required=True); ap.add_argument("--mfma", required=True)
    ap.add_argument("--steps", type=int, required=True, help="steps + warm-up steps of the profiled command")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    fetch, write, mfma = read(a.fetch), read(a.write), read(a.mfma)
    kernels = []
    for name in sorted(fetch, key=lambda k: -sum(fetch[k]["FETCH_SIZE"])):
        f = fetch[name]["FETCH_SIZE"]
        w = write.get(name, {}).get("WRITE_SIZE", [])
        if sum(f) * 1024 < 1e6 and sum(w) * 1024 < 1e6:
            continue
        row = {"kernel": name, "launches": len(f), "launches_per_step": round(len(f) / a.steps, 2),
               "fetch_MB_per_launch_x2": round(2 * sum(f) * 1024 / 1e6 / len(f), 2),
               "write_MB_per_launch": round(sum(w) * 1024 / 1e6 / max(len(w), 1), 2)}
        m = mfma.get(name, {})
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and sum(m["GRBM_GUI_ACTIVE"]) > 0:
            row["mfma_util"] = round(sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(m["GRBM_GUI_ACTIVE"]) / 8 * 1024), 4)
        kernels.append(row)
    # the full hop of the training step = the unmasked, Adam-less sell_tier_kernel group with the largest grid
    hop = sorted([k for k in kernels if k["kernel"].startswith("void elimrec::sell_tier_kernel<") and
                  k["kernel"].split(">")[0].replace(" ", "").endswith("false,false,false")],
                 key=lambda k: -int(k["kernel"].split("grid=")[1]))
    doc = {"note": __doc__.split("Usage")[0].strip(), "steps_profiled": a.steps, "kernels": kernels}
    if hop:
        doc["propagation_hop_kernel"] = hop[0]["kernel"]
        doc["propagation_hop_traffic_bytes"] = (hop[0]["fetch_MB_per_launch_x2"] + hop[0]["write_MB_per_launch"]) * 1e6
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1)[:3000])


if __name__ == "__main__":
    main()
