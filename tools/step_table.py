"""Per-kernel table of the training step from a rocprofv3 --kernel-trace csv of tools/step_trace.py: for the steady-state steps (the
last `n` of the trace), every (kernel, grid, queue) with its launches per step and average duration, the main queue's kernel time,
the gaps between its launches, and the step span they add up to. Kernels that share a name (full hops / the long-rows hop) are split
by grid size. usage: step_table.py <kernel_trace.csv> <out.json> [n_steps]"""
import collections, csv, json, sys


def _grid(r):
    """Work-items of the launch as the counter csv names them (Grid_Size = X * Y * Z of the kernel-trace csv)."""
    if r.get("Grid_Size"):
        return str(r["Grid_Size"])
    try:
        return str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    except (KeyError, ValueError):
        return "?"


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
starts = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("segment_plan", "triplet_rows_kernel", "plan_bits_kernel"))]
a, b = starts[-n - 1], starts[-1]
steps = n
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 / steps
per = collections.OrderedDict()
main_q = collections.Counter(r.get("Queue_Id", "?") for r in rows[a:b] if "sell_tier" in r["Kernel_Name"]).most_common(1)[0][0]
for r in rows[a:b]:
    key = (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("elimrec::", ""), _grid(r), r.get("Queue_Id", "?"))
    e = per.setdefault(key, [0, 0.0])
    e[0] += 1
    e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
main = [r for r in rows[a:b] if r.get("Queue_Id", "?") == main_q]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main) / 1e3 / steps
gaps = collections.OrderedDict()
for x, y in zip(main[:-1], main[1:]):
    g = (int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3
    k = "%s -> %s" % (x["Kernel_Name"].split("(")[0].replace("void ", "").replace("elimrec::", "")[:44], y["Kernel_Name"].split("(")[0].replace("void ", "").replace("elimrec::", "")[:44])
    e = gaps.setdefault(k, [0, 0.0])
    e[0] += 1
    e[1] += g
doc = {"what": __doc__.split("usage")[0].strip(), "steps_averaged": steps, "step_span_us": round(span, 2), "main_queue": main_q,
       "main_queue_kernel_us_per_step": round(busy, 2), "main_queue_gap_us_per_step": round(span - busy, 2),
       "kernels": [{"kernel": k[0], "grid": k[1], "queue": k[2], "on_main_queue": k[2] == main_q, "launches_per_step": round(v[0] / steps, 2),
                    "avg_us": round(v[1] / v[0], 2), "us_per_step": round(v[1] / steps, 2)} for k, v in per.items()],
       "main_queue_gaps": [{"between": k, "per_step": round(v[0] / steps, 2), "avg_us": round(v[1] / v[0], 2)} for k, v in gaps.items() if v[1] / v[0] > 0.5]}
json.dump(doc, open(sys.argv[2], "w"), indent=1)
print("step span %.1f us = main-queue kernels %.1f + gaps %.1f; %d kernel groups" % (span, busy, span - busy, len(per)))
