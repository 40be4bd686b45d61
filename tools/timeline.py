"""Timeline of one training step from a rocprofv3 --kernel-trace csv: start offset, duration, queue, kernel. Dev tool.
usage: timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = the launches after one Adam hop (the optimizer's launch ends a step on the main queue) up to and including the next;
# traces without it (other optimizers): from one batch planner to the next
ends = [i for i, r in enumerate(rows) if "sell_tier_adam_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(ends) > k + 1:
    a, b = ends[-k - 1] + 1, ends[-k] + 1
else:
    starts = [i for i, r in enumerate(rows) if any(n in r["Kernel_Name"] for n in ("segment_plan", "triplet_rows_kernel", "plan_bits_kernel"))]
    a, b = starts[-k - 1], starts[-k]
t0 = int(rows[a]["Start_Timestamp"])
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    busy += e - s
    print("%8.1f us  +%6.1f us  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
print("step span %.1f us (first launch to the next step's first launch), kernel time sum %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
