#!/usr/bin/env python3
"""Per-stream launch sequence of ONE train step from a rocprofv3 kernel trace (csv):
   python tools/timeline.py <..._kernel_trace.csv> [step_index_from_end]
Prints per stream (queue) the kernels in
start order with duration, the gap to the previous kernel on the same stream, and the wall span of the step."""
import csv, sys, collections
path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id") or r.get("Queue_Id"), int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)))
rows.sort()
def short(n):
    n = n.replace("dgg::", "").replace("void ", "")
    return n.split("(")[0][:46]
# one step period = the window between two consecutive gen_loss_kernel launches (once per step, in the G phase:
# generator forward): phase-shifted by that forward, but a full period of every stream
marks = [i for i, r in enumerate(rows) if "gen_loss_kernel" in r[2]]
lo, hi = marks[-back - 1], marks[-back]
step = rows[lo:hi]
t0 = step[0][0]
print("step wall %.1f us, %d launches, kernel time %.1f us" % ((max(r[1] for r in step) - t0) / 1e3, len(step), sum(r[1] - r[0] for r in step) / 1e3))
byq = collections.defaultdict(list)
for r in step:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(r[1] - r[0] for r in rs) / 1e3
    print("\n== queue %s: %d launches, busy %.1f us" % (q, len(rs), busy))
    prev = None
    for r in rs:
        gap = (r[0] - prev) / 1e3 if prev is not None else 0.0
        print("  t=%8.1f  dur %7.1f  gap %6.1f  grid %7d  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap, r[4], short(r[2])))
        prev = r[1]
