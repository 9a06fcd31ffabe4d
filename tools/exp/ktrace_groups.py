"""Average duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV:  python3 tools/exp/ktrace_groups.py <dir> [name filter ...]"""
import csv, glob, os, sys
from collections import OrderedDict
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
assert files, "no kernel_trace.csv under " + d
groups = OrderedDict()
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if sys.argv[2:] and not any(k in n for k in sys.argv[2:]):
            continue
        key = (n.split("(")[0][-60:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        groups.setdefault(key, []).append(dur)
for (n, gs, wg), v in groups.items():
    v2 = sorted(v)
    print("%-60s grid %8s wg %4s  calls %5d  avg %8.1f us  median %8.1f  min %8.1f" % (n, gs, wg, len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
