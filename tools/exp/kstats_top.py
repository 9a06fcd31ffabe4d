"""Top kernels of a rocprofv3 --stats run: python3 tools/exp/kstats_top.py <dir> [n] [divide-by-steps]"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 25; div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.1f us; launches per step %.1f" % (tot / 1e3 / div, sum(int(r["Calls"]) for r in rows) / div))
for r in rows[:n]:
    print("%-64s calls/step %6.1f avg %8.1f us  per step %8.1f us  %5.1f%%" % (r["Name"].split("(")[0][-64:], int(r["Calls"]) / div, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / div, 100 * float(r["TotalDurationNs"]) / tot))
