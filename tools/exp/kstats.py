import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in sys.argv[2:]):
        print("   %-70s calls %4s avg %.1f us" % (n.split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3))
