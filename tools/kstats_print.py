import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.2f" % (tot / 1e6))
for r in rows[:22]:
    print("%9.2f ms %6d calls  avg %9.3f ms  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]),
                                                    float(r["AverageNs"]) / 1e6, r["Name"][:80]))
