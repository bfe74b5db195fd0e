"""Print a rocprofv3 kernel_stats.csv as ms/step.  argv: csv steps [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", round(tot / 1e6 / steps, 3))
for r in rows[:top]:
    print(f"{r['Name'][:90]:90s} n/step {int(r['Calls']) / steps:6.2f} avg_us {float(r['AverageNs']) / 1e3:8.1f} ms/step {float(r['TotalDurationNs']) / 1e6 / steps:6.3f}")
