"""Per-step view of a rocprofv3 kernel_stats.csv: python tools/kstats_step.py <csv> <steps> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"device time {tot / 1e6:.2f} ms = {tot / steps / 1e6:.3f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for r in rows[:top]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']) / steps:7.1f}/step {int(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step "
          f"{float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):5.1f}%")
