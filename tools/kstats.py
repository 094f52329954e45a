"""Print the top kernels of a rocprofv3 --stats CSV: python tools/kstats.py <kernel_stats.csv> [n_forwards] [n_rows]"""
import csv
import sys

rows = list(csv.DictReader(ln for ln in open(sys.argv[1]) if not ln.startswith("#")))
nf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6:.2f} ms  ({tot / nf / 1e6:.2f} ms per forward)")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
    print(f"{r['Name'][:78]:78s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):5.1f}%")
