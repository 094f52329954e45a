"""Summarise rocprofv3 --pmc counter CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict


def main(root, filt=""):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if filt and filt not in name:
                continue
            acc[name[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} n={len(v):4d} mean={sum(v) / len(v):.4g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
