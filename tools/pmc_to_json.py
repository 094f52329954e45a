"""Per-kernel counter table of a whole evaluation from the passes of tools/pmc_collect.sh, and (with --write) the entry of the
mode (GECCO_PRECISION, default w2) in profiles/gemm_hbm_traffic.json that bench.py quotes.

  python tools/pmc_to_json.py <pmc dir> <evaluations in the run> <tag> [--write]

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, "HBM"); matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def main():
    root, nev, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    mode = os.environ.get("GECCO_PRECISION", "w2")
    tree = os.environ.get("GECCO_TREE", "unknown")
    acc = load(root)
    rows = []
    for k, cs in acc.items():
        n = max(len(v) for v in cs.values())
        mean = lambda c: (sum(cs[c]) / len(cs[c])) if c in cs else 0.0
        by = (2 * mean("FETCH_SIZE") + mean("WRITE_SIZE")) * 1024
        busy = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * mean("GRBM_GUI_ACTIVE") / 8) if mean("GRBM_GUI_ACTIVE") else 0.0
        ldsc = mean("SQ_LDS_BANK_CONFLICT") / mean("SQ_LDS_IDX_ACTIVE") if mean("SQ_LDS_IDX_ACTIVE") else 0.0
        rows.append((by * n / nev, short(k), n / nev, by, busy, ldsc, 2 * mean("FETCH_SIZE") * 1024, mean("WRITE_SIZE") * 1024))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"# {tag} (tree {tree}): counters per kernel over {nev} evaluations of the C2 forward ({mode} mode, one stream); HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE")
    print(f"# total HBM bytes per evaluation: {tot / 1e9:.2f} GB")
    print(f"{'kernel':64s} {'launches/eval':>13s} {'MB/launch':>10s} {'fetch MB':>9s} {'write MB':>9s} {'MB/eval':>9s} {'mfma busy':>9s} {'lds confl':>9s}")
    for t, k, n, by, busy, ldsc, fe, wr in rows[:24]:
        print(f"{k[:64]:64s} {n:13.1f} {by / 1e6:10.1f} {fe / 1e6:9.1f} {wr / 1e6:9.1f} {t / 1e6:9.1f} {busy:9.3f} {ldsc:9.3f}")
    if "--write" in sys.argv:
        pj = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "gemm_hbm_traffic.json")
        d = json.load(open(pj))
        dom = max((r for r in rows), key=lambda r: r[0])
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import __graft_entry__ as ge   # the content hash of the kernel sources the profiled library was built from (bench.py checks it)
        d[mode] = {
            "tree": tree, "sources_sha": ge.built_sources_sha(),
            "kernel": dom[1] + f" (the kernel with the largest HBM traffic per evaluation of the {mode} mode)",
            "source": f"profiles/{tag}_forward_pmc_summary.txt (tools/pmc_collect.sh on tree {tree}: separate --pmc passes over 2 whole evaluations, one stream)",
            "bytes_per_launch": dom[3], "mfma_busy": dom[4],
            "bytes_per_evaluation": tot,
            "per_kernel": {k: {"launches_per_evaluation": n, "bytes_per_launch": by, "mfma_busy": busy} for t, k, n, by, busy, ldsc, fe, wr in rows[:12]},
        }
        json.dump(d, open(pj, "w"), indent=1)


if __name__ == "__main__":
    main()
