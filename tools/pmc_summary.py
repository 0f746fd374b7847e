"""Aggregate rocprofv3 --pmc output: mean counter value per kernel launch.

usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [substring filter on kernel name]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if flt and flt not in k:
                    continue
                k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                k = k[-64:] if os.environ.get("PMC_KEEP_TEMPLATE") else k.split("<")[0][-48:]  # (template arguments kept on request)
                c = row.get("Counter_Name")
                a = acc[k][c]
                a[0] += float(row.get("Counter_Value", 0))
                a[1] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            print(f"   {c:28s} mean/launch {s / max(n, 1):16.1f}   launches {n}")


if __name__ == "__main__":
    main()
