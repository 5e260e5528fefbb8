#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, mean counter value per dispatch.
usage: pmc_summary.py <dir-or-csv> [kernel-substring]"""
import collections, csv, glob, os, sys
def main():
    path = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} mean {sum(v)/len(v):.6g}  (n={len(v)})")
if __name__ == "__main__":
    main()
