#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter rows per kernel name: pmc_sum.py <counter_collection.csv> [substring]."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in rows:
    name = r["Kernel_Name"]
    if want not in name:
        continue
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        n[name] += 1
for name, c in acc.items():
    print(f"{name[:90]}  ({n[name]} launches; per launch)")
    for k in sorted(c):
        print(f"  {k:28s} {c[k] / n[name]:.4g}")
