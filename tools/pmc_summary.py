#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, per counter: mean over dispatches.
Usage: python tools/pmc_summary.py <dir> [name-substring]"""
import csv, glob, sys, collections, re
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub and sub not in k: continue
        k = re.sub(r"\(anonymous namespace\)::", "", k)[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
