#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace --stats run: reads the rocpd
sqlite database rocprofv3 writes (ROCm 7.2 default output) and prints / writes the CSV kept under profiles/.
Usage: python tools/rocprof_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
for n, c, t, a, lo, hi in rows:
    out.append((n, c, int(t), round(a, 1), round(100.0 * t / tot, 3), int(lo), int(hi)))
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
