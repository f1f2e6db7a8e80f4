#!/usr/bin/env python3
"""Lab: where a step's wall time goes on the GPU, from a rocprofv3 --kernel-trace database: per stream (queue) busy time, the union of all
streams' busy intervals, and the idle gaps of the union (GPU idle: launch latency / host-bound / dependency bubbles).
Usage: python tools/lab/trace_gaps.py <results.db> [first_fraction last_fraction]  (default: the middle 60 % of the trace = steady state)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(db.execute(f"select start, end, {qcol if qcol else '0'}, name from kernels order by start"))
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
sel = [r for r in rows if r[0] >= a and r[1] <= b]
span = b - a
print(f"window {span / 1e6:.2f} ms, {len(sel)} kernels, columns {cols}")
per = {}
for s, e, q, n in sel:
    per.setdefault(q, [0, 0])
    per[q][0] += e - s; per[q][1] += 1
for q, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"  queue {q}: busy {t / span * 100:5.1f} %  {c} kernels, avg {t / c / 1e3:.1f} us")
# union of busy intervals
ev = sorted((s, e, n) for s, e, _, n in sel)
busy, gaps, cs, ce, last = 0, [], ev[0][0], ev[0][1], ev[0][2]
big = []
for s, e, n in ev[1:]:
    if s > ce:
        busy += ce - cs; gaps.append(s - ce); big.append((s - ce, last, n)); cs, ce, last = s, e, n
    else:
        if e > ce:
            ce, last = e, n
busy += ce - cs
print(f"  any kernel running: {busy / span * 100:.1f} % of the window; {len(gaps)} idle gaps, sum {sum(gaps) / 1e6:.2f} ms "
      f"({sum(gaps) / span * 100:.1f} %), median {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, > 20 us: {sum(1 for g in gaps if g > 20000)} gaps = {sum(g for g in gaps if g > 20000) / 1e6:.2f} ms")
agg = {}
for g, before, after in big:
    if g > 15000:
        k = (before.replace("(anonymous namespace)::", "")[:48], after.replace("(anonymous namespace)::", "")[:48])
        v = agg.setdefault(k, [0, 0]); v[0] += g; v[1] += 1
for (bf, af), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  gaps > 15 us: {c:4d} x avg {t / c / 1e3:7.1f} us = {t / 1e6:6.2f} ms   after [{bf}] before [{af}]")
