"""Lab: per-kernel totals of a rocprofv3 results database (rocprofv3 --kernel-trace --stats writes <name>_results.db), per step."""
import sqlite3, sys
db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average from top_kernels"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / steps:.1f} us per step over {steps:g} steps")
for n, c, d, a in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{n[:84]:84s} {c / steps:7.1f}/step x {a:8.1f} us = {d / steps:8.1f} us/step")
