#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench.py run -> gpurun_out/<tag>/<name>.csv (per-kernel summary) + bench json line
# usage: bash tools/lab/prof_bench.sh <tag> <name> <bench args...>
export TMPDIR=/tmp
TAG=$1; NAME=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME -o p -- python3 bench.py "$@" > $OUT/$NAME.log 2>&1
db=$(find /tmp/prof_$NAME -name "*.db" | head -1)
python3 tools/rocprof_stats.py $db $OUT/${NAME}_kernel_stats.csv
grep '^{' $OUT/$NAME.log > $OUT/${NAME}_bench.json
python3 - $OUT/${NAME}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:22]:
    n = r["Name"].replace("(anonymous namespace)::", "")[:60]
    print(f"  {n:62s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f} us  {r['Percentage']:>6s} %")
PY
