#!/bin/bash
# same-box A/B of a stage line (item | user): the product library against tools/lab/libs/<name>.so, alternating, 3 rounds each
# usage (on the GPU box): bash tools/lab/stage_ab.sh <item|user> <lib name> <out tag>
OUT=gpurun_out/$3; mkdir -p $OUT
for r in 1 2 3; do
  timeout -k 10 200 python bench.py --workload $1 --no-cpu-baseline --steps 30 > $OUT/new_$r.json 2>> $OUT/err.txt || exit 1
  UNIREC_HIP_LIB=tools/lab/libs/$2.so timeout -k 10 200 python bench.py --workload $1 --no-cpu-baseline --steps 30 > $OUT/old_$r.json 2>> $OUT/err.txt || exit 1
done
python - <<PY
import json
for k in ("new_1","old_1","new_2","old_2","new_3","old_3"):
    d=json.loads(open("$OUT/%s.json"%k).read().strip().splitlines()[-1])
    print(k, d["value"], d["ms_per_step"])
PY
