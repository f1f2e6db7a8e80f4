#!/bin/bash
# same-box A/B of the headline step: the product library against tools/lab/libs/<name>.so (lib_at_commit.sh), alternating, 2 rounds each
# usage (on the GPU box): bash tools/lab/bench_ab.sh <lib name> <out tag>
OUT=gpurun_out/$2; mkdir -p $OUT
for r in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-stages --steps 4 > $OUT/new_$r.json 2>> $OUT/err.txt || exit 1
  UNIREC_HIP_LIB=tools/lab/libs/$1.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-stages --steps 4 > $OUT/old_$r.json 2>> $OUT/err.txt || exit 1
done
python - <<PY
import json
for k in ("new_1","old_1","new_2","old_2"):
    d=json.loads(open("$OUT/%s.json"%k).read().strip().splitlines()[-1])
    print(k, d["value"], d["ms_per_step"], "attn bwd", d["attention"]["bwd"]["avg_launch_ms"], "rope", d["roofline"]["hbm_bound_families"].get("qknorm_rope_bwd",{}).get("ms_per_step"))
PY
