#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
for S in 1024 2048 4096; do
  rm -rf /tmp/sfit_$S
  S=$S rocprofv3 --kernel-trace --stats -d /tmp/sfit_$S -o p -- python3 tools/lab/c128_sfit.py > $OUT/sfit_$S.log 2>&1
  db=$(find /tmp/sfit_$S -name "*.db" | head -1)
  python3 tools/rocprof_stats.py $db $OUT/sfit_$S.csv > /dev/null
done
python3 - $OUT <<'PY'
import csv, sys
out = sys.argv[1]
t = {}
for S in (1024, 2048, 4096):
    for r in csv.DictReader(open(f"{out}/sfit_{S}.csv")):
        for k in ("attn_fwd_c128", "attn_bwd_dq_c128", "attn_bwd_dkv_c128"):
            if k in r["Name"]: t[(k, S)] = float(r["AverageNs"]) / 1e3
for k in ("attn_fwd_c128", "attn_bwd_dq_c128", "attn_bwd_dkv_c128"):
    a, b, c = t[(k, 1024)], t[(k, 2048)], t[(k, 4096)]
    # time = n_units * (fixed + per_tile * tiles): units per CU constant, tiles per unit double with S
    slope = (c - b) / 2.0; fixed = b - 2 * slope       # in us of kernel time: the part that does not scale with S at S = 2048 is `fixed`
    print(f"{k}: S 1024 {a:.0f} us, 2048 {b:.0f} us, 4096 {c:.0f} us; at S 2048: {fixed:.0f} us do not scale with S ({100*fixed/b:.0f} %), {2*slope:.0f} us do")
PY
