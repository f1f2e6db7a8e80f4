#!/bin/bash
# Lab: ablated builds of attn_bwd_dkv2_kernel (results WRONG by construction when != 0; timing only), per-kernel averages
# from rocprofv3 --kernel-trace --stats over tools/kernel_bench.py attn --B 64.  Run on the GPU box from the repo root:
#   bash tools/lab/dkv2_ablate.sh <outdir> "<variants>" [MACRO]     e.g.  r2e "0 1 2 3 4 5" UR_DKV2_ABLATE
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
MACRO=${3:-UR_DKV2_ABLATE}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p /tmp/base
for f in unirec_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  [ $b = attn ] && continue
  /opt/rocm/bin/hipcc $FLAGS -c $f -o /tmp/base/$b.o 2>/dev/null &
done
for v in $2; do
  mkdir -p /tmp/abl$v
  /opt/rocm/bin/hipcc $FLAGS -D$MACRO=$v -c unirec_amd/csrc/attn.hip -o /tmp/abl$v/attn.o 2>/dev/null &
done
wait
for v in $2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abl$v/lib.so /tmp/abl$v/attn.o /tmp/base/*.o 2>/dev/null
  export UNIREC_HIP_LIB=/tmp/abl$v/lib.so
  rocprofv3 --kernel-trace --stats -d /tmp/prof$v -o p -- python3 tools/kernel_bench.py attn --B 64 --iters 3 > $OUT/abl$v.log 2>&1
  db=$(find /tmp/prof$v -name "*.db" | head -1)
  python3 tools/rocprof_stats.py $db $OUT/abl$v.csv
  echo "== $MACRO=$v"
  python3 - $OUT/abl$v.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_" in r["Name"]:
        n = r["Name"][r["Name"].index("attn_"):][:44]
        print(f"   {n:46s} calls {r['Calls']:>3s}  avg {float(r['AverageNs'])/1000:8.1f} us  min {float(r['MinNs'])/1000:8.1f}")
PY
done
