#!/bin/bash
# Lab: ablated builds of lora_reduce_kernel (results WRONG when != 0; timing only): tools/kernel_bench.py lora per variant.
#   bash tools/lab/red_ablate.sh <outdir> "<variants>"
OUT=gpurun_out/$1; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p /tmp/base
for f in unirec_amd/csrc/*.hip; do b=$(basename $f .hip); [ $b = lora ] && continue; /opt/rocm/bin/hipcc $FLAGS -c $f -o /tmp/base/$b.o 2>/dev/null & done
for v in $2; do mkdir -p /tmp/red$v; /opt/rocm/bin/hipcc $FLAGS -DUR_RED_ABLATE=$v -c unirec_amd/csrc/lora.hip -o /tmp/red$v/lora.o 2>/dev/null & done
wait
for v in $2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/red$v/lib.so /tmp/red$v/lora.o /tmp/base/*.o 2>/dev/null
  echo "== UR_RED_ABLATE=$v"
  UNIREC_HIP_LIB=/tmp/red$v/lib.so python3 tools/kernel_bench.py lora --B 64 --S 2048 --iters 20 2>/dev/null | grep "reduce" | tee $OUT/red$v.txt
done
