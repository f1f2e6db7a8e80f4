#!/bin/bash
# Lab: ablated builds of lora_bgrad (results are WRONG by construction; timing only).  Run on the GPU box from the repo root.
set -e
for v in ${VARIANTS:-0 1 2 3 4 5}; do
  mkdir -p /tmp/bga$v
  for f in unirec_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    if [ $b = lora ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -DUR_BG_ABLATE=$v -c $f -o /tmp/bga$v/$b.o 2>/dev/null
    else
      cp build/obj/$b.o /tmp/bga$v/$b.o
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/bga$v/lib.so /tmp/bga$v/*.o 2>/dev/null
  echo "== ablate $v (0 full, 1 no dB phase, 2 loads + tb MFMAs only, 3 no cross-wave exchange / barriers, 4 = 2 with unconditional loads, 5 = 4 without MFMAs)"
  UNIREC_HIP_LIB=/tmp/bga$v/lib.so python tools/kernel_bench.py lora --B 64 --iters 10 2>/dev/null | grep bgrad
done
