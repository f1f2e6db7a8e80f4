#!/bin/bash
# Lab: build ablated variants of the GEMM (results are WRONG by construction; timing only) and time them.
# Run on the GPU box from the repo root.
set -e
for v in 0 1 2 3; do
  mkdir -p /tmp/abl$v
  for f in unirec_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    if [ $b = gemm ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -DUR_GEMM_ABLATE=$v -c $f -o /tmp/abl$v/$b.o 2>/dev/null
    else
      cp build/obj/$b.o /tmp/abl$v/$b.o 2>/dev/null || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -c $f -o /tmp/abl$v/$b.o 2>/dev/null
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abl$v/lib.so /tmp/abl$v/*.o 2>/dev/null
  echo "== ablate $v (0 full, 1 no DMA, 2 no MFMA, 3 no barrier)"
  UNIREC_HIP_LIB=/tmp/abl$v/lib.so python tools/kernel_bench.py gemm --B 64 --iters 5 | head -3
done
