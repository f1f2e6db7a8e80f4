#!/bin/bash
# Lab: build the library with in-kernel stamps in the GEMM and print the per-block timeline.  Run on the GPU box from the repo root.
set -e
mkdir -p /tmp/stamps
for f in unirec_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  X=""; [ $b = gemm ] && X="-DUR_GEMM_STAMPS=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form $X -c $f -o /tmp/stamps/$b.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/stamps/lib.so /tmp/stamps/*.o 2>/dev/null
for K in ${KS:-1024 3072}; do UNIREC_HIP_LIB=/tmp/stamps/lib.so python tools/lab/gemm_stamps.py $K 2>/dev/null; done
