#!/bin/bash
# Lab: ablated builds of the persistent GEMM (UR_PERS_ABLATE; WRONG results, timing only).  Build here (hipcc cross-compiles),
# the libraries travel under tools/lab/libs/; time them on the GPU box with
#   for v in 0 1 2 3 4; do UNIREC_HIP_LIB=tools/lab/libs/pers_abl$v.so python tools/lab/gemm_pers_ab.py --ksweep; done
set -e
cd /root/repo
make -C unirec_amd/csrc -j8 >/dev/null
mkdir -p tools/lab/libs
for v in "$@"; do
  mkdir -p /tmp/pabl$v
  cp build/obj/*.o /tmp/pabl$v/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -DUR_PERS_ABLATE=$v -c unirec_amd/csrc/gemm_pers.hip -o /tmp/pabl$v/gemm_pers.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libs/pers_abl$v.so /tmp/pabl$v/*.o
done
ls -la tools/lab/libs/
