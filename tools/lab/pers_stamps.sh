#!/bin/bash
# Lab: build the library with stamps around the n-th tile of every workgroup of the persistent GEMM (here; it travels under
# tools/lab/libs/), then on the GPU box: UNIREC_HIP_LIB=tools/lab/libs/pers_stamps.so python tools/lab/pers_stamps.py 1024 4096
set -e
cd /root/repo
make -C unirec_amd/csrc -j8 >/dev/null
mkdir -p tools/lab/libs /tmp/pstamps
cp build/obj/*.o /tmp/pstamps/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -DUR_PERS_STAMPS=${1:-6} $EXTRA -c unirec_amd/csrc/gemm_pers.hip -o /tmp/pstamps/gemm_pers.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libs/${OUT:-pers_stamps}.so /tmp/pstamps/*.o
