#!/bin/bash
# Lab: the product library + round 5's wave-specialised GEMM (tools/lab/gemm_ws.hip) -> tools/lab/libs/gemm_ws.so; ur_gemm_persistent_mode(2) selects it.
# (It is correct and 4-18 % slower than the persistent kernel on every C4 shape, profiles/r5_gemm_ws_ab.txt: not part of the product build.)
set -e
cd /root/repo
make -C unirec_amd/csrc -j8 >/dev/null
mkdir -p tools/lab/libs /tmp/libvar_ws
cp build/obj/*.o /tmp/libvar_ws/
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -DUR_LAB=1"
/opt/rocm/bin/hipcc $F -c unirec_amd/csrc/gemm_pers.hip -o /tmp/libvar_ws/gemm_pers.o
/opt/rocm/bin/hipcc $F -x hip -c tools/lab/gemm_ws.hip -o /tmp/libvar_ws/gemm_ws.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libs/gemm_ws.so /tmp/libvar_ws/*.o
ls -la tools/lab/libs/gemm_ws.so
