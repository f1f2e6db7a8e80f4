#!/bin/bash
# Lab: the product library with ONE source rebuilt under extra flags -> tools/lab/libs/<out>.so (built here, travels with gpurun;
# use with UNIREC_HIP_LIB=tools/lab/libs/<out>.so).  usage: tools/lab/lib_variant.sh <source stem> <out> <flags...>
set -e
cd /root/repo
f=$1; out=$2; shift 2
make -C unirec_amd/csrc -j8 >/dev/null
mkdir -p tools/lab/libs /tmp/libvar_$out
cp build/obj/*.o /tmp/libvar_$out/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form "$@" -c unirec_amd/csrc/$f.hip -o /tmp/libvar_$out/$f.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libs/$out.so /tmp/libvar_$out/*.o
ls -la tools/lab/libs/$out.so
