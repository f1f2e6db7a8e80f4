#!/bin/bash
# Lab: the library as of <commit> (csrc + include) -> tools/lab/libs/<out>.so, for same-box A/B runs (UNIREC_HIP_LIB=...)
set -e
cd /root/repo
c=$1; out=$2
rm -rf /tmp/libat_$out && mkdir -p /tmp/libat_$out/src
git archive $c unirec_amd/csrc include | tar -x -C /tmp/libat_$out/src
mkdir -p /tmp/libat_$out/obj tools/lab/libs
for f in /tmp/libat_$out/src/unirec_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/tmp/libat_$out/src/include -I/tmp/libat_$out/src/unirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -c $f -o /tmp/libat_$out/obj/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/lab/libs/$out.so /tmp/libat_$out/obj/*.o
ls -la tools/lab/libs/$out.so
