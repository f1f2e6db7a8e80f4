#!/bin/bash
# Lab: phase timeline of the dK/dV fast path for several ablated builds.  bash tools/lab/attn_stamps_abl.sh "0 6 7 8 9"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p /tmp/stb
for f in unirec_amd/csrc/*.hip; do
  b=$(basename $f .hip); [ $b = attn ] && continue
  /opt/rocm/bin/hipcc $FLAGS -c $f -o /tmp/stb/$b.o 2>/dev/null &
done
for v in $1; do /opt/rocm/bin/hipcc $FLAGS -DUR_DKV2_STAMPS=1 -DUR_DKV2_ABLATE=$v $EXTRA -c unirec_amd/csrc/attn.hip -o /tmp/stb_attn$v.o 2>/dev/null & done
wait
for v in $1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/stb_lib$v.so /tmp/stb_attn$v.o /tmp/stb/*.o
  echo "== UR_DKV2_ABLATE=$v $EXTRA"
  UNIREC_HIP_LIB=/tmp/stb_lib$v.so python3 tools/lab/attn_stamps.py 2>/dev/null | grep -v amdgpu.ids
done
