#!/bin/bash
# Lab: build with -DUR_DKV2_STAMPS=1 and print the phase timeline.  Run on the GPU box from the repo root.
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iunirec_amd/csrc -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p /tmp/stamps
for f in unirec_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b = attn ]; then /opt/rocm/bin/hipcc $FLAGS -DUR_DKV2_STAMPS=1 $EXTRA -c $f -o /tmp/stamps/$b.o 2>/dev/null & else /opt/rocm/bin/hipcc $FLAGS -c $f -o /tmp/stamps/$b.o 2>/dev/null & fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/stamps/lib.so /tmp/stamps/*.o
UNIREC_HIP_LIB=/tmp/stamps/lib.so python3 tools/lab/attn_stamps.py
