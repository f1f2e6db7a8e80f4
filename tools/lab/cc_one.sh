#!/bin/bash
# compile ONE csrc file with the product flags and print the per-kernel resource usage (VGPRs, spills, scratch, LDS)
# usage: tools/lab/cc_one.sh gemm_pers [extra flags]
f=$1; shift
mkdir -p /root/repo/build/obj && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I/root/repo/unirec_amd/csrc -Wall -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Rpass-analysis=kernel-resource-usage "$@" -c /root/repo/unirec_amd/csrc/$f.hip -o /root/repo/build/obj/$f.o 2>&1 | grep -E "error|warning:|Function Name|VGPRs:|Spill|ScratchSize|Occupancy|SGPRs:|LDS Size" 
