#!/bin/bash
# SQ counter passes over the three generated attention kernels (dense causal B 64 S 2048); separate --pmc passes, no trace domains.
# usage (on the GPU box): tools/lab/c128_pmc_all.sh <tag>   -> gpurun_out/<tag>/pmc.txt
set +e
export TMPDIR=/tmp
OUT=gpurun_out/$1
mkdir -p $OUT
export N=3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p1 -o p1 --output-format csv -- python3 tools/lab/c128_fb.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $OUT/p2 -o p2 --output-format csv -- python3 tools/lab/c128_fb.py > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_INSTS_SMEM -d $OUT/p3 -o p3 --output-format csv -- python3 tools/lab/c128_fb.py > $OUT/p3.log 2>&1
for kern in attn_fwd_c128 attn_bwd_dq_c128 attn_bwd_dkv_c128; do
  echo "# $kern"
  for p in p1 p2 p3; do echo "## $p"; python3 tools/pmc_summary.py $OUT/$p $kern; done
done > $OUT/pmc.txt 2>&1
cat $OUT/pmc.txt | head -90
