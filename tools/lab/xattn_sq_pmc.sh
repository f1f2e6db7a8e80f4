#!/bin/bash
# SQ / LDS counters of the user Q-Former's cross-attention kernels at the C3 shape -> gpurun_out/<tag>/xattn_sq_pmc.txt
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
rm -rf /tmp/pmc_x1 /tmp/pmc_x2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d /tmp/pmc_x1 -o p --output-format csv -- python3 tools/kernel_bench.py xattn --iters 1 > $OUT/x1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d /tmp/pmc_x2 -o p --output-format csv -- python3 tools/kernel_bench.py xattn --iters 1 > $OUT/x2.log 2>&1
(python3 tools/pmc_summary.py /tmp/pmc_x1 attn; python3 tools/pmc_summary.py /tmp/pmc_x2 attn) > $OUT/xattn_sq_pmc.txt 2>&1
cat $OUT/xattn_sq_pmc.txt
