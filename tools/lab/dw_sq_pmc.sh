#!/bin/bash
# LDS counters of the token-reduction (dW) GEMM launches -> gpurun_out/<tag>/dw_sq_pmc.txt
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
rm -rf /tmp/pmc_d2
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d /tmp/pmc_d2 -o p --output-format csv -- python3 tools/kernel_bench.py dw --iters 1 > $OUT/d2.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_d2 gemm > $OUT/dw_sq_pmc.txt 2>&1
cat $OUT/dw_sq_pmc.txt
