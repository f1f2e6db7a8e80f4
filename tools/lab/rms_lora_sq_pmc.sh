#!/bin/bash
# SQ busy / wait / issue counters of rms_lora_kernel (separate --pmc passes) -> gpurun_out/<tag>/rms_lora_sq_pmc.txt
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
rm -rf /tmp/pmc_rl1 /tmp/pmc_rl2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d /tmp/pmc_rl1 -o p --output-format csv -- python3 tools/kernel_bench.py rmslora --B 64 --S 2048 --iters 2 > $OUT/rl1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_WAVES -d /tmp/pmc_rl2 -o p --output-format csv -- python3 tools/kernel_bench.py rmslora --B 64 --S 2048 --iters 2 > $OUT/rl2.log 2>&1
(python3 tools/pmc_summary.py /tmp/pmc_rl1 rms_lora; python3 tools/pmc_summary.py /tmp/pmc_rl2 rms_lora) > $OUT/rms_lora_sq_pmc.txt 2>&1
cat $OUT/rms_lora_sq_pmc.txt
