#!/bin/bash
# SQ counters of the projection GEMM kernels (persistent + generic) over the step's seven launches (separate --pmc passes, no
# trace domains) -> gpurun_out/<tag>/gemm_sq_pmc.txt.  usage: bash tools/lab/gemm_sq_pmc.sh <tag>
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
rm -rf /tmp/gsq1 /tmp/gsq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/gsq1 -o a --output-format csv -- python3 tools/kernel_bench.py gemm_step --B 64 --iters 0 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d /tmp/gsq2 -o b --output-format csv -- python3 tools/kernel_bench.py gemm_step --B 64 --iters 0 > $OUT/sq2.log 2>&1
for p in /tmp/gsq1 /tmp/gsq2; do echo "## $p"; python3 tools/pmc_summary.py $p gemm_; done > $OUT/gemm_sq_pmc.txt 2>&1
