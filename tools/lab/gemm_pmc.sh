#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the seven projection launches of the step (separate --pmc passes) -> gpurun_out/<tag>/gemm_pmc.json
# (--iters 0 ends kernel_bench with a division by zero AFTER the measured dispatches: its exit code is ignored)
# usage: bash tools/lab/gemm_pmc.sh <tag>      (UR_GEMM_CW=0 in the environment: the plain tile order)
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f --output-format csv -- python3 tools/kernel_bench.py gemm_step --B 64 --iters 0 > $OUT/pmc_f.log 2>&1;
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w --output-format csv -- python3 tools/kernel_bench.py gemm_step --B 64 --iters 0 > $OUT/pmc_w.log 2>&1;
python3 tools/gemm_pmc.py /tmp/pmc_f /tmp/pmc_w $OUT/gemm_pmc.json
