#!/bin/bash
# Round-2 PMC passes over the three causal attention kernels at the C4 shape (B=64: the C4 shape; per-workgroup
# behaviour does not depend on B).  Separate passes: --pmc never together with trace domains.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/$1
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
python3 tools/kernel_bench.py attn --B 64 --iters 3 > $OUT/attn_plain.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d $OUT/p1 -o p1 --output-format csv -- python3 tools/kernel_bench.py attn --B 64 --iters 2 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM -d $OUT/p2 -o p2 --output-format csv -- python3 tools/kernel_bench.py attn --B 64 --iters 2 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $OUT/p3 -o p3 --output-format csv -- python3 tools/kernel_bench.py attn --B 64 --iters 2 > $OUT/p3.log 2>&1
for p in p1 p2 p3; do echo "## $p"; python3 tools/pmc_summary.py $OUT/$p attn_; done > $OUT/attn_pmc.txt 2>&1
