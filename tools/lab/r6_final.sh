#!/bin/bash
# the round's closing measurements on one box -> gpurun_out/r6_final/ (copied into profiles/r6_* afterwards)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_final; mkdir -p $OUT
timeout -k 10 560 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench failed"
# the whole bucket path on ONE rank over RCCL (UNIREC_DP_FORCE=1): what the data-parallel machinery itself costs, beside the plain line
UNIREC_DP_FORCE=1 MASTER_PORT=29611 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-stages --no-cpu-baseline > $OUT/dp1_forced_rccl_bench.json 2> $OUT/dp1_forced.err || echo "dp forced failed"
UNIREC_DP_FORCE=1 UNIREC_DP_COMM=native MASTER_PORT=29612 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-stages --no-cpu-baseline > $OUT/dp1_forced_native_bench.json 2>> $OUT/dp1_forced.err || echo "dp native failed"
bash tools/lab/prof_bench.sh r6_final joint_b64 --no-cpu-baseline --no-stages --steps 3 --warmup 2 > $OUT/joint_b64_prof.txt 2>&1
bash tools/lab/prof_bench.sh r6_final user_c3 --workload user --steps 10 --no-cpu-baseline > $OUT/user_c3_prof.txt 2>&1
bash tools/lab/prof_bench.sh r6_final item_c2 --workload item --steps 20 --no-cpu-baseline > $OUT/item_c2_prof.txt 2>&1
bash tools/lab/gemm_pmc.sh r6_final > $OUT/gemm_pmc.txt 2>&1
tail -c 300 $OUT/bench_default.json; head -14 $OUT/joint_b64_prof.txt; cat $OUT/dp1_forced.err | tail -3
