#!/bin/bash
# the round's closing measurements on one box -> gpurun_out/r5_final/ (copied into profiles/r5_* afterwards)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_final; mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench failed"
bash tools/lab/prof_bench.sh r5_final joint_b64 --no-cpu-baseline --no-stages --steps 3 --warmup 2 > $OUT/joint_b64_prof.txt 2>&1
bash tools/lab/prof_bench.sh r5_final user_c3 --workload user --steps 10 --no-cpu-baseline > $OUT/user_c3_prof.txt 2>&1
bash tools/lab/prof_bench.sh r5_final item_c2 --workload item --steps 20 --no-cpu-baseline > $OUT/item_c2_prof.txt 2>&1
tail -c 300 $OUT/bench_default.json; head -14 $OUT/joint_b64_prof.txt
