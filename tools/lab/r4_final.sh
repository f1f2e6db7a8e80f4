#!/bin/bash
# the round's closing measurements on one box -> gpurun_out/r4_final/
OUT=gpurun_out/r4_final; mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench failed"
timeout -k 10 200 python tools/lab/lora_shapes_bench.py > $OUT/lora_shapes.txt 2>&1
WHAT=fwd UNIREC_HIP_LIB=tools/lab/libs/c128_stamps.so timeout -k 10 200 python tools/lab/c128_stamps.py > $OUT/stamps_fwd.txt 2>&1
WHAT=bwd UNIREC_HIP_LIB=tools/lab/libs/c128_stamps.so timeout -k 10 200 python tools/lab/c128_stamps.py > $OUT/stamps_bwd.txt 2>&1
tail -c 400 $OUT/bench_default.json
