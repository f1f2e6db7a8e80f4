#!/bin/bash
# Lab: the SwiGLU-backward / masked-LoRA dX launches of the persistent GEMM with whole XCDs (UR_PERS_STAGGER_XCD) or 8-workgroup
# groups (UR_PERS_STAGGER) started some cycles apart.  Needs tools/lab/libs/pers_lab.so (lib_variant.sh gemm_pers pers_lab -DUR_LAB=1).
OUT=gpurun_out/$1; mkdir -p $OUT
export UNIREC_HIP_LIB=tools/lab/libs/pers_lab.so
for v in 0 1500 3000 6000 12000; do
  echo "== UR_PERS_STAGGER_XCD=$v" >> $OUT/stagger.txt
  UR_PERS_STAGGER_XCD=$v timeout -k 10 200 python3 tools/lab/gemm_pers_ab.py --only s,d --rounds 4 >> $OUT/stagger.txt 2>&1 || exit 1
done
for v in 400 1500; do
  echo "== UR_PERS_STAGGER=$v" >> $OUT/stagger.txt
  UR_PERS_STAGGER=$v timeout -k 10 200 python3 tools/lab/gemm_pers_ab.py --only s,d --rounds 4 >> $OUT/stagger.txt 2>&1 || exit 1
done
cat $OUT/stagger.txt
