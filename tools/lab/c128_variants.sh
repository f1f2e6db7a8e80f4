#!/bin/bash
# Lab: timing-only variants of the generated forward loop (results WRONG when ablated).  For each tag in "$@" (a UR_ASMGEN_ABLATE
# value, '+' for ','; "base" = the product schedule) builds tools/lab/libs/c128_<tag>.so.  Run here (no GPU needed), then on the box:
#   for t in ...; do UNIREC_HIP_LIB=tools/lab/libs/c128_$t.so python tools/kernel_bench.py attn --B 64; done
set -e
cd /root/repo
for t in "$@"; do
  mkdir -p /tmp/c128var/$t
  abl=${t//+/,}; [ "$t" = base ] && abl=""
  stamps=0; extra=""
  if [ "$t" = stamps ]; then abl=""; stamps=1; extra="-DUR_C128_STAMPS=1"; fi
  UR_ASMGEN_ABLATE=$abl UR_ASMGEN_STAMPS=$stamps python - <<PY
import sys
sys.path.insert(0, "tools/asmgen")
import emit
open("/tmp/c128var/$t/fwd.h", "w").write(emit.fwd_header())
open("/tmp/c128var/$t/dq.h", "w").write(emit.dq_header())
open("/tmp/c128var/$t/dkv.h", "w").write(emit.dkv_header())
PY
  tools/lab/lib_variant.sh attn c128_$t -DUR_ATTN_FWD_C128_HDR="\"/tmp/c128var/$t/fwd.h\"" -DUR_ATTN_DQ_C128_HDR="\"/tmp/c128var/$t/dq.h\"" -DUR_ATTN_DKV_C128_HDR="\"/tmp/c128var/$t/dkv.h\"" $extra >/dev/null
  echo built c128_$t
done
