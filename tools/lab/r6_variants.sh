#!/bin/bash
# Lab (round 6): generator variants of the three causal loops as whole libraries.  usage: tools/lab/r6_variants.sh name "ENV=..." [name "ENV=..."]...
# builds tools/lab/libs/r6_<name>.so from the generators run under that environment (UR_ASMGEN_BALANCE, UR_DKV_PRE_ADDR, UR_ASMGEN_STAMPS ...).
# Run here (no GPU needed); on the box: UNIREC_HIP_LIB=tools/lab/libs/r6_<name>.so python tools/lab/dkv_time.py
set -e
cd /root/repo
while [ $# -ge 2 ]; do
  t=$1; envs=$2; shift 2
  mkdir -p /tmp/r6var/$t
  extra=""
  case "$envs" in *UR_ASMGEN_STAMPS=1*) extra="-DUR_C128_STAMPS=1";; esac
  env $envs python - <<PY
import sys
sys.path.insert(0, "tools/asmgen")
import emit
open("/tmp/r6var/$t/fwd.h", "w").write(emit.fwd_header())
open("/tmp/r6var/$t/dq.h", "w").write(emit.dq_header())
open("/tmp/r6var/$t/dkv.h", "w").write(emit.dkv_header())
PY
  tools/lab/lib_variant.sh attn r6_$t -DUR_ATTN_FWD_C128_HDR="\"/tmp/r6var/$t/fwd.h\"" -DUR_ATTN_DQ_C128_HDR="\"/tmp/r6var/$t/dq.h\"" -DUR_ATTN_DKV_C128_HDR="\"/tmp/r6var/$t/dkv.h\"" $extra >/dev/null
  echo built r6_$t
done
