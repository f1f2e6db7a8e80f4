#!/bin/bash
# Lab: libraries whose dK/dV loop reads its row fragments UR_DKV_LEAD_ROW MFMA slots ahead -> tools/lab/libs/dkv_lead<N>.so
set -e
cd /root/repo
for L in "$@"; do
  mkdir -p /tmp/dkvlead/$L
  UR_DKV_LEAD_ROW=$L python - <<PY
import sys
sys.path.insert(0, "tools/asmgen")
import emit
open("/tmp/dkvlead/$L/dkv.h", "w").write(emit.dkv_header())
PY
  tools/lab/lib_variant.sh attn dkv_lead$L -DUR_ATTN_DKV_C128_HDR="\"/tmp/dkvlead/$L/dkv.h\"" >/dev/null
  echo built dkv_lead$L
done
