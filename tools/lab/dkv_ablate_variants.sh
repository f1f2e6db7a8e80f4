#!/bin/bash
# Lab: timing-only builds of the dK/dV loop (results WRONG): UR_ASMGEN_ABLATE tags (dma, soft, frag, mfma; '+' joins) -> tools/lab/libs/dkv_abl_<tag>.so
set -e
cd /root/repo
for t in "$@"; do
  mkdir -p /tmp/dkvabl/$t
  abl=${t//+/,}; [ "$t" = base ] && abl=""
  UR_ASMGEN_ABLATE=$abl python - <<PY
import sys
sys.path.insert(0, "tools/asmgen")
import emit
open("/tmp/dkvabl/$t/dkv.h", "w").write(emit.dkv_header())
PY
  tools/lab/lib_variant.sh attn dkv_abl_$t -DUR_ATTN_DKV_C128_HDR="\"/tmp/dkvabl/$t/dkv.h\"" >/dev/null
  echo built dkv_abl_$t
done
