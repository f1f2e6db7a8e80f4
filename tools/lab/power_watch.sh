#!/bin/bash
# Lab: sample socket power / shader clock (rocm-smi, 5 Hz) while a command runs.  tools/lab/power_watch.sh <tag> <cmd...>
tag=$1; shift
out=gpurun_out/power_$tag.txt
mkdir -p gpurun_out
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import sys, json, time
try:
    d = json.load(sys.stdin)
    c = d.get('card0', {})
    pw = [v for k, v in c.items() if 'ower' in k]
    sc = [v for k, v in c.items() if 'sclk' in k]
    mc = [v for k, v in c.items() if 'mclk' in k]
    print('%.2f' % time.time(), 'power', pw, 'sclk', sc, 'mclk', mc, flush=True)
except Exception as e:
    print('err', e, flush=True)
"; sleep 0.1; done ) > $out 2>&1 &
wpid=$!
"$@"
rc=$?
kill $wpid
exit $rc
