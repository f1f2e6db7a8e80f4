#!/bin/bash
# usage: bash tools/lab/trace_gaps.sh <tag> <bench args...>  -> gpurun_out/<tag>_gaps.txt
export TMPDIR=/tmp
TAG=$1; shift
rm -rf /tmp/prof_gaps
rocprofv3 --kernel-trace -d /tmp/prof_gaps -o p -- python3 bench.py "$@" > /tmp/prof_gaps.log 2>&1
db=$(find /tmp/prof_gaps -name "*.db" | head -1)
python3 tools/lab/trace_gaps.py $db > gpurun_out/${TAG}_gaps.txt 2>&1
cat gpurun_out/${TAG}_gaps.txt
