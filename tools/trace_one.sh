#!/bin/bash
# one kernel-trace + timeline of bench.py with the given args:  tools/trace_one.sh <tag> <bench args...>   -> gpurun_out/r<ROUND>/{kernels,timeline}_<tag>.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-5}; mkdir -p $O
tag=$1; shift
rm -rf /tmp/kt_$tag
timeout 400 rocprofv3 --kernel-trace -d /tmp/kt_$tag -o t -- python $R/bench.py --no-cpu-baseline --no-throughput-mode --no-strong --no-deterministic-leg "$@" > /tmp/kt_$tag.log 2>&1
db=$(find /tmp/kt_$tag -name "*.db" | head -1)
python $R/tools/kstat.py $db 24 > $O/kernels_$tag.txt
python $R/tools/ktimeline.py $db 100 > $O/timeline_$tag.txt
