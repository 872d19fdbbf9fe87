#!/bin/bash
# per-kernel trace + timeline of one secondary workload: tools/trace_model.sh <model> [bench args]  ->  gpurun_out/r<ROUND>/kernels_<model>.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-3}; mkdir -p $O
m=$1; shift
rm -rf /tmp/kt_$m
timeout 400 rocprofv3 --kernel-trace -d /tmp/kt_$m -o t -- python $R/bench.py --model $m --no-cpu-baseline --no-throughput-mode --no-strong "$@" > /tmp/kt_$m.log 2>&1
db=$(find /tmp/kt_$m -name "*.db" | head -1)
python $R/tools/kstat.py $db 40 > $O/kernels_$m.txt
python $R/tools/ktimeline.py $db 160 > $O/timeline_$m.txt
[ "$m" = metamodel ] && python $R/tools/kouter.py $db > $O/outer_$m.txt
head -45 $O/kernels_$m.txt
