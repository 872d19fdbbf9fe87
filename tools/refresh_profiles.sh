#!/bin/bash
# Regenerates the round's measurement artefacts on the GPU box into gpurun_out/r<ROUND>/ (copy what you keep into profiles/).
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-5}; mkdir -p $O
trace() {  # tag, bench args...
  tag=$1; shift
  rm -rf /tmp/kt_$tag
  timeout 400 rocprofv3 --kernel-trace -d /tmp/kt_$tag -o t -- python $R/bench.py --no-cpu-baseline --no-throughput-mode --no-strong --no-deterministic-leg "$@" > /tmp/kt_$tag.log 2>&1
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  python $R/tools/kstat.py $db 24 > $O/kernels_$tag.txt
  python $R/tools/ktimeline.py $db 100 > $O/timeline_$tag.txt
}
trace sasrec_B256 --steps 200 --warmup 20
trace sasrec_dense_B256 --steps 200 --warmup 20 --dense
trace sasrec_B8192 --steps 60 --warmup 10 --batch 8192
trace sasrec_dense_B8192 --steps 40 --warmup 10 --batch 8192 --dense
# the literal rocprofv3 --stats summary of the default bench command
rm -rf /tmp/st_default; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_default -o d -- python $R/bench.py --no-cpu-baseline --no-throughput-mode --no-strong --no-deterministic-leg > /tmp/st_default.log 2>&1
cp $(find /tmp/st_default -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_default.csv
cd $R
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
timeout 300 python bench.py --dense --no-cpu-baseline --no-throughput-mode --no-strong 2>/dev/null | tail -1 > $O/bench_sasrec_dense.json
timeout 300 python bench.py --batch 8192 --dense --no-cpu-baseline --steps 60 2>/dev/null | tail -1 > $O/bench_sasrec_B8192_dense.json
timeout 400 python bench.py --model gru4rec 2>/dev/null | tail -1 > $O/bench_gru4rec.json
timeout 300 python bench.py --model fmlp --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fmlp.json
timeout 300 python bench.py --model metamodel 2>/dev/null | tail -1 > $O/bench_metamodel.json
timeout 300 python bench.py --model cl4srec 2>/dev/null | tail -1 > $O/bench_cl4srec.json
timeout 300 python bench.py --embed-dim 128 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_sasrec_d128.json
ls -la $O | tail -20
