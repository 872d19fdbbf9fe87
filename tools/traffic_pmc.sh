#!/bin/bash
# HBM traffic of the training-step kernels: two PMC passes (FETCH_SIZE / WRITE_SIZE) per workload -> gpurun_out/r<ROUND>/pmc_traffic_<tag>.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-5}; mkdir -p $O
run() {  # tag, bench args
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_${tag}_$c
    timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pm_${tag}_$c -o t -- python $R/bench.py --no-graph --no-cpu-baseline --no-throughput-mode --no-strong --no-deterministic-leg --steps 6 --warmup 2 "$@" > /tmp/pm_${tag}_$c.log 2>&1
  done
  python $R/tools/traffic_pmc.py $(find /tmp/pm_${tag}_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pm_${tag}_WRITE_SIZE -name "*.db" | head -1) > $O/pmc_traffic_$tag.json
}
run B256_toys
run B8192_toys --batch 8192
run B8192_dense --batch 8192 --dense
head -c 600 $O/pmc_traffic_B8192_dense.json
