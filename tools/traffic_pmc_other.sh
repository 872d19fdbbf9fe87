#!/bin/bash
# HBM traffic of the FMLP and GRU4Rec step kernels (the same two PMC passes as tools/traffic_pmc.sh) -> gpurun_out/r<ROUND>/pmc_traffic_<model>_B256.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-5}; mkdir -p $O
run() {  # tag, bench args
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_${tag}_$c
    timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pm_${tag}_$c -o t -- python $R/bench.py --no-graph --no-cpu-baseline --steps 6 --warmup 2 "$@" > /tmp/pm_${tag}_$c.log 2>&1
  done
  python $R/tools/traffic_pmc.py $(find /tmp/pm_${tag}_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pm_${tag}_WRITE_SIZE -name "*.db" | head -1) > $O/pmc_traffic_$tag.json
}
run fmlp_B256 --model fmlp
run gru4rec_B256 --model gru4rec
head -c 900 $O/pmc_traffic_fmlp_B256.json
