#!/bin/bash
# HBM bytes of the K1 gather microbench: two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) -> gpurun_out/r<ROUND>/gather_pmc_raw.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-3}; mkdir -p $O
: > $O/gather_pmc_raw.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pg_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pg_$c -o t -- python $R/tools/gather_pmc.py > /tmp/pg_$c.log 2>&1
  db=$(find /tmp/pg_$c -name "*.db" | head -1)
  python $R/tools/pmcstat.py $db embed_dense >> $O/gather_pmc_raw.txt
done
cat $O/gather_pmc_raw.txt
