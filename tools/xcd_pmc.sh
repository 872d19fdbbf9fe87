#!/bin/bash
# L2 evidence for the XCD-aware tile order: bytes the L2s fetch (FETCH_SIZE, KB per dispatch as rocprofv3 reports it) and L2 hits / misses per launch
# of the latency regime's tile kernels, default order against DR4SR_TILE_ORDER_PLAIN=1 (separate PMC passes) -> gpurun_out/r<ROUND>/xcd_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-4}; mkdir -p $O
: > $O/xcd_pmc.txt
for order in xcd plain; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo ${order}_$c | tr ' ' '_')
    rm -rf /tmp/xp_$tag
    if [ $order = plain ]; then export DR4SR_TILE_ORDER_PLAIN=1; else unset DR4SR_TILE_ORDER_PLAIN; fi
    timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/xp_$tag -o t -- python $R/bench.py --no-graph --no-cpu-baseline --no-throughput-mode --no-strong --no-deterministic-leg --steps 40 --warmup 5 ${ARGS:-} > /tmp/xp_$tag.log 2>&1
    echo "== order=$order counters=$c ${ARGS:-}" >> $O/xcd_pmc.txt
    python $R/tools/pmcstat.py $(find /tmp/xp_$tag -name "*.db" | head -1) k_embqkv_fwd k_post_fwd k_post_mid k_post_bwd k_wgrad_blk >> $O/xcd_pmc.txt 2>&1
  done
done
cat $O/xcd_pmc.txt
