cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pm_$c -o t -- python $R/bench.py --no-graph --no-cpu-baseline --no-throughput-mode --no-strong --no-deterministic-leg --steps 6 --warmup 2 --batch 8192 --dense > /tmp/pm_$c.log 2>&1
done
python $R/tools/traffic_pmc.py $(find /tmp/pm_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*.db" | head -1) | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if any(s in k for s in ('k_wt_','k_post','k_embqkv','k_qkv_embed','k_wgrad')): print(k, v)
"
