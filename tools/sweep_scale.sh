R=$GRAFT_REPO_ROOT
show() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernel_us_per_step']
print(sys.argv[1], round(j['value']), round(j['ms_per_step'],4), {a: k[a] for a in ('post_fwd','post_mid','post_bwd','wgrad_fused','embqkv_fwd')})" "$1"; }
for bm in 32 64; do DR4SR_BM=$bm python $R/bench.py --steps 60 --warmup 10 --batch 8192 --no-cpu-baseline --no-strong 2>/dev/null | show "BM=$bm"; done
for gw in 32 64 96; do DR4SR_WGRAD_GW=$gw python $R/bench.py --steps 60 --warmup 10 --batch 8192 --no-cpu-baseline --no-strong 2>/dev/null | show "GW=$gw"; done
