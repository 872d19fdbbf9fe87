R=${GRAFT_REPO_ROOT:-/root/repo}
show() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernel_us_per_step']
print(sys.argv[1], round(j['value']), round(j['ms_per_step'],4), {a: k[a] for a in ('attn_fwd','attn_bwd')})" "$1"; }
python $R/bench.py --embed-dim 128 --steps 40 --warmup 10 --batch 8192 --no-cpu-baseline --no-strong 2>/dev/null | show "d128 tiny"
DR4SR_ATTN_NOTINY=1 python $R/bench.py --embed-dim 128 --steps 40 --warmup 10 --batch 8192 --no-cpu-baseline --no-strong 2>/dev/null | show "d128 notiny"
DR4SR_DE_ATOMIC=1 python $R/bench.py --embed-dim 128 --steps 40 --warmup 10 --batch 8192 --no-cpu-baseline --no-strong 2>/dev/null | show "d128 atomic-dE"
