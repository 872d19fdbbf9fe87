run() { python bench.py --no-cpu-baseline --no-strong --no-throughput-mode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), round(d['value']), 'wgrad', k.get('wgrad_fused'), 'post_mid', k.get('post_mid'))"; }
echo -n "toys8192 "; run --batch 8192 --steps 100
echo -n "dense8192 "; run --batch 8192 --steps 40 --dense
