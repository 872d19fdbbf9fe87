# latency vs at-scale regime by batch size (DR4SR_LATENCY_TMAX moves the boundary): bash tools/regime_sweep.sh [--dense]
run() { python bench.py --no-cpu-baseline --no-strong --no-throughput-mode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']), d['valid_tokens_last_step'])"; }
for b in ${BATCHES:-384 512 768 1024 1536 2048}; do for t in 1 1000000; do echo -n "B=$b latency_tmax=$t $* "; DR4SR_LATENCY_TMAX=$t run --batch $b --steps 60 "$@"; done; done
