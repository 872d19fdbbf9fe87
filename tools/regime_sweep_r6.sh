#!/bin/bash
# round 6: where do the at-scale forms (wave tiles + wave-per-tile attention) overtake the latency forms, and is the middle regime (at-scale tiles,
# one attention workgroup per sequence) still worth having?  ms per step, toys-shaped rows
mkdir -p gpurun_out
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --no-deterministic-leg --no-dp-leg --batch $2 --steps 200 --repeats 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f(T=%d)' % (d['ms_per_step'], d['valid_tokens_last_step']), end='  ')"; }
for B in ${RS_SIZES:-1024 1280 1536 1792 2048 2304 2560 2816}; do
  echo -n "B=$B default: "; run X=1 $B; echo -n "latency forms: "; run DR4SR_FORCE_SCALE=0 $B; echo -n "at scale + wave attention: "; run DR4SR_FORCE_SCALE=1 $B; echo -n "at scale + per-sequence attention: "; run "DR4SR_FORCE_SCALE=1 DR4SR_FORCE_ATTN_SPLIT=0" $B; echo
done 2>&1 | tee gpurun_out/regime_sweep_r6.txt
echo "== deterministic mode at B = 256 / 8192"
for B in 256 8192; do python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --no-dp-leg --batch $B --steps $([ $B = 256 ] && echo 300 || echo 100) --repeats 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($B, round(d['ms_per_step'],4), d.get('deterministic_mode'))"; done 2>&1 | tee -a gpurun_out/regime_sweep_r6.txt
