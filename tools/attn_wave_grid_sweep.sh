#!/bin/bash
# sweep of the persistent grid of the wave-per-tile attention launches (DR4SR_ATTN_WAVE_GRID): us per step of the attention kinds, 2 layers
mkdir -p gpurun_out
run() { echo -n "grid=$1 B=$2: "; env DR4SR_ATTN_WAVE_GRID=$1 timeout 300 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --no-deterministic-leg --batch $2 --steps 60 --repeats 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), 'attn_fwd', k.get('attn_fwd'), 'attn_bwd', k.get('attn_bwd'))"; }
for B in ${SW_SIZES:-8192 4096}; do for g in ${SW_GRIDS:-8192 2048 1536 1024 768 512 256}; do run $g $B; done; done 2>&1 | tee gpurun_out/attn_wave_grid_sweep.txt
