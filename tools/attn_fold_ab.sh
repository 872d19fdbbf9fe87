#!/bin/bash
# A/B on one box (EXPERIMENTS build: make -C dr4sr_amd/csrc EXPERIMENTS=1): the wave attention's forward folded into the wave-tile forward kernels
# (DR4SR_ATTN_FOLD=1, linear_wave.hip wt_attn_ctx) against the launch of its own (default).  profiles/round6_attn_fold_ab.txt was taken when the
# fold was the default of the tree and DR4SR_ATTN_NOFOLD the switch — same two forms.
mkdir -p gpurun_out
export DR4SR_LIB_PATH=$PWD/dr4sr_amd/csrc/libdr4sr_hip_exp.so
run() { echo -n "$1 B=$2: "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --no-deterministic-leg --batch $2 --steps 100 --repeats 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), {x: k.get(x) for x in ('attn_fwd','post_fwd','post_mid','attn_bwd')})"; }
for B in ${AB_SIZES:-8192 4096 2048 32768}; do run DR4SR_ATTN_FOLD=1 $B; run X=1 $B; done 2>&1 | tee gpurun_out/attn_fold_ab.txt
