mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_attn_wave.py -x -q -m gpu; echo rc=$?) > gpurun_out/attn_wave_tests.txt 2>&1
tail -5 gpurun_out/attn_wave_tests.txt
run() { echo -n "$1 B=$2 $3: "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --no-deterministic-leg --batch $2 --steps 100 --repeats 5 $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), round(d['value']), {x: k.get(x) for x in ('embqkv_fwd','attn_fwd','post_fwd','post_mid','attn_bwd','post_bwd','qkv_embed_bwd','wgrad_fused','adam')}, 'loss', round(d['final_loss'],5))"; }
for B in 8192 4096 2048 32768; do run X=1 $B; run DR4SR_ATTN_NOFOLD=1 $B; done 2>&1 | tee gpurun_out/attn_fold_ab.txt
