# round 3: where do the at-scale token-tile forms (wave tiles, k_wgrad_bf, owner job) and the attention lists start to win?
# per batch size: latency forms | at-scale tiles with per-sequence attention | at-scale tiles + attention lists
run() { python bench.py --no-cpu-baseline --no-strong --no-throughput-mode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), d['valid_tokens_last_step'], end='  ')"; }
for b in ${BATCHES:-256 384 512 768 1024 1536 2048}; do echo -n "B=$b $*: "; for cfg in "DR4SR_FORCE_SCALE=0" "DR4SR_FORCE_SCALE=1 DR4SR_FORCE_ATTN_SPLIT=0" "DR4SR_FORCE_SCALE=1"; do env $cfg bash -c "$(declare -f run); run --batch $b --steps 60 $*"; done; echo; done
