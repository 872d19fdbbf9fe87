#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
run() { python bench.py --no-cpu-baseline --no-strong --no-throughput-mode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), {a:k[a] for a in ('post_fwd','post_mid','post_bwd','wgrad_fused')})"; }
for v in "DR4SR_X=0" "DR4SR_WT_SAVE_A=1" "DR4SR_X=0" "DR4SR_WT_SAVE_A=1"; do
  echo "== $v"
  echo -n "toys8192 "; env $v bash -c "$(declare -f run); run --batch 8192 --steps 100"
  echo -n "dense8192 "; env $v bash -c "$(declare -f run); run --batch 8192 --steps 40 --dense"
  echo -n "toys131072 "; env $v bash -c "$(declare -f run); run --batch 131072 --steps 20"
done
for o in 1 0 2 1 0 2; do echo -n "gru order $o: "; DR4SR_GRU_WAVE_ORDER=$o timeout 300 python bench.py --model gru4rec --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('gru4rec', d['ms_per_step'], d['value'])"; done
( time timeout 900 python -m pytest tests/test_gpu_cl.py tests/test_gpu_meta.py -q -k "data_parallel or fit_end_to_end" ) > $O/pytest_cl_meta2.txt 2>&1
tail -8 $O/pytest_cl_meta2.txt
