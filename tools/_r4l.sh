#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --no-cpu-baseline --no-strong --no-throughput-mode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernel_us_per_step']; print(round(d['ms_per_step'],4), round(d['value']), 'wgrad', k.get('wgrad_fused'))"; }
for v in "DR4SR_X=0" "DR4SR_WGRAD_BF16X3=1" "DR4SR_WGRAD_BF16X3=1,DR4SR_WGRAD_WIDE=1" "DR4SR_X=0" "DR4SR_WGRAD_BF16X3=1"; do
  echo "== $v"; echo -n "B256 "; env ${v//,/ } bash -c "$(declare -f run); run --steps 200"
  echo -n "d128 B256 "; env ${v//,/ } bash -c "$(declare -f run); run --steps 200 --embed-dim 128"
done
DR4SR_WGRAD_BF16X3=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trained.py -q -x -k "golden or full_size or trained or trajectory" 2>&1 | tail -3
