#!/bin/bash
# A/B on one box: the shipped library (experiment switches compiled out) against the EXPERIMENTS=1 build of the same source, alternating
mkdir -p gpurun_out
R=$(pwd)
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --no-deterministic-leg --no-dp-leg --batch $2 --steps ${3:-300} --repeats 7 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), end=' ')"; }
for B in 256 8192; do
  for rep in 1 2 3; do
    echo -n "B=$B shipped: "; run X=1 $B $([ $B = 8192 ] && echo 100); echo -n " | experiments build: "; run DR4SR_LIB_PATH=$R/dr4sr_amd/csrc/libdr4sr_hip_exp.so $B $([ $B = 8192 ] && echo 100); echo
  done
done 2>&1 | tee gpurun_out/prune_ab.txt
