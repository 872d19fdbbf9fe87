#!/bin/bash
# A/B of two builds of the library on one box, alternating: the tree's libdr4sr_hip.so against $1 (default tools/ab_libs/libdr4sr_hip_prev.so)
mkdir -p gpurun_out
OTHER=${1:-$PWD/tools/ab_libs/libdr4sr_hip_prev.so}
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-strong --no-throughput-mode --no-deterministic-leg --no-dp-leg $2 --repeats 7 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), end=' ')"; }
for cfg in "--batch 256 --steps 300" "--batch 8192 --steps 100" "--batch 256 --steps 300 --embed-dim 128" "--model fmlp --steps 200" "--model gru4rec --steps 100"; do
  for rep in 1 2 3; do echo -n "[$cfg] tree: "; run X=1 "$cfg"; echo -n "| other: "; run DR4SR_LIB_PATH=$OTHER "$cfg"; echo; done
done 2>&1 | tee gpurun_out/lib_ab.txt
