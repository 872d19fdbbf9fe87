#!/bin/bash
# round 5, first GPU call: the two-bucket data-parallel step (tests), the bench line with the 1-rank RCCL leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_dp.py -q -x --durations=20 > gpurun_out/r5_test_dp.txt 2>&1; echo "test_gpu_dp rc=$?"
tail -5 gpurun_out/r5_test_dp.txt
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_r3_paths.py -q -k "data_parallel or bench or rccl or metamodel_outer" > gpurun_out/r5_test_dp_old.txt 2>&1; echo "old dp tests rc=$?"
tail -5 gpurun_out/r5_test_dp_old.txt
timeout 900 python bench.py --steps 100 --warmup 20 > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    j = json.loads([l for l in open("gpurun_out/r5_bench_a.json") if l.startswith("{")][-1])
    print("value", j["value"], "ms", j["ms_per_step"], "err", j.get("dp_1rank_rccl_error"))
    for s in j.get("strong", []):
        print(s["global_batch"], s["ms_per_step"], json.dumps(s.get("dp_1rank_rccl")))
except Exception as e:
    print("parse failed", e)
PY
tail -3 gpurun_out/r5_bench_a.err
