#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest_gpu_full3.txt 2>&1
tail -16 $O/pytest_gpu_full3.txt
( time python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) 2>&1 | tail -4
( time python bench.py > $O/bench_default_final.json 2> $O/bench_default_final.err ) 2>&1 | tail -3
python -c "
import json; d=json.loads(open('$O/bench_default_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic_source'])
tm=d['throughput_mode']; print(tm['value'], tm['ms_per_step'], tm['roofline']['kernel'], tm['roofline']['bound'], tm['roofline']['frac'], tm['roofline'].get('traffic_source'))"
