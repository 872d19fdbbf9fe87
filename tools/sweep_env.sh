#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." [bench args]   — one bench line per value: ms/step and a kernel's us
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --no-cpu-baseline --no-throughput-mode "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$var=$v', round(d['ms_per_step'], 5), d.get('kernel_us_per_step'))"
done
