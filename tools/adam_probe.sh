#!/bin/bash
# B = 256 headline: which of the optimizer launch's two parts sets its time? (DR4SR_ADAM_PROBE: 1 = no prep chain, 2 = no sweep; timing only)
cd "$(dirname "$0")/.."
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-throughput-mode --no-strong --no-dp-leg --repeats 9"
for cfg in ${ADAM_CFGS:-"" "DR4SR_ADAM_PROBE=1" "DR4SR_ADAM_PROBE=2" "DR4SR_ADAM_BLOCKS=128" "DR4SR_ADAM_BLOCKS=416" "DR4SR_ADAM_BLOCKS=832"}; do
  r=$(env $cfg $B 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.5f ms  adam %s us' % (j['ms_per_step'], j.get('kernel_us_per_step',{}).get('adam')))")
  echo "ADAM_PROBE [$cfg] $r"
done
