#!/bin/bash
# Same-box alternating A/B of bench legs under an A3V_* switch: tools/ab_env_legs.sh A3V_GEMM_RING_192 "lora,forward" 3
var=$1; legs=${2:-lora,forward}; n=${3:-3}
for i in $(seq 1 $n); do
  for v in 1 0; do
    env $var=$v python bench.py --legs $legs --steps 12 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'headline', d['ms_per_step'], 'forward', (d.get('forward') or {}).get('ms_per_step'), 'train', (d.get('train') or {}).get('ms_per_step'))"
  done
done
