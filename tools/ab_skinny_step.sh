#!/bin/bash
# LoRA step with the adapter-sized NT products in their round-3 form (64-row tiles, two LDS stages, S = 4) against the current one
# (256-row blocks, three stages, one resident block per CU), alternating on one box.
for rep in 1 2 3; do
  for leg in legacy new; do
    if [ $leg = legacy ]; then export A3V_SKINNY_LEGACY=1 A3V_SKINNY_NARROW=3 A3V_SKINNY_STAGES=2; else unset A3V_SKINNY_LEGACY A3V_SKINNY_NARROW A3V_SKINNY_STAGES; fi
    python bench.py --legs lora --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$leg', 'ms_per_step', d.get('ms_per_step'), 'value', d.get('value'))"
  done
done
