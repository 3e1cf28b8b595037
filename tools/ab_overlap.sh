#!/bin/bash
# A/B of the overlapped optimizer step on the train leg of bench.py (same box, alternating)
for rep in 1 2; do
for v in 1 0; do
  echo "== A3V_ADAMW_OVERLAP=$v"
  A3V_ADAMW_OVERLAP=$v python bench.py --legs train --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
done
