#!/bin/bash
for rep in 1 2; do
for v in 1 5; do
  echo "== XMAP=$v (5 = + 16x16 super-tiles)"
  A3V_GEMM_XMAP=$v A3V_GEMM_XMAP_TN=$v python bench.py --legs forward,train --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['train']['ms_per_step'], 'fwd', d['forward']['ms_per_step'], d['roofline']['families'])"
done
done
