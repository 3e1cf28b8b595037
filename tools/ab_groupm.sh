#!/bin/bash
# A/B of GROUP_M (tile rows per group of the tile walk) 8 vs 16 with the round-major XCD map: two builds of the library
for rep in 1 2; do
for v in g8 g16; do
  cp gpurun_tmp/lib_$v.so a3vlm_amd/liba3vlm_hip.so
  echo "== $v"
  python bench.py --legs forward,train --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['train']['ms_per_step'], 'fwd', d['forward']['ms_per_step'])"
done
done
