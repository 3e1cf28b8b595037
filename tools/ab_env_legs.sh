#!/bin/bash
# usage: tools/ab_env_legs.sh NAME v0 v1 [legs]   -- alternate NAME=v0 / NAME=v1 over bench legs on one box
name=$1; a=$2; b=$3; legs=${4:-forward,train}
for rep in 1 2; do
for v in $a $b; do
  echo "== $name=$v"
  env $name=$v python bench.py --legs $legs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d.get('train',{}).get('ms_per_step'), 'fwd', d.get('forward',{}).get('ms_per_step'), 'decode', d.get('decode_tok_s'))"
done
done
