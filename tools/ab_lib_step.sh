#!/bin/bash
# usage: tools/ab_lib_step.sh OTHER_LIB.so [legs]  -- bench legs with another build of the library (A3V_LIB_PATH) against the in-tree one,
# alternating processes on one box
other=$1; legs=${2:-lora}
for rep in 1 2 3; do
  for v in other tree; do
    if [ $v = other ]; then export A3V_LIB_PATH=$other; else unset A3V_LIB_PATH; fi
    for leg in ${legs//,/ }; do
      python bench.py --legs $leg --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$leg', 'ms_per_step', d.get('ms_per_step'))"
    done
  done
done
