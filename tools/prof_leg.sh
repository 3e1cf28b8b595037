#!/bin/bash
# rocprofv3 --kernel-trace --stats table OF THE STEP ALONE for one bench leg: tools/prof_leg.sh <out-name> <leg> [marker] [skip] [ENV=VAL ...]
name=$1; leg=$2; marker=${3:-embed_assemble}; skip=${4:-3}; shift 4
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe_$name
env "$@" timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pe_$name -o t -- python $root/bench.py --legs $leg --steps 6 --warmup 2 --no-roofline --no-cpu-baseline > /dev/null 2> $root/gpurun_out/prof/$name.err
python $root/tools/rocprof_summary.py /tmp/pe_$name/t_results.db --step-marker $marker --skip $skip > $root/gpurun_out/prof/kernel_stats_$name.txt
head -14 $root/gpurun_out/prof/kernel_stats_$name.txt | cut -c1-190
