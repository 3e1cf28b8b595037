#!/bin/bash
# Kernel tables (rocprofv3 --kernel-trace --stats) of the bench's own commands at the current commit: the full fine-tune leg and the
# forward + decode legs, each next to the JSON line of that profiled run.  usage (repo root on the GPU box): tools/round_evidence.sh r02h
tag=${1:-r02x}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe1 /tmp/pe2
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pe1 -o t -- python $root/bench.py --legs train --steps 5 --warmup 2 2> $out/prof_train.err | tail -1 > $out/bench_train_profiled.json
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pe2 -o f -- python $root/bench.py --legs forward,decode --no-train 2> $out/prof_fwd.err | tail -1 > $out/bench_forward_decode_profiled.json
cd $root
python tools/rocprof_summary.py /tmp/pe1/t_results.db --step-marker embed_assemble --skip 2 > $out/kernel_stats_train_step.txt
python tools/rocprof_summary.py /tmp/pe2/f_results.db > $out/kernel_stats_forward_decode.txt
head -12 $out/kernel_stats_train_step.txt | cut -c1-170
cut -c1-300 $out/bench_train_profiled.json
