#!/bin/bash
# Round-end evidence on one GPU box: full -m gpu suite, the default bench line, and the rocprofv3 kernel tables of the same
# bench command (whole bench, and forward+decode only).  usage (from the repo root on the box): tools/final_run.sh r01i
tag=${1:-r01x}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $out/pytest_gpu.txt
timeout 900 python bench.py 2> $out/bench.err | tail -1 > $out/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof1 /tmp/prof2
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o all -- python $root/bench.py --no-cpu-baseline 2> $out/prof.err | tail -1 > $out/bench_prof.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o fwd -- python $root/bench.py --no-cpu-baseline --no-train 2>> $out/prof.err | tail -1 > $out/bench_fwd.json
cd $root
python tools/rocprof_summary.py /tmp/prof1/all_results.db > $out/kernel_stats.txt
python tools/rocprof_summary.py /tmp/prof2/fwd_results.db > $out/kernel_stats_fwd_decode.txt
tail -3 $out/pytest_gpu.txt
cut -c1-400 $out/bench.json
