#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06d_pytest_gpu.txt 2>&1
tail -5 gpurun_out/r06d_pytest_gpu.txt
timeout 1500 python bench.py > gpurun_out/r06d_bench_7b.json 2> gpurun_out/r06d_bench_stderr.txt
echo "bench rc=$?"
tail -c 3000 gpurun_out/r06d_bench_7b.json
tail -5 gpurun_out/r06d_bench_stderr.txt
