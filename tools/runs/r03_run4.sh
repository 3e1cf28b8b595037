set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_trainer.py tests/test_gpu_model.py -q -m gpu > gpurun_out/r4/tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r4/tests.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pl
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pl -o l -- python $R/bench.py --legs lora --no-roofline --no-cpu-baseline --steps 6 --warmup 2 2> $R/gpurun_out/r4/prof.err | tail -1 > $R/gpurun_out/r4/bench_lora_profiled.json
cd $R
python tools/rocprof_summary.py /tmp/pl/l_results.db --step-marker embed_assemble --skip 3 > gpurun_out/r4/kernel_stats_lora_step.txt
head -45 gpurun_out/r4/kernel_stats_lora_step.txt | cut -c1-175
