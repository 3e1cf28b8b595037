cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r10
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_lora.py tests/test_gpu_two_image.py tests/test_gpu_trainer.py tests/test_gpu_dp.py -q -m gpu > gpurun_out/r10/tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r10/tests.log | tail -20
B="python bench.py --legs lora,train --no-roofline --no-cpu-baseline --steps 8 --warmup 3"
for cfg in "A3V_STREAM_FP32=0" "A3V_STREAM_FP32=1"; do
  env $cfg $B 2>gpurun_out/r10/err_$cfg.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'lora', d['train_lora']['ms_per_step'], d['train_lora']['loss'], d['train_lora']['hbm_gib'], 'train', d['train']['ms_per_step'], d['train']['loss'], d['train']['hbm_gib'])"
done | tee gpurun_out/r10/stream_ab.txt
