set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_13b_shapes.py tests/test_gpu_lora.py tests/test_gpu_preprocess.py tests/test_gpu_eval_entry.py tests/test_gpu_trainer.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2/tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r2/tests.log
B="python bench.py --legs lora --no-roofline --no-cpu-baseline --steps 8 --warmup 3"
for cfg in "default" "A3V_LORA_NT_DGRAD=0" "A3V_LORA_NT_DGRAD=all" "A3V_ADAMW_MULTI=0" "default"; do
  if [ "$cfg" = "default" ]; then env $B > gpurun_out/r2/lora_$cfg.json 2> gpurun_out/r2/err.log; else env $cfg $B > gpurun_out/r2/lora_$cfg.json 2> gpurun_out/r2/err.log; fi
  echo "$cfg: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r2/lora_$cfg.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['train_lora'].get('loss'))")"
done
