set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5
for n in 1 2 3; do
  echo "== NARROW=$n"; A3V_SKINNY_NARROW=$n timeout 300 python tools/lora_skinny_bench.py 2>&1 | grep "^nt" 
done > gpurun_out/r5/skinny.log 2>&1
cat gpurun_out/r5/skinny.log
timeout 300 python tools/lora_skinny_bench.py 2>&1 | grep "^tn" > gpurun_out/r5/tn.log; cat gpurun_out/r5/tn.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "stand_ins or attach" 2>&1 | tail -5
