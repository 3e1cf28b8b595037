cd ${GRAFT_REPO_ROOT:-/root/repo}
fail=0
for i in $(seq 1 12); do
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "runs_on_across or qkv_rope or ragged_rows or swiglu_backward or fast_epilogue" 2>&1 | tail -1 | grep -q "passed" || fail=$((fail+1))
done
echo "stress failures: $fail of 12"
timeout 900 python -m pytest tests/test_gpu_lora.py tests/test_gpu_train.py -q 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_lora.py tests/test_gpu_train.py -q 2>&1 | tail -1
