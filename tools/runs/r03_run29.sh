cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_13b_shapes.py -q -x -k "attention or attn or prefill or generate" 2>&1 | tail -4
for v in 1 0 1 0; do echo "A3V_ATTN_PP=$v"; A3V_ATTN_PP=$v timeout 300 python tools/attn_bench.py 2>&1 | head -3; done
