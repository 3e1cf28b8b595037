cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -q -x -k "attention or attn" 2>&1 | tail -2
for v in 0 1 0; do echo "A3V_ATTN_PP=$v"; A3V_ATTN_PP=$v timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu; done
