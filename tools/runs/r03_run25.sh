cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "qkv_rope" 2>&1 | tail -1; done
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_13b_shapes.py -q -x 2>&1 | tail -2
