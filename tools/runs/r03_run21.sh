cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r21
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -q -x -k "attention or attn" 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/attn_bench.py 2>&1 | tail -1; done | tee gpurun_out/r21/attn_bench.txt
