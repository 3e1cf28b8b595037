cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r22
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -k "attention or attn or prefill or generate" 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/attn_bench.py 2>&1 | tail -4; done | tee gpurun_out/r22/attn_bench.txt
