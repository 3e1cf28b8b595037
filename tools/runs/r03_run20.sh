cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r20
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -q -x -k "attention or attn" 2>&1 | tail -5
timeout 600 python tools/attn_bwd_stamps.py 2>&1 | grep "mean\|stamped" | tee gpurun_out/r20/bwd_stamps.txt
timeout 300 python tools/attn_bench.py 2>&1 | tail -4 | tee gpurun_out/r20/attn_bench.txt
