cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r13
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "tn_strip" 2>&1 | tail -15
timeout 300 python tools/lora_skinny_bench.py 2>&1 | grep "tn dA\|strip" | tee gpurun_out/r13/strip.txt
