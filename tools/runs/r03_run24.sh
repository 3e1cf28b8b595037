cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r24
timeout 900 python tools/ring_latew_ab.py ring=5 latew=3 2>&1 | tail -16 | sed 's/latew/nocont/g' | tee gpurun_out/r24/cont_ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm" 2>&1 | tail -3
