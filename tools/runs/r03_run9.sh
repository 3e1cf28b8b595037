cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r9
timeout 600 python tools/ring_m32_ab.py 2>&1 | tail -12 | tee gpurun_out/r9/ring_m32_ab.txt
timeout 300 python -m pytest tests/test_gpu_dp.py -q -m gpu 2>&1 | tail -3
