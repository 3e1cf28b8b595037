cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r17
timeout 600 python tools/ring_latew_ab.py 2>&1 | tail -16 | tee gpurun_out/r17/latew.txt
