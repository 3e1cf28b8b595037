cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r19
timeout 600 python tools/attn_bwd_stamps.py 2>&1 | tail -40 | tee gpurun_out/r19/bwd_stamps.txt
timeout 300 python tools/attn_bench.py 2>&1 | tail -4 | tee gpurun_out/r19/attn_bench.txt
