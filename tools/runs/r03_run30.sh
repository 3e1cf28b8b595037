cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in 1 0; do echo "A3V_ATTN_PP=$v"; A3V_ATTN_PP=$v timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu | head -3; done
