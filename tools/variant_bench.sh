#!/bin/bash
# build the library with each EXTRA flag set on the GPU box and run tools/gemm_bench.py (PP_ONLY=1 -> only pp config)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  echo "=== variant: $v"
  make -C a3vlm_amd/csrc clean >/dev/null
  make -C a3vlm_amd/csrc -j8 EXTRA="$v" 2>&1 | grep -E "error" | head -3
  PP_ONLY=1 timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "^\{" 
done
make -C a3vlm_amd/csrc clean >/dev/null
