#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  echo "=== variant: $v"
  make -C a3vlm_amd/csrc clean >/dev/null
  make -C a3vlm_amd/csrc -j8 EXTRA="$v" 2>&1 | grep -E " error" | head -3
  timeout 300 python tools/attn_bench.py 2>&1 | grep "^{"
done
make -C a3vlm_amd/csrc clean >/dev/null
