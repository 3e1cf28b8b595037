#!/bin/bash
# timing experiments on the prefill attention kernel: rebuild a3v_attn.hip with -D switches that remove one ingredient
# (results are wrong by construction; only the time is read).  usage: tools/variant_attn.sh "" -DAP_NO_EXP -DAP_NO_QK -DAP_NO_PV -DAP_NO_SM -DAP_NO_DMA -DAP_NO_BAR ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  echo "=== variant: '$v'"
  touch a3vlm_amd/csrc/a3v_attn.hip
  make -C a3vlm_amd/csrc EXTRA="$v" 2>&1 | grep -E " error" | head -3
  timeout 300 python tools/attn_bench.py 2>&1 | grep "^{" | head -3
done
