#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for sk in ""; do
  touch a3vlm_amd/csrc/a3v_attn.hip
  make -C a3vlm_amd/csrc EXTRA="-DAPP_STAMP=8 $sk" 2>&1 | grep -E " error" | head -3
  echo "=== stamps, extra='$sk'"
  timeout 120 python tools/attn_persist_stamps.py 1091 2>&1 | tail -12
  timeout 120 python tools/attn_persist_stamps.py 2048 2>&1 | tail -12
done > gpurun_out/r06b_persist_stamps.txt 2>&1
cat gpurun_out/r06b_persist_stamps.txt
