#!/bin/bash
# round 6, lease a: persistent prefill attention -- parity + A/B, then the skeleton gate (no DMA, no arithmetic) on both launches
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python tools/ab_attn_persist.py > gpurun_out/r06a_ab_attn_persist.txt 2>&1
tail -30 gpurun_out/r06a_ab_attn_persist.txt
echo "=== skeleton gate"
SK="-DAP_NO_DMA -DAP_NO_SM -DAP_NO_QK -DAP_NO_PV"
touch a3vlm_amd/csrc/a3v_attn.hip
make -C a3vlm_amd/csrc EXTRA="$SK" 2>&1 | grep -E " error" | head -3
for v in 0 1; do
  echo "--- skeleton, A3V_ATTN_PERSIST=$v"
  A3V_ATTN_PERSIST=$v timeout 300 python tools/attn_bench.py 2>&1 | grep "^{" | head -3
done > gpurun_out/r06a_skeleton_gate.txt 2>&1
cat gpurun_out/r06a_skeleton_gate.txt
touch a3vlm_amd/csrc/a3v_attn.hip
make -C a3vlm_amd/csrc 2>&1 | grep -E " error" | head -3
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -3
