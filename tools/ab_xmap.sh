#!/bin/bash
# A/B of the round-major XCD tile map of the ring kernel (A3V_GEMM_XMAP=1), alternating processes on one box
for rep in 1 2; do
for v in 0 1; do
  echo "== A3V_GEMM_XMAP=$v"
  A3V_GEMM_XMAP=$v W4_MODES=0,0,0 python tools/gemm_w4_ab.py 8192x8192x8192 8728x22016x4096 8728x4096x11008 8728x12288x4096 8728x4096x4096 2>&1 | grep ring
done
done
