#!/usr/bin/env python3
"""Run ONE gemm configuration a few times (for rocprofv3 --pmc passes).
usage: gemm_one.py M N K cfg(auto|t128|t256|pp|ring|ring_direct) [reps]   pp = two-stage ping-pong kernel, ring = its 160-KiB ring form"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops, lib
M, N, K = map(int, sys.argv[1:4])
flag = {"auto": 0, "t128": lib.EPI_TILE_128, "t256": lib.EPI_TILE_256, "pp": lib.EPI_TILE_256PP | (7 << 24), "ring": lib.EPI_TILE_256PP | (5 << 24),
        "ring_direct": lib.EPI_TILE_256PP | (11 << 24)}[sys.argv[4]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm_nt(a, w, o, epilogue=flag)
torch.cuda.synchronize()
