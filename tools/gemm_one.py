#!/usr/bin/env python3
"""Run ONE gemm configuration a few times (for rocprofv3 --pmc passes).
usage: gemm_one.py M N K cfg(auto|t128|t256|pp|ring|ring_direct|t192|fp8|tn|nn) [reps]   pp = two-stage ping-pong kernel, ring = its 160-KiB ring form;
tn: C[M,N] (fp32) = At[K,M]^T Wt[K,N] (weight gradients), nn: C[M,N] = A[M,K] W[K,N] (input gradients)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops, lib
M, N, K = map(int, sys.argv[1:4])
cfg = sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
if cfg in ("tn", "nn"):
    if cfg == "tn":
        a = torch.randn(K, M, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(K, N, device="cuda", dtype=torch.bfloat16) * 0.02
        o = torch.empty(M, N, device="cuda", dtype=torch.float32)
        f = lambda: ops.gemm_tn(a, w, o, epilogue=ops.EPI_OUT_F32)  # noqa: E731
    else:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(K, N, device="cuda", dtype=torch.bfloat16) * 0.02
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        f = lambda: ops.gemm_nn(a, w, o)  # noqa: E731
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    sys.exit(0)
if cfg == "fp8":        # W8A8 product on the ring kernel (round 5)
    aq = torch.randint(0, 120, (M, K), device="cuda", dtype=torch.uint8)
    wq = torch.randint(0, 120, (N, K), device="cuda", dtype=torch.uint8)
    sa, sw = torch.rand(M, device="cuda") * 1e-2 + 1e-3, torch.rand(N, device="cuda") * 1e-2 + 1e-3
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(reps):
        ops.gemm_nt_fp8(aq, sa, wq, sw, o)
    torch.cuda.synchronize()
    sys.exit(0)
flag = {"auto": 0, "t192": lib.EPI_TILE_192PP, "t128": lib.EPI_TILE_128, "t256": lib.EPI_TILE_256, "pp": lib.EPI_TILE_256PP | (7 << 24), "ring": lib.EPI_TILE_256PP | (5 << 24),
        "ring_direct": lib.EPI_TILE_256PP | (11 << 24)}[sys.argv[4]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm_nt(a, w, o, epilogue=flag)
torch.cuda.synchronize()
