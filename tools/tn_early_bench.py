#!/usr/bin/env python3
"""TN / NN GEMM on the 7B training shapes for one library build (A3V_LIB_PATH): used to compare -DTN_EARLY=n builds of gemm_tn_bf16_pp_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
DEV = "cuda:0"
def t_us(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
T = 8728
tot_tn = tot_nn = 0.0
for name, N, K in [("qkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]:
    dy = (torch.randn(T, N, device=DEV) * 0.1).bfloat16()
    x = (torch.randn(T, K, device=DEV) * 0.1).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    g = torch.zeros(N, K, device=DEV)
    dx = torch.empty(T, K, device=DEV, dtype=torch.bfloat16)
    ref = dy.float().t()[:64] @ x.float()
    ops.gemm_tn(dy, x, g, epilogue=ops.EPI_OUT_F32)
    e1 = float((g[:64] - ref).abs().max() / ref.abs().max())
    ops.gemm_nn(dy, w, dx)
    r2 = dy[:64].float() @ w.float()
    e2 = float((dx[:64].float() - r2).abs().max() / r2.abs().max())
    a = min(t_us(lambda: ops.gemm_tn(dy, x, g, epilogue=ops.EPI_OUT_F32)) for _ in range(3))
    b = min(t_us(lambda: ops.gemm_nn(dy, w, dx)) for _ in range(3))
    fl = 2.0 * T * N * K
    tot_tn += a; tot_nn += b
    print(f"{name:4s} tn {a:7.1f} us ({fl / a / 1e6:6.1f} TF, err {e1:.1e})   nn {b:7.1f} us ({fl / b / 1e6:6.1f} TF, err {e2:.1e})", flush=True)
print(f"sum tn {tot_tn:.1f} us  nn {tot_nn:.1f} us")
