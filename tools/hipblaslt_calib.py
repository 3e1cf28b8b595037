#!/usr/bin/env python3
"""Calibration only (never on the product path): what the vendor library (hipBLASLt behind torch.matmul) reaches on this box for
the step's GEMM shapes, next to a3v_gemm_nt / _nn / _tn on the same operands -- says how much of the gap to the nominal MFMA peak
is schedule and how much is the part (clock / power under real data)."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from a3vlm_amd import ops  # noqa: E402

dev = "cuda"


def ev(fn, reps=8, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


rows = 8728
shapes = [(rows, 12288, 4096), (rows, 4096, 4096), (rows, 22016, 4096), (rows, 4096, 11008), (8192, 8192, 8192), (4616, 4096, 1024), (4616, 1024, 4096)]
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
t_end = __import__("time").perf_counter() + 0.5
while __import__("time").perf_counter() < t_end:
    x @ x
torch.cuda.synchronize()
print(f"{'shape':>24} {'kind':>4} {'a3v us':>9} {'a3v TF':>8} {'blaslt us':>10} {'blaslt TF':>9}")
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t1 = ev(lambda: ops.gemm_nt(a, w, o))
    t2 = ev(lambda: torch.matmul(a, w.t(), out=o))
    print(f"{str((M, N, K)):>24} {'nt':>4} {t1 * 1e6:9.1f} {fl / t1 / 1e12:8.1f} {t2 * 1e6:10.1f} {fl / t2 / 1e12:9.1f}", flush=True)
    if M == rows:
        # input gradient dX[M, K] = dY[M, N] @ W[N, K]; weight gradient dW[N, K] = dY^T @ X
        dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        t1 = ev(lambda: ops.gemm_nn(dy, w, dx))
        t2 = ev(lambda: torch.matmul(dy, w, out=dx))
        print(f"{str((M, K, N)):>24} {'nn':>4} {t1 * 1e6:9.1f} {fl / t1 / 1e12:8.1f} {t2 * 1e6:10.1f} {fl / t2 / 1e12:9.1f}", flush=True)
        gw = torch.empty(N, K, device=dev, dtype=torch.float32)
        gwb = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        t1 = ev(lambda: ops.gemm_tn(dy, a, gw, epilogue=ops.EPI_OUT_F32))
        t2 = ev(lambda: torch.matmul(dy.t(), a, out=gwb))
        print(f"{str((N, K, M)):>24} {'tn':>4} {t1 * 1e6:9.1f} {fl / t1 / 1e12:8.1f} {t2 * 1e6:10.1f} {fl / t2 / 1e12:9.1f}", flush=True)
        # the NT ring kernel on a pre-transposed weight (what a frozen-weight LoRA step can afford): dX = dY @ (W^T)^T
        wt = w.t().contiguous()
        t3 = ev(lambda: ops.gemm_nt(dy, wt, dx))
        print(f"{str((M, K, N)):>24} {'nt*':>4} {t3 * 1e6:9.1f} {fl / t3 / 1e12:8.1f}   (dgrad through the NT ring kernel on W^T)", flush=True)
        del dy, dx, gw, gwb, wt
    del a, w, o
