#!/usr/bin/env python3
"""What the rows beyond whole tile rounds cost: a3v_gemm_nt at M = 8728 (34 full tile rows + 24 rows) against M = 8192 (exactly two rounds at
N = 4096) and the tail rows alone, N = 4096 shapes of the 7B step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
def ev(fn, reps=8):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for _ in range(30): x @ x
for (N, K) in [(4096, 4096), (4096, 11008), (4096, 12352), (4096, 22080), (12288, 4096), (22016, 4096), (11008, 4160)]:
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    row = []
    for M in (8192, 8728, 8704, 8960, 536):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = min(ev(lambda: ops.gemm_nt(a, w, o)) for _ in range(3))
        row.append(f"M={M}: {t:7.1f} us {2.0*M*N*K/t/1e6:7.1f} TF")
    print(f"N={N:6d} K={K:6d}  " + "   ".join(row), flush=True)
