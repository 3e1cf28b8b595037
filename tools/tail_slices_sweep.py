#!/usr/bin/env python3
"""The hybrid dispatch's tail (M = 8728: 536 rows beyond the two whole rounds at N = 4096) by its split-K slice count
(A3V_GEMM_TAIL_SLICES; 0 = the rule: CUs / tail tiles = 5): whole a3v_gemm_nt call at M = 8728 minus the call at M = 8192, us."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import lib, ops
dev = "cuda"
def ev(fn, reps=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for (N, K) in [(4096, 4096), (4096, 4160), (4096, 11072), (4096, 12352), (4096, 22080)]:
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    a = torch.randn(8728, K, device=dev, dtype=torch.bfloat16)
    o = torch.empty(8728, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(8728, N, device=dev, dtype=torch.bfloat16)
    base = min(ev(lambda: ops.gemm_nt(a[:8192], w, o[:8192], residual=res[:8192])) for _ in range(3))
    row = [f"M=8192: {base:6.1f}"]
    for S in (0, 3, 4, 5, 6, 8, 10):
        with lib.env(A3V_GEMM_TAIL_SLICES=S):
            t = min(ev(lambda: ops.gemm_nt(a, w, o, residual=res)) for _ in range(3))
        row.append(f"S={S}: {t:6.1f} (+{t - base:5.1f})")
    print(f"N={N} K={K:6d}  " + "  ".join(row), flush=True)
