#!/usr/bin/env python3
"""rmsnorm backward on the bf16 residual stream at the bench geometry (x, dy, dh bf16; dw fp32 through per-block partials)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
for rows, dim in ((8728, 4096), (8728, 5120)):
    x = torch.randn(rows, dim, device="cuda").bfloat16()
    w = torch.ones(dim, device="cuda")
    dy = torch.randn(rows, dim, device="cuda", dtype=torch.bfloat16)
    dh = torch.zeros(rows, dim, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(dim, device="cuda")
    f = lambda: ops.rmsnorm_bwd(x, w, dy, dh, dw, 1e-5)
    for _ in range(3): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    t = sorted(ts)[2]
    print(f"rows {rows} dim {dim}: {t*1e3:.1f} us  ({rows*dim*2*4/t/1e9:.2f} TB/s of 4 bf16 streams)")
