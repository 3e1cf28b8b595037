#!/usr/bin/env python3
"""rmsnorm backward at the bench geometry, with and without the weight-gradient output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
rows, dim = 8728, 4096
x = torch.randn(rows, dim, device="cuda")
w = torch.ones(dim, device="cuda")
dy = torch.randn(rows, dim, device="cuda", dtype=torch.bfloat16)
dh = torch.zeros(rows, dim, device="cuda")
dw = torch.zeros(dim, device="cuda")
for name, d in (("with dw", dw), ("no dw", None)):
    for _ in range(3): ops.rmsnorm_bwd(x, w, dy, dh, d, 1e-5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.rmsnorm_bwd(x, w, dy, dh, d, 1e-5)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"{name}: {t*1e3:.1f} us  ({(rows*dim*(4+2+8))/t/1e9:.2f} TB/s)")
