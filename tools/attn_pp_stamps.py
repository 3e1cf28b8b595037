#!/usr/bin/env python3
"""Cycle stamps of one block of the ping-pong prefill kernel (build with EXTRA='-DPP_STAMP=0'): wave 0 (group 0) and wave 4 (group 1);
codes 1 / 2 = before / after a barrier, 3 = QK issued, 4 = PV issued, 5 = softmax done, 6 = DMA issue + counted wait done."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1091
B, H, hd = 8, 32, 128
sp = (S + 63) // 64 * 64
q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
o = torch.empty_like(q)
lse = torch.zeros(B, H, S, device=dev)
st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
for _ in range(3):
    lse.zero_()
    ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, True)
torch.cuda.synchronize()
t = lse.view(-1)[:2048].view(torch.int64).view(2, 256, 2).cpu()
names = {1: "->bar", 2: "bar", 3: "qk", 4: "pv", 5: "softmax", 6: "io+wait"}
for g in range(2):
    r = t[g]
    n = int((r[:, 0] != 0).sum())
    print(f"group {g}: {n} stamps")
    line = []
    for i in range(1, min(n, 120)):
        line.append(f"{names.get(int(r[i,0]), '?')}:{int(r[i,1]-r[i-1,1])}")
    print("  " + " ".join(line))
