#!/usr/bin/env python3
"""Cycle stamps of one wave of attn_prefill_w64_kernel (library built with EXTRA=-DW64_STAMP=<block>): per step
[start, after phase 1, after vmcnt(0), after the barrier, after the DMA issue] then the end of the loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
B, S, H, hd = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 32, 128
causal = (sys.argv[2] == "1") if len(sys.argv) > 2 else False
sp = (S + 63) // 64 * 64
q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
o = torch.empty_like(q); lse = torch.zeros(B * H * S + 2048, device=dev)
for _ in range(3):
    lse.zero_()
    ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, causal)
torch.cuda.synchronize()
t = lse[B * H * S:].view(torch.int64).flatten()[:512].cpu().tolist()
t = [x for x in t if x != 0]
print("stamps", len(t))
print("prologue: start -> DMA issued + Q loaded", t[1] - t[0], " -> landed + barrier", t[2] - t[1])
body = t[2:]
rows = []
for i in range(0, len(body) - 1, 5):
    r = body[i:i + 6]
    if len(r) < 6: break
    rows.append([r[j + 1] - r[j] for j in range(5)])
print("per step: phase1, wait vmcnt, barrier, dma issue, phase2")
for i, r in enumerate(rows): print(i, r)
if rows:
    n = len(rows)
    print("mean", [round(sum(r[j] for r in rows[1:-1]) / max(1, n - 2)) for j in range(5)], "sum", round(sum(sum(r) for r in rows[1:-1]) / max(1, n - 2)))
print("total", t[-1] - t[0])
