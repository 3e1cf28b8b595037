#!/usr/bin/env python3
"""attn_prefill_w64_kernel built with -DW64_STAMP=<block> -DW64_STAMP_FINE: intervals between consecutive stamps of one wave (raw list)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
B, S, H, hd = 8, 2048, 32, 128
sp = S
q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
o = torch.empty_like(q); lse = torch.zeros(B * H * S + 2048, device=dev)
for _ in range(3):
    lse.zero_()
    ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, False)
torch.cuda.synchronize()
t = lse[B * H * S:].view(torch.int64).flatten()[:512].cpu().tolist()
t = [x for x in t if x != 0]
d = [t[i + 1] - t[i] for i in range(len(t) - 1)]
print(len(t)); print(d[:120])
