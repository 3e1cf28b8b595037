#!/usr/bin/env python3
"""Decode attention (Sq = 1) at the bench geometry with KV caches rotating through > 256 MB (true HBM reads)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
B, H, hd, Sk, smax = 8, 32, 128, int(os.environ.get("SK", 1110)), 2048
L = 6
ks = [torch.randn(B, H, smax, hd, device=dev, dtype=torch.bfloat16) for _ in range(L)]
vs = [torch.randn(B, H, hd, smax, device=dev, dtype=torch.bfloat16) for _ in range(L)]
q = torch.randn(B, 1, 3 * H * hd, device=dev, dtype=torch.bfloat16)
o = torch.empty(B, 1, H * hd, device=dev, dtype=torch.bfloat16)
scratch = torch.empty(2 * ops.attention_scratch_floats(B, H, hd, smax + 64), dtype=torch.float32, device=dev)
ldq = 3 * H * hd
st = (ldq, ldq, hd, H * smax * hd, smax * hd, hd, H * hd * smax, hd * smax, smax, H * hd, H * hd, hd)
def f(i): ops.attention(q, ks[i % L], vs[i % L], o, B, 1, Sk, H, H, hd, st, False, scratch)
for i in range(6): f(i)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 60
for i in range(n): f(i)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n * 1e-3
by = 2.0 * B * H * Sk * hd * 2
print(json.dumps(dict(Sk=Sk, want=os.environ.get("A3V_DECODE_WANT"), us=round(t * 1e6, 1), tbs=round(by / t / 1e12, 2))))
