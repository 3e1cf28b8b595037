#!/usr/bin/env python3
"""Cycle stamps of one prefill-attention block (build with EXTRA='-DAP_STAMP=<virtual block>'): per KV tile the s_memtime
deltas of [wait+barrier | QK^T | softmax | PV | end barrier].  usage: attn_stamps.py [S]"""
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
    ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, True)
torch.cuda.synchronize()
t = lse.view(-1)[:64 * 8 * 2].view(torch.int64).view(64, 8).cpu()
n = int((t[:, 0] != 0).sum())
print("tiles", n, "(s_memtime ticks are 100 MHz: 1 tick = 10 ns)")
for i in range(n):
    r = t[i]
    nxt = t[i + 1, 0] if i + 1 < n else r[4]
    print(f"tile {i:2d}: dma issue {int(r[5]-r[0]):5d}  vmcnt wait {int(r[6]-r[5]):5d}  barrier {int(r[1]-r[6]):5d}  qk {int(r[2]-r[1]):5d}  softmax {int(r[3]-r[2]):5d}  pv {int(r[4]-r[3]):5d}  endbar {int(nxt-r[4]):5d}   total {int(nxt-r[0]):5d}")
