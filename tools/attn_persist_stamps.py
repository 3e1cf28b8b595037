#!/usr/bin/env python3
"""100-MHz stamps of one persistent prefill-attention block over its units (build with EXTRA='-DAPP_STAMP=<block>'):
per unit [start -> tile-0 data landed | barrier | tile loop | seam barrier | DMA issue | O stores issued].  usage: attn_persist_stamps.py [S]"""
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
t = lse.view(-1)[:32 * 8 * 2].view(torch.int64).view(32, 8).cpu()
n = int((t[:, 0] != 0).sum())
print("units", n, "(10-ns ticks)")
for i in range(n):
    r = [int(x) for x in t[i]]
    nxt = int(t[i + 1, 0]) if i + 1 < n else r[6]
    print(f"unit {i:2d}: tiles {r[7]:3d}  wait tile0 {r[1]-r[0]:5d}  barrier {r[2]-r[1]:4d}  loop {r[3]-r[2]:6d}  seam-barrier {r[4]-r[3]:5d}  decode+dma {r[5]-r[4]:4d}  patch+stores {r[6]-r[5]:5d}  total {nxt-r[0]:6d}")

sp_ = lse.view(-1)[32 * 8 * 2:(32 * 8 + 512 * 8) * 2].view(torch.int64).view(512, 8).cpu()
t0 = int(sp_[:, 0].min())
import collections
print("block spans (10-ns ticks from the first block's start): start min/max", int(sp_[:, 0].min()) - t0, int(sp_[:, 0].max()) - t0,
      " end min/median/max", int(sp_[:, 1].min()) - t0, int(sp_[:, 1].median()) - t0, int(sp_[:, 1].max()) - t0)
for x in range(8):
    m = sp_[x::8]
    print(f"queue {x}: xcc ids {sorted(set(int(v) & 0xf for v in m[:, 4]))}  end min/max {int(m[:, 1].min()) - t0} {int(m[:, 1].max()) - t0}  tiles per block min/max {int(m[:, 3].min())} {int(m[:, 3].max())}"
          f"  units min/max {int(m[:, 2].min())} {int(m[:, 2].max())}  ticks per tile {float((m[:, 1] - m[:, 0]).sum()) / float(m[:, 3].sum()):.1f}")

m = sp_[0::8]
order = sorted(range(m.shape[0]), key=lambda i: int(m[i, 1]))
print("queue 0, blocks by end time: (end, last unit's tiles, last unit's start, units, tiles)")
print([(int(m[i, 1]) - t0, int(m[i, 5]), int(m[i, 6]) - t0, int(m[i, 2]), int(m[i, 3])) for i in order])
