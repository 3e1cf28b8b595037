#!/usr/bin/env python3
"""Block timeline of the one-pass dK + dV kernel (stamp build: make EXTRA=-DAB_STAMP=0, or A3V_LIB_PATH=<stamp build>): per block
s_memrealtime (100 MHz) at entry / loop start / loop end / exit + the CU it ran on -> block phases and the gap between consecutive blocks of a CU."""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import lib, ops
dev = "cuda"
B, S, H, hd = 8, 1091, 32, 128
sp = (S + 63) // 64 * 64
q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
v = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev)
st = (S * H * hd, H * hd, hd, H * sp * hd, sp * hd, hd, H * hd * sp, hd * sp, sp, S * H * hd, H * hd, hd)
ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, True)
do = torch.randn_like(q); dq = torch.empty_like(q)
dk = torch.empty(B, H, S, hd, device=dev, dtype=torch.bfloat16); dv = torch.empty_like(dk)
D = torch.empty(B, S, H, device=dev); ws = torch.empty(256, dtype=torch.uint8, device=dev)
nblk = ((S + 127) // 128) * H * B
stamps = torch.zeros(4096 + 8 * nblk, dtype=torch.int64, device=dev)
L = lib.load()
L.a3v_debug_set_bwd_stamps.argtypes = [ctypes.c_void_p]; L.a3v_debug_set_bwd_stamps.restype = None
f = lambda: ops.attention_bwd(q, k, H * sp * hd, sp * hd, v, S * H * hd, H * hd, hd, o, do, lse, D, dq, dk, dv, B, S, H, H, hd, True, workspace=ws)
for _ in range(3): f()
torch.cuda.synchronize()
L.a3v_debug_set_bwd_stamps(ctypes.c_void_p(stamps.data_ptr())); f(); torch.cuda.synchronize(); L.a3v_debug_set_bwd_stamps(ctypes.c_void_p(0))
t = stamps[4096:].view(nblk, 8).cpu()
t0 = int(t[:, 0].min())
pro = (t[:, 1] - t[:, 0]).float() / 100; loop = (t[:, 2] - t[:, 1]).float() / 100; epi = (t[:, 3] - t[:, 2]).float() / 100
print(f"blocks {nblk}; kernel span {(int(t[:, 3].max()) - t0) / 100:.1f} us; first->last start {(int(t[:, 0].max()) - t0) / 100:.1f} us")
setup = (t[:, 6] - t[:, 0]).float() / 100; p0 = (t[:, 7] - t[:, 6]).float() / 100; rest = (t[:, 1] - t[:, 7]).float() / 100
print(f"prologue parts (us): index / descriptor set-up mean {setup.mean():.2f} | pair-0 DMA issue {p0.mean():.2f} | K, V loads + AGPR init + wait + pair 1 + A(0) {rest.mean():.2f}")
print(f"per block (us): prologue mean {pro.mean():.2f} max {pro.max():.2f} | loop mean {loop.mean():.2f} max {loop.max():.2f} | epilogue mean {epi.mean():.2f} max {epi.max():.2f}")
cus = collections.defaultdict(list)
for i in range(nblk):
    cus[(int(t[i, 5]) & 0xf, (int(t[i, 4]) >> 8) & 0xff)].append((int(t[i, 0]), int(t[i, 3]), i))
gaps, busy = [], []
for key, lst in cus.items():
    lst.sort()
    busy.append(sum(e - s for s, e, _ in lst) / 100)
    for (s0, e0, _), (s1, e1, _) in zip(lst, lst[1:]): gaps.append((s1 - e0) / 100)
gaps = torch.tensor(gaps)
print(f"CUs seen {len(cus)}; blocks per CU min {min(len(v) for v in cus.values())} max {max(len(v) for v in cus.values())}; busy per CU mean {sum(busy) / len(busy):.1f} us max {max(busy):.1f}")
print(f"gap between consecutive blocks of a CU (us): mean {gaps.mean():.2f} median {gaps.median():.2f} max {gaps.max():.2f} min {gaps.min():.2f}")
key0 = sorted(cus)[0]
print("one CU:", [(round((s - t0) / 100, 1), round((e - t0) / 100, 1)) for s, e, _ in cus[key0]])
