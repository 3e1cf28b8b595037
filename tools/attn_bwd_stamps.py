#!/usr/bin/env python3
"""Cycle stamps of one block of each attention-backward kernel (build with `make EXTRA=-DAB_STAMP=<blockIdx>`): per loop iteration the
s_memtime deltas (shader cycles) of [vmcnt wait | barrier | DMA issue | S / dP products + softmax algebra | accumulation products]."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import lib, ops  # noqa: E402

dev = "cuda"
B, S, H, hd = 8, 1091, 32, 128
sp = (S + 63) // 64 * 64
q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
v = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
o = torch.empty_like(q)
lse = torch.empty(B, H, S, device=dev)
st = (S * H * hd, H * hd, hd, H * sp * hd, sp * hd, hd, H * hd * sp, hd * sp, sp, S * H * hd, H * hd, hd)
ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, True)
do = torch.randn_like(q)
dq = torch.empty_like(q)
dk = torch.empty(B, H, S, hd, device=dev, dtype=torch.bfloat16)
dv = torch.empty_like(dk)
D = torch.empty(B, S, H, device=dev)
ws = torch.empty(256, dtype=torch.uint8, device=dev)
stamps = torch.zeros(3 * 64 * 8, dtype=torch.int64, device=dev)
L = lib.load()
L.a3v_debug_set_bwd_stamps.argtypes = [ctypes.c_void_p]
L.a3v_debug_set_bwd_stamps.restype = None
f = lambda: ops.attention_bwd(q, k, H * sp * hd, sp * hd, v, S * H * hd, H * hd, hd, o, do, lse, D, dq, dk, dv, B, S, H, H, hd, True, workspace=ws)
for _ in range(3):
    f()
torch.cuda.synchronize()
L.a3v_debug_set_bwd_stamps(ctypes.c_void_p(stamps.data_ptr()))
f()
torch.cuda.synchronize()
L.a3v_debug_set_bwd_stamps(ctypes.c_void_p(0))
t = stamps.view(3, 64, 8).cpu()
for kern, name in enumerate(("dQ", "dV", "dK")):
    r = t[kern]
    n = int((r[:, 0] != 0).sum())
    print(f"{name}: {n} iterations stamped")
    tot = [0] * 6
    for i in range(n):
        nxt = r[i + 1, 0] if i + 1 < n else r[i, 5]
        d = [int(r[i, 1] - r[i, 0]), int(r[i, 2] - r[i, 1]), int(r[i, 3] - r[i, 2]), int(r[i, 4] - r[i, 3]), int(r[i, 5] - r[i, 4]), int(nxt - r[i, 0])]
        for j in range(6):
            tot[j] += d[j]
        if i < 6 or i >= n - 2:
            extra = f"   [one-pass dK+dV: A||B of step 1 {int(r[i, 6] - r[i, 3])}, C {int(r[i, 4] - r[i, 6])}; step 2: {int(r[i, 7] - r[i, 4])}, {int(r[i, 5] - r[i, 7])}]" if int(r[i, 6]) else ""
            print(f"  it {i:2d}: vmcnt {d[0]:4d}  barrier {d[1]:4d}  issue {d[2]:4d}  s/dp+softmax {d[3]:4d}  acc-mfma {d[4]:4d}   total {d[5]:4d}" + extra)
    if n:
        print("  mean:  vmcnt %.0f  barrier %.0f  issue %.0f  s/dp+softmax %.0f  acc-mfma %.0f   total %.0f  (shader cycles)" % tuple(x / n for x in tot))

pe = t[2, 0:2].reshape(-1)[:9]
if int(pe[0]):
    names = ["issue 3 pairs", "K / V loads + AGPR init", "vmcnt(pair 0)", "barrier + A(0) + fetch", "(loop)", "AGPR reads + scale", "barrier", "patch writes", "row stores issued"]
    print("one-pass dK+dV, block prologue / epilogue (shader cycles):", ", ".join(f"{n} {int(pe[i + 1] - pe[i])}" for i, n in enumerate(names[:8])))
