#!/usr/bin/env python3
"""Cost of the fused qkv / RoPE / KV-cache epilogue: a3v_gemm_qkv_rope against the plain GEMM of the same shape (7B: 8 x 1091
tokens, 4096 -> 12288), with and without the token-major v copy the training forward asks for."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops

dev = "cuda"
BF = torch.bfloat16
B, S, H, hd, K = 8, 1091, 32, 128, 4096
Smax = 2048
x = torch.randn(B * S, K, device=dev, dtype=BF)
w = torch.randn(3 * H * hd, K, device=dev, dtype=BF) * 0.02
qkv = torch.zeros(B * S, 3 * H * hd, device=dev, dtype=BF)
kc = torch.zeros(B, H, Smax, hd, device=dev, dtype=BF)
vt = torch.zeros(B, H, hd, Smax, device=dev, dtype=BF)
vr = torch.zeros(B * S, H * hd, device=dev, dtype=BF)
t = torch.arange(2 * Smax, device=dev, dtype=torch.float32)
fr = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
ang = torch.outer(t, fr)
cs = torch.stack([ang.cos(), ang.sin()], -1).contiguous()


def timed(fn, reps=8, rounds=9):
    ts = []
    for _ in range(rounds):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


fl = 2.0 * B * S * 3 * H * hd * K
import time
t0 = time.time()
while time.time() - t0 < 0.6:          # clock ramp: the first few hundred ms after an idle gap read ~15 % low
    ops.gemm_nt(x, w, qkv)
    torch.cuda.synchronize()
res = {}
for fast in ("1", "0"):
    os.environ["A3V_GEMM_FAST_EPI"] = fast
    __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
    res["plain_" + fast] = timed(lambda: ops.gemm_nt(x, w, qkv))
    res["rope_" + fast] = timed(lambda: ops.gemm_qkv_rope(x, w, qkv, kc, vt, cs, B, S, H, H, hd, 0, 0))
    res["rope_vrows_" + fast] = timed(lambda: ops.gemm_qkv_rope(x, w, qkv, kc, vt, cs, B, S, H, H, hd, 0, 0, v_rows=vr))
os.environ["A3V_GEMM_FAST_EPI"] = "1"
__import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
print(json.dumps({k: {"us": round(v * 1e3, 1), "tf": round(fl / v / 1e9, 1)} for k, v in res.items()}))
