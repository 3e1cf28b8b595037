#!/usr/bin/env python3
"""A/B of the prefill attention's V^T operand path inside one process (A3V_ATTN_PSWAP is read per launch): interleaved rounds,
random data, plus equality of the two variants' outputs up to bf16 rounding of P (same values, same order of the k sum)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
for (B, S, H, hd, causal) in [(8, 1091, 32, 128, True), (8, 2182, 32, 128, True), (8, 1967, 32, 128, True), (40, 577, 16, 64, False)]:
    sp = (S + 63) // 64 * 64
    q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
    st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
    outs, times = {}, {"1": [], "0": []}
    for v in ("1", "0"):
        os.environ["A3V_ATTN_PSWAP"] = v
        __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
        outs[v] = torch.empty_like(q)
        lse = torch.empty(B, H, S, device=dev)
        ops.attention_lse(q, k, vt, outs[v], lse, B, S, S, H, H, hd, st, causal)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, S, device=dev)
    for r in range(5):
        for v in ("1", "0"):
            os.environ["A3V_ATTN_PSWAP"] = v
            __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
            f = lambda: ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, causal)
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5 * 1e-3)
    fl = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    print(json.dumps(dict(B=B, S=S, H=H, hd=hd, causal=causal, pswap_us=round(med["1"] * 1e6, 1), half_reads_us=round(med["0"] * 1e6, 1),
                          pswap_tf=round(fl / med["1"] / 1e12, 1), half_tf=round(fl / med["0"] / 1e12, 1),
                          speedup=round(med["0"] / med["1"], 3), max_abs_diff=float((outs["1"].float() - outs["0"].float()).abs().max()))), flush=True)
