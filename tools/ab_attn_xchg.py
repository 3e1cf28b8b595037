#!/usr/bin/env python3
"""A/B of the row-maximum exchange of the prefill kernel inside one process: v_permlane32_swap (A3V_ATTN_LAZY=1, product) against the
__shfl_xor / ds_bpermute form (A3V_ATTN_LAZY=3).  Outputs must be bit-identical; interleaved timing rounds on random data."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
from a3vlm_amd import lib as L
dev = "cuda"
def setv(v):
    os.environ["A3V_ATTN_LAZY"] = v
    L.load().a3v_reload_env()
for (B, S, H, hd, causal) in [(8, 1091, 32, 128, True), (8, 2182, 32, 128, True), (8, 1967, 32, 128, True), (8, 1091, 40, 128, True), (40, 577, 16, 64, False)]:
    sp = (S + 63) // 64 * 64
    q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
    st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
    outs, lses, times = {}, {}, {"1": [], "3": []}
    for v in ("1", "3"):
        setv(v)
        outs[v] = torch.empty_like(q); lses[v] = torch.empty(B, H, S, device=dev)
        ops.attention_lse(q, k, vt, outs[v], lses[v], B, S, S, H, H, hd, st, causal)
    o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev)
    for r in range(7):
        for v in ("1", "3"):
            setv(v)
            f = lambda: ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, causal)
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5 * 1e-3)
    fl = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    print(json.dumps(dict(B=B, S=S, H=H, hd=hd, causal=causal, permlane_us=round(med["1"] * 1e6, 1), bpermute_us=round(med["3"] * 1e6, 1),
                          permlane_tf=round(fl / med["1"] / 1e12, 1), speedup=round(med["3"] / med["1"], 3),
                          bit_identical=bool(torch.equal(outs["1"], outs["3"]) and torch.equal(lses["1"], lses["3"])))), flush=True)
