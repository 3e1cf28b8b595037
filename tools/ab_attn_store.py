#!/usr/bin/env python3
"""Output stores of the prefill attention: whole rows through an LDS patch (default) against per-lane row stores
(A3V_ATTN_STAGED_O=0, read per launch).  Interleaved rounds, identical outputs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
for (B, S, H, hd, causal) in [(8, 1091, 32, 128, True), (8, 2182, 32, 128, True), (40, 577, 16, 64, False)]:
    sp = (S + 63) // 64 * 64
    q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
    st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
    o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev)
    ref, times = None, {"1": [], "0": []}
    for r in range(5):
        for v in ("1", "0"):
            os.environ["A3V_ATTN_STAGED_O"] = v
            __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
            f = lambda: ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, causal)
            o.zero_(); f()
            if ref is None:
                ref = o.clone()
            assert torch.equal(o, ref), v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    print(json.dumps(dict(B=B, S=S, H=H, hd=hd, staged_us=round(sorted(times["1"])[2], 1), per_lane_us=round(sorted(times["0"])[2], 1))), flush=True)
os.environ.pop("A3V_ATTN_STAGED_O", None)
__import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
