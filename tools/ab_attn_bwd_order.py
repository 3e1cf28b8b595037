#!/usr/bin/env python3
"""Block order of the causal attention backward (A3V_ATTN_HEAD_GROUP, read per launch): interleaved rounds, identical outputs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"; BF = torch.bfloat16
groups = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,4,8,16".split(","))]
for (B, S, H, hd) in [(8, 1091, 32, 128), (8, 2182, 32, 128)]:
    q = torch.randn(B, S, H, hd, device=dev, dtype=BF); k = torch.randn(B, S, H, hd, device=dev, dtype=BF); v = torch.randn(B, S, H, hd, device=dev, dtype=BF)
    out = torch.randn(B, S, H, hd, device=dev, dtype=BF) * 0.1; dout = torch.randn(B, S, H, hd, device=dev, dtype=BF) * 0.1
    lse = torch.randn(B, H, S, device=dev).abs() + 5.0; D = torch.empty(B, S, H, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(256, dtype=torch.uint8, device=dev)
    ksb, ksh, vsb, vss, vsh = S * H * hd, hd, S * H * hd, H * hd, hd          # k, v token-major [B, S, H, hd]
    kk = k.permute(0, 2, 1, 3).contiguous()                                    # k as [B, H, S, hd] (cache layout): sb = H*S*hd, sh = S*hd
    f = lambda: ops.attention_bwd(q, kk, H * S * hd, S * hd, v, vsb, vss, vsh, out, dout, lse, D, dq, dk, dv, B, S, H, H, hd, True, ws)
    ref, times = None, {g: [] for g in groups}
    for r in range(5):
        for g in groups:
            os.environ["A3V_ATTN_HEAD_GROUP"] = str(g)
            __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
            f()
            cur = (dq.clone(), dk.clone(), dv.clone())
            if ref is None:
                ref = cur
            assert all(torch.equal(a, b) for a, b in zip(cur, ref)), g
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f()
            e1.record(); torch.cuda.synchronize()
            times[g].append(e0.elapsed_time(e1) / 3 * 1e3)
    print(json.dumps(dict(B=B, S=S, H=H, **{f"g{g}_us": round(sorted(t)[2], 1) for g, t in times.items()})), flush=True)
os.environ.pop("A3V_ATTN_HEAD_GROUP", None)
__import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
