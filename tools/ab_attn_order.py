#!/usr/bin/env python3
"""Block order of the causal prefill attention: A3V_ATTN_HEAD_GROUP = heads per tile-rank-major group (1 = every head heavy-first on
its own; read per launch).  Interleaved rounds in one process, outputs must be bit-identical (same blocks, another order)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
groups = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8,16,32".split(","))]
for (B, S, H, hd) in [(8, 1091, 32, 128), (8, 2182, 32, 128), (8, 1967, 32, 128), (4, 1091, 40, 128)]:
    sp = (S + 63) // 64 * 64
    q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
    st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
    o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev)
    ref, times = None, {g: [] for g in groups}
    for r in range(5):
        for g in groups:
            os.environ["A3V_ATTN_HEAD_GROUP"] = str(g)
            __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
            f = lambda: ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, True)
            f()
            if ref is None:
                ref = o.clone()
            assert torch.equal(o, ref), g
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            times[g].append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 4.0 * B * H * S * S * hd * 0.5
    print(json.dumps(dict(B=B, S=S, H=H, **{f"g{g}_us": round(sorted(t)[2], 1) for g, t in times.items()},
                          **{f"g{g}_tf": round(fl / sorted(t)[2] / 1e6, 1) for g, t in times.items()})), flush=True)
os.environ.pop("A3V_ATTN_HEAD_GROUP", None)
__import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
