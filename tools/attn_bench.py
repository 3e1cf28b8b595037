#!/usr/bin/env python3
"""GPU micro-benchmark of a3v_attention on the bench step's shapes (LLM causal hd=128, ViT hd=64) + backward."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"
def run(B, S, H, hd, causal, bwd=False):
    sp = (S + 63) // 64 * 64
    q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, S, device=dev)
    st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
    f = lambda: ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, causal)
    if bwd:
        v = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
        do = torch.randn_like(q); dq = torch.empty_like(q)
        dk = torch.empty(B, H, S, hd, device=dev, dtype=torch.bfloat16); dv = torch.empty_like(dk)
        D = torch.empty(B, S, H, device=dev)
        ws = torch.empty(ops.attention_bwd_workspace_bytes(B, S, H, H, hd), dtype=torch.uint8, device=dev)
        f()
        f = lambda: ops.attention_bwd(q, k, H*sp*hd, sp*hd, v, S*H*hd, H*hd, hd, o, do, lse, D, dq, dk, dv, B, S, H, H, hd, causal, workspace=ws)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    fl = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0) * (2.5 if bwd else 1.0)
    print(json.dumps(dict(B=B, S=S, H=H, hd=hd, causal=causal, bwd=bwd, us=round(t*1e6, 1), tflops=round(fl/t/1e12, 1))), flush=True)
run(8, 1091, 32, 128, True)
run(8, 577, 16, 64, False)
run(8, 2048, 32, 128, True)
run(8, 1091, 32, 128, True, bwd=True)
