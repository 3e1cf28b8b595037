#!/usr/bin/env python3
"""SwiGLU forward / backward passes of the training step at 7B size under A3V_STREAM_NT = 0 / 1 / 3 (read per launch); buffers rotate."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"; BF = torch.bfloat16
rows, F = 8728, 11008
sets = [(torch.randn(rows, 2 * F, device=dev, dtype=BF), torch.randn(rows, F, device=dev, dtype=BF), torch.empty(rows, 2 * F, device=dev, dtype=BF),
         torch.empty(rows, F, device=dev, dtype=BF)) for _ in range(2)]
res = {}
for r in range(3):
    for nt in ("0", "1", "3"):
        os.environ["A3V_STREAM_NT"] = nt
        __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
        for name, fn in (("fwd", lambda s: ops.swiglu_fwd(s[0], s[3], F, False)), ("bwd", lambda s: ops.swiglu_bwd(s[0], s[1], s[2], F, False))):
            for i in range(2): fn(sets[i % 2])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(8): fn(sets[i % 2])
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name + "_nt" + nt, []).append(e0.elapsed_time(e1) / 8 * 1e3)
print(json.dumps({k: round(sorted(v)[1], 1) for k, v in res.items()}))
