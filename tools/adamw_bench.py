#!/usr/bin/env python3
"""AdamW kernel (a3v_adamw_scaled with the bf16 image) on 7B-sized tensors, cache policy variants (A3V_ADAMW_NT: bit 0 = non-temporal
loads, bit 1 = non-temporal stores; read per launch).  Tensors rotate through > 256 MB so that nothing stays in the Infinity Cache."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import lib
dev = "cuda"
n = 11008 * 4096
sets = []
for i in range(3):
    sets.append([torch.randn(n, device=dev) for _ in range(4)] + [torch.empty(n, device=dev, dtype=torch.bfloat16)])
coef = torch.ones(1, device=dev)
L = lib.load()
def f(i):
    p, g, m, v, img = sets[i % 3]
    v.abs_() if i < 0 else None
    rc = L.a3v_adamw_scaled(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-5, 0.9, 0.95, 1e-8, 0.02, 10, img.data_ptr(), coef.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0
for s_ in sets:
    s_[3].abs_()
res = {}
for r in range(4):
    for nt in ("0", "1", "2", "3"):
        os.environ["A3V_ADAMW_NT"] = nt
        __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
        for i in range(3): f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(9): f(i)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(nt, []).append(e0.elapsed_time(e1) / 9 * 1e3)
print(json.dumps({f"nt{k}": {"us": round(sorted(v)[1], 1), "tbs": round(30.0 * n / sorted(v)[1] / 1e6, 2)} for k, v in res.items()}))
