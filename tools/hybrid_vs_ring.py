#!/usr/bin/env python3
"""Dispatch check after the epilogue work: hybrid (ring on whole tile rounds + split-K 128x128 tail) against the persistent ring
kernel on all tiles, on the 7B forward shapes (residual epilogue where the model has one)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops, lib
dev = "cuda"; BF = torch.bfloat16
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8728
shapes = [("qkv", T, 12288, 4096, False), ("wo", T, 4096, 4096, True), ("w13", T, 22016, 4096, False), ("w2", T, 4096, 11008, True)]
a0 = torch.randn(T, 4096, device=dev, dtype=BF); w0 = torch.randn(4096, 4096, device=dev, dtype=BF) * 0.02; o0 = torch.zeros(T, 4096, device=dev, dtype=BF)
t0 = time.time()
while time.time() - t0 < 0.6:
    ops.gemm_nt(a0, w0, o0); torch.cuda.synchronize()
for name, M, N, K, res in shapes:
    a = torch.randn(M, K, device=dev, dtype=BF); w = torch.randn(N, K, device=dev, dtype=BF) * 0.02
    out = torch.zeros(M, N, device=dev, dtype=BF)
    ts = {"auto": [], "ring_all": []}
    for r in range(7):
        for k, f in (("auto", 0), ("ring_all", lib.EPI_TILE_256PP)):
            ops.gemm_nt(a, w, out, residual=out if res else None, epilogue=f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm_nt(a, w, out, residual=out if res else None, epilogue=f)
            e1.record(); torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 4)
    fl = 2.0 * M * N * K
    print(json.dumps({"shape": name, "M": M, **{k: round(sorted(v)[3] * 1e3, 1) for k, v in ts.items()}, **{k + "_tf": round(fl / sorted(v)[3] / 1e9, 1) for k, v in ts.items()}}), flush=True)
