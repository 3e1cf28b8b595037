#!/usr/bin/env python3
"""XCD start skew of the persistent ring GEMM (A3V_GEMM_SKEW = cycles per XCD index, read per launch): the eight XCDs' tile
boundaries -- and with them their 4 MB store bursts -- are spread over time instead of hitting HBM together.  Interleaved rounds
in one process, random operands."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops, lib

dev = "cuda"
skews = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1500,3000,4500,6000,9000".split(","))]
shapes = [(8728, 12288, 4096, 0), (8728, 4096, 4096, 0), (8728, 22016, 4096, ops.EPI_SWIGLU), (8728, 4096, 11008, 0),
          (8192, 8192, 8192, 0), (4096, 4096, 4096, 0)]
f = lib.EPI_TILE_256PP | (5 << 24)
for (M, N, K, epi) in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    out = torch.zeros(M, N // 2 if epi else N, device=dev, dtype=torch.bfloat16)
    ref = None
    times = {k: [] for k in skews}
    for r in range(6):
        for k in skews:
            os.environ["A3V_GEMM_SKEW"] = str(k)
            __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
            ops.gemm_nt(a, w, out, epilogue=epi | f)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm_nt(a, w, out, epilogue=epi | f)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e-3 / 4)
    fl = 2.0 * M * N * K
    print(json.dumps({"M": M, "N": N, "K": K, **{f"skew{k}_tf": round(fl / sorted(times[k])[3] / 1e12, 1) for k in skews}}), flush=True)
os.environ["A3V_GEMM_SKEW"] = "0"
__import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
