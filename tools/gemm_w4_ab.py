"""A/B of the one-wave-per-SIMD NT kernel (A3V_GEMM_W4=1) against the ring kernel: equality of results and TFLOP/s."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from a3vlm_amd import ops  # noqa: E402

dev = "cuda:0"
shapes = [(8192, 8192, 8192), (8728, 22016, 4096), (8728, 4096, 11008), (8728, 12288, 4096), (8728, 4096, 4096), (1000, 768, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K in shapes:
    a = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.5).bfloat16()
    outs = {}
    line = f"{M}x{N}x{K}:"
    for mode in os.environ.get("W4_MODES", "0,1,0,1").split(","):
        os.environ["A3V_GEMM_W4"] = mode
        __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm_nt(a, w, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.gemm_nt(a, w, c)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        line += f"  {'ring' if mode == '0' else 'w4.' + mode} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
        outs[mode] = c
    cmp = os.environ.get("W4_CMP", "1")
    if cmp not in outs or "0" not in outs:
        print(line, flush=True)
        continue
    eq = torch.equal(outs["0"], outs[cmp])
    ref = (a[:256].float() @ w[:512].float().T)
    err = float((outs[cmp][:256, :512].float() - ref).abs().max() / ref.abs().max())
    dmax = float((outs["0"].float() - outs[cmp].float()).abs().max())
    print(line, " equal" if eq else f"  DIFFERENT max|d|={dmax:.4g}", f"relerr_vs_fp32={err:.2e}", flush=True)
