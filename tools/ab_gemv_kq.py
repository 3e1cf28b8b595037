#!/usr/bin/env python3
"""A/B of the two decode GEMV forms inside one process (A3V_GEMV_KQ is read per launch): K slices as the waves of one block per tile
(gemv_kq_bf16_kernel, partials meet in LDS) against K slices across blocks (gemv_dma_bf16_kernel, split-K fix-up through HBM).
Values against an fp32 torch product, equality of the two forms, then interleaved timing on the 7B decode shapes (M = 8)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
from a3vlm_amd import lib as L
dev = "cuda"
def setv(v):
    os.environ["A3V_GEMV_KQ"] = v
    L.load().a3v_reload_env()
torch.manual_seed(0)
ok = True
for (M, N, K, epi) in [(8, 4096, 4096, 0), (8, 12288, 4096, 0), (8, 4096, 11008, ops.EPI_RESIDUAL), (8, 22016, 4096, ops.EPI_SWIGLU), (8, 256, 256, 0), (3, 4096, 4096, 0),
                       (8, 32000, 4096, ops.EPI_OUT_F32), (5, 10240, 5120, ops.EPI_SWIGLU), (1, 64, 512, 0)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    ws = ops.gemm_skinny_workspace(M, N, K, dev)
    No = N // 2 if epi & ops.EPI_SWIGLU else N
    res = torch.randn(M, No, device=dev, dtype=torch.bfloat16) if epi & ops.EPI_RESIDUAL else None
    outs = {}
    for v in ("1", "0"):
        setv(v)
        out = torch.zeros(M, No, device=dev, dtype=torch.float32 if epi & ops.EPI_OUT_F32 else torch.bfloat16)
        ops.gemm_skinny(a, w, out, ws, residual=res, epilogue=epi)
        torch.cuda.synchronize()
        outs[v] = out.float()
    y = a.float() @ w.float().t()
    if epi & ops.EPI_SWIGLU:
        y = y.view(M, N // 32, 2, 16)
        g, u = y[:, :, 0].to(torch.bfloat16).float(), y[:, :, 1].to(torch.bfloat16).float()
        y = (torch.nn.functional.silu(g).to(torch.bfloat16).float() * u).reshape(M, N // 2)
    if res is not None:
        y = y.to(torch.bfloat16).float() + res.float()
    e1 = float((outs["1"] - y).abs().max() / y.abs().max()); e0 = float((outs["0"] - y).abs().max() / y.abs().max())
    good = e1 < 2e-2 and e1 <= 2 * e0 + 1e-3
    ok &= good
    print(json.dumps(dict(M=M, N=N, K=K, epi=epi, err_kq=round(e1, 5), err_across=round(e0, 5), kq_vs_across=float((outs["1"] - outs["0"]).abs().max()), ok=good)), flush=True)
print("PARITY", "ok" if ok else "FAILED", flush=True)
for (M, N, K, epi) in [(8, 12288, 4096, 0), (8, 4096, 4096, ops.EPI_RESIDUAL), (8, 22016, 4096, ops.EPI_SWIGLU), (8, 4096, 11008, ops.EPI_RESIDUAL), (8, 32000, 4096, ops.EPI_OUT_F32),
                       (8, 15360, 5120, 0), (8, 5120, 5120, ops.EPI_RESIDUAL), (8, 27648, 5120, ops.EPI_SWIGLU), (8, 5120, 13824, ops.EPI_RESIDUAL), (8, 32000, 5120, ops.EPI_OUT_F32)]:
    L_ = 8       # rotate through several weight copies so that nothing stays in the caches
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    ws_ = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(L_)]
    ws = ops.gemm_skinny_workspace(M, N, K, dev)
    No = N // 2 if epi & ops.EPI_SWIGLU else N
    res = torch.randn(M, No, device=dev, dtype=torch.bfloat16) if epi & ops.EPI_RESIDUAL else None
    out = torch.zeros(M, No, device=dev, dtype=torch.float32 if epi & ops.EPI_OUT_F32 else torch.bfloat16)
    times = {"1": [], "0": []}
    for r in range(5):
        for v in ("1", "0"):
            setv("2" if v == "1" else "0")
            for w in ws_[:2]: ops.gemm_skinny(a, w, out, ws, residual=res, epilogue=epi)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in ws_: ops.gemm_skinny(a, w, out, ws, residual=res, epilogue=epi)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / L_ * 1e-3)
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    by = N * K * 2.0
    print(json.dumps(dict(M=M, N=N, K=K, epi=epi, kq_us=round(med["1"] * 1e6, 1), across_us=round(med["0"] * 1e6, 1), kq_TBs=round(by / med["1"] / 1e12, 2),
                          across_TBs=round(by / med["0"] / 1e12, 2), speedup=round(med["0"] / med["1"], 3))), flush=True)
