#!/usr/bin/env python3
"""Adapter weight-gradient strips out[64, N] = T[tokens, 64]^T . X[tokens, N]: the TN split-K form (both operands token-major, transposing
LDS reads on both) against the NN form on a pre-transposed T^T [64, tokens] (row-major A fragments, transposing reads on X only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import ops  # noqa: E402
from tools.gemm_fp8_bench import t_us  # noqa: E402

dev = "cuda"
T = 8728
for N in (4096, 11008, 12288, 22016):
    t = (torch.randn(T, 64, device=dev) * 0.5).bfloat16()
    x = (torch.randn(T, N, device=dev) * 0.5).bfloat16()
    g = torch.empty(64, N, device=dev, dtype=torch.float32)
    blocks = (N + 255) // 256
    S = 1
    while blocks * S < 256 and S < 16 and 2 * S <= (T + 63) // 64:
        S *= 2
    sc = torch.empty(S * 64 * N, device=dev, dtype=torch.float32)
    us_tn = t_us(lambda: ops.gemm_tn_splitk(t, x, g, sc, S))
    want = g.clone()
    Tp = (T + 63) // 64 * 64
    tT = torch.zeros(64, Tp, device=dev, dtype=torch.bfloat16)
    tT[:, :T] = t.t()
    xp = x if Tp == T else torch.cat([x, torch.zeros(Tp - T, N, device=dev, dtype=torch.bfloat16)])
    g2 = torch.empty(64, N, device=dev, dtype=torch.float32)
    try:
        us_nn = t_us(lambda: ops.gemm_nn(tT, xp, g2, epilogue=ops.EPI_OUT_F32))
        err = float((g2 - want).abs().max() / want.abs().max())
    except Exception as e:
        us_nn, err = float("nan"), repr(e)[:80]
    us_tr = t_us(lambda: ops.transpose(t, tT, T, 64, Tp))
    mb = T * N * 2 / 1e6
    print(f"N={N:6d}: tn split-K S={S:2d} {us_tn:6.1f} us ({mb / us_tn:.2f} TB/s)   nn on T^T {us_nn:6.1f} us ({mb / us_nn:.2f} TB/s)  rel diff {err}   transpose of T {us_tr:.1f} us", flush=True)
