"""Adapter-sized products of one LoRA step at the 7B geometry (8728 tokens, rank pad 64): t = x A^T / dt = dy B (NT, N = 64) and
the adapter weight gradients dB = dy^T t, dA = dt^T x (TN strips), against the bytes each must stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import ops  # noqa: E402
from tools.gemm_fp8_bench import t_us  # noqa: E402

DEV = "cuda:0"
T, R = 8728, 64
BF = torch.bfloat16


def pick_s(blocks, nk, cap=32, want=512):
    S = 1
    while blocks * S < want and S * 2 <= nk and S < cap:
        S *= 2
    return S


tot = 0.0
for K in (4096, 11008, 12288, 22016):
    x = (torch.randn(T, K, device=DEV) * 0.5).to(BF)
    a = (torch.randn(R, K, device=DEV) * 0.02).to(BF)
    t = torch.empty(T, R, device=DEV, dtype=BF)
    mb = T * K * 2 / 1e6
    for S in ((int(os.environ["S"]),) if "S" in os.environ else (4, 8, 16, 32)):
        scratch = torch.empty(S * T * R, device=DEV, dtype=torch.float32)
        us = t_us(lambda: ops.gemm_nt_splitk(x, a, t, scratch, S))
        print(f"nt   [T,{K}] x [64,{K}]^T  S={S:2d}: {us:6.1f} us  ({mb / us:.2f} TB/s of the {mb:.0f} MB operand)", flush=True)
    # strips: dB[K, 64] = x^T t  (x plays dy), dA[64, K] = t^T x
    for nm, at, wt in (("tn dB", x, t), ("tn dA", t, x)):
        N_, K_ = at.shape[1], wt.shape[1]
        blocks = ((N_ + 255) // 256) * ((K_ + 255) // 256)
        S2 = 1
        while blocks * S2 < 256 and S2 < 16 and 2 * S2 <= (T + 63) // 64:
            S2 *= 2
        g = torch.empty(N_, K_, device=DEV, dtype=torch.float32)
        sc = torch.empty(S2 * N_ * K_, device=DEV, dtype=torch.float32)
        us2 = t_us(lambda: ops.gemm_tn_splitk(at, wt, g, sc, S2))
        print(f"{nm} [{N_},{K_}] over T  S={S2:2d}: {us2:6.1f} us  ({mb / us2:.2f} TB/s)", flush=True)
        if N_ <= 64:        # the small-block streaming kernel on the same product
            for S3 in sorted({max(1, -(-512 // ((K_ + 127) // 128))), max(1, -(-768 // ((K_ + 127) // 128))), max(1, -(-1024 // ((K_ + 127) // 128))), max(1, -(-1536 // ((K_ + 127) // 128)))}):
                sc3 = torch.empty(S3 * N_ * K_, device=DEV, dtype=torch.float32)
                us3 = t_us(lambda: ops.gemm_tn_strip(at, wt, g, sc3, S3))
                print(f"   strip [{N_},{K_}] over T  S={S3:2d}: {us3:6.1f} us  ({mb / us3:.2f} TB/s)", flush=True)
