"""bf16 NT GEMM on the CLIP ViT-L/14-336 shapes of the headline batch (8 images x 577 tokens = 4616 rows): automatic dispatch vs
the forced 128x128 / 256x256 ping-pong kernels vs a manual split (ping-pong on the first 4096 rows, 128x128 on the rest)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import lib, ops  # noqa: E402
from tools.gemm_fp8_bench import t_us  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4616
for name, N, K in [("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]:
    a = (torch.randn(M, K, device=DEV) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    gf = 2.0 * M * N * K / 1e6
    t0 = t_us(lambda: ops.gemm_nt(a, w, out))
    t1 = t_us(lambda: ops.gemm_nt(a, w, out, epilogue=lib.EPI_TILE_128))
    t2 = t_us(lambda: ops.gemm_nt(a, w, out, epilogue=lib.EPI_TILE_256PP))
    mb = M // 256 * 256 if M // 256 * 256 < M else M - 256

    def hyb():
        ops.gemm_nt(a[:mb], w, out[:mb], epilogue=lib.EPI_TILE_256PP)
        ops.gemm_nt(a[mb:], w, out[mb:], epilogue=lib.EPI_TILE_128)
    t3 = t_us(hyb)
    print(f"{name} M={M} N={N} K={K}: auto {t0:.1f} us ({gf / t0:.0f} TF)  t128 {t1:.1f}  pp {t2:.1f}  pp[:{mb}]+t128 {t3:.1f}", flush=True)
