"""bf16 GEMM epilogue cost on the 7B shapes: plain store vs bf16 residual vs fp32 residual (RES_F32) vs SwiGLU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import ops  # noqa: E402
from tools.gemm_fp8_bench import t_us  # noqa: E402

DEV = "cuda:0"
T = 8728
for name, N, K in [("wo", 4096, 4096), ("w2", 4096, 11008), ("qkv", 12288, 4096)]:
    a = (torch.randn(T, K, device=DEV) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    out = torch.zeros(T, N, device=DEV, dtype=torch.bfloat16)
    res = torch.randn(T, N, device=DEV).bfloat16()
    of = torch.zeros(T, N, device=DEV)
    t0 = t_us(lambda: ops.gemm_nt(a, w, out))
    t1 = t_us(lambda: ops.gemm_nt(a, w, out, residual=res))
    t2 = t_us(lambda: ops.gemm_nt(a, w, of, residual=of, epilogue=ops.EPI_RES_F32))
    t3 = t_us(lambda: ops.gemm_nt(a, w, of, epilogue=ops.EPI_OUT_F32))
    print(f"{name}: plain {t0:.1f} us, bf16 residual {t1:.1f}, f32 store {t3:.1f}, f32 residual {t2:.1f}", flush=True)
