"""Input-gradient product on the 7B shapes: a3v_gemm_nn on the forward weight image vs a3v_gemm_nt on a transposed copy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import ops  # noqa: E402
from tools.gemm_fp8_bench import t_us  # noqa: E402

DEV = "cuda:0"
T = 8728
for name, N, K in [("qkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]:
    dy = (torch.randn(T, N, device=DEV) * 0.1).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    wt = w.t().contiguous()
    dx = torch.empty(T, K, device=DEV, dtype=torch.bfloat16)
    a = t_us(lambda: ops.gemm_nt(dy, wt, dx))
    b = t_us(lambda: ops.gemm_nn(dy, w, dx))
    c = t_us(lambda: ops.transpose(w, wt, N, K, N))
    fl = 2.0 * T * N * K
    print(f"{name:4s} dX[{T},{K}] = dY[{T},{N}] W: nt {a:7.1f} us ({fl / a / 1e6:6.1f} TF)  nn {b:7.1f} us ({fl / b / 1e6:6.1f} TF)  transpose of W {c:5.1f} us", flush=True)
