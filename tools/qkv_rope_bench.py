"""Cost of the fused RoPE / cache-write epilogue: a3v_gemm_qkv_rope vs plain a3v_gemm_nt (+ the separate a3v_rope_kvcache)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import ops  # noqa: E402
from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin  # noqa: E402
from tools.gemm_fp8_bench import t_us  # noqa: E402

DEV = "cuda:0"
B, S, H, Hkv, hd, K = 8, 1091, 32, 32, 128, 4096
rows, N = B * S, (H + 2 * Hkv) * hd
Smax = 2048
x = (torch.randn(rows, K, device=DEV) * 0.5).bfloat16()
w = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
qkv = torch.empty(rows, N, device=DEV, dtype=torch.bfloat16)
kc = torch.zeros(B, Hkv, Smax, hd, device=DEV, dtype=torch.bfloat16)
vc = torch.zeros(B, Hkv, hd, Smax, device=DEV, dtype=torch.bfloat16)
cs = precompute_cos_sin(hd, 2 * Smax, 10000.0, None).to(DEV)
a = t_us(lambda: ops.gemm_nt(x, w, qkv))
b = t_us(lambda: ops.rope_kvcache(qkv, qkv, kc, vc, cs, B, S, H, Hkv, hd, 0, 0))
c = t_us(lambda: ops.gemm_qkv_rope(x, w, qkv, kc, vc, cs, B, S, H, Hkv, hd, 0, 0))
d = t_us(lambda: ops.gemm_qkv_rope(x, w, qkv, kc, vc, cs, B, S, H, Hkv, hd, 0, 0, v_rows=qkv[:, (H + Hkv) * hd:]))
print(f"gemm_nt {a:.1f} us + rope_kvcache {b:.1f} us = {a + b:.1f};  fused {c:.1f} us;  fused + v_rows {d:.1f} us")
