"""TEST INFRASTRUCTURE ONLY (like everything under oracle/): a torch restatement of the fp8 (OCP e4m3fn, the gfx950 format)
quantiser and of the W8A8 arithmetic of BASELINE config 5 / SURVEY 8(a) row Q, against which the tests check the HIP quantiser
(``a3v_quantize_rows_fp8``), the fp8 GEMV / GEMM and the quantised model.  The reference's quantised path is bitsandbytes NF4 /
INT8 (util/quant.py:95-163, CUDA only): there is NO reference oracle for fp8 -- the parity statement is "the bf16 arithmetic on
the dequantised operands" (tests/test_gpu_fp8.py).  The product quantises with the HIP kernel, never with this file.

Per output row n: scale[n] = max|W[n, :]| / 448, Wq[n, k] = fp8(W[n, k] / scale[n]); the GEMV multiplies the fp32
accumulator of row n by scale[n].  The quantisation itself is one-time weight preparation (like weight packing)."""
from __future__ import annotations

from typing import Tuple

import torch

FP8_MAX = 448.0


def quantize_rows_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """w [N, K] (any float dtype) -> (uint8 view of float8_e4m3fn [N, K], fp32 scales [N])."""
    wf = w.detach().float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-12)
    scale = amax / FP8_MAX
    q = (wf / scale[:, None]).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale.contiguous()


def dequantize_rows_fp8(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.view(torch.float8_e4m3fn).float() * scale[:, None]


class W8A8OracleDecoder:
    """Mixin-free factory: ``make(ref_cpu.OracleDecoder subclass)`` -- the CPU restatement with the W8A8 arithmetic of the prefill
    path spelled out: per-row fp8 weights (dequantised), the input of every decoder linear fake-quantised per token
    (scale = max|x| / 448), fp32 product, one bf16 rounding."""

    @staticmethod
    def make(base):
        import torch.nn.functional as F

        class _W8A8(base):
            def lin(self, x, name):
                if not name.startswith("layers."):
                    return super().lin(x, name)
                w = dequantize_rows_fp8(*quantize_rows_fp8(self.sd[name + ".weight"]))
                xf = x.float()
                sc = xf.abs().amax(dim=-1, keepdim=True).clamp_min(1e-12) / FP8_MAX
                xq = (xf / sc).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float() * sc
                return F.linear(xq, w).to(x.dtype)
        return _W8A8
