"""Weight-only fp8 (OCP e4m3fn, the gfx950 format) quantisation of the decode weight images -- BASELINE config 5 /
SURVEY 8(a) row Q.  The reference's quantised path is bitsandbytes NF4 / INT8 (util/quant.py:95-163, CUDA only): there is
no reference oracle for fp8, so this path is opt-in (``Transformer.quantize_decode_weights()``), never the default, and
its parity statement is "the bf16 kernels on the dequantised weights" (tests/test_gpu_fp8.py).

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
