#!/usr/bin/env python3
"""Adapter-sized NT products of one LoRA step (t = x A^T, dt = dy B: 8728 x 64 x K) by LDS stages of the 64-row kernel
(A3V_SKINNY_STAGES: 2 = the two-stage kernel, 3 / 4 / 5 = gemm_nt_skinny_kernel), rows per block (A3V_SKINNY_NARROW) and split-K slices, the streamed operand
ROTATING through more than the 256 MB of the Infinity Cache (a fixed operand of 71-214 MB is partly served from it).
us per call incl. the reduce pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import lib, ops  # noqa: E402

DEV = "cuda:0"
T, R = 8728, 64
BF = torch.bfloat16
for K in (4096, 11008, 12288, 22016):
    mb = T * K * 2 / 1e6
    L = max(2, int(700 // mb) + 1)
    xs = [(torch.randn(T, K, device=DEV) * 0.5).to(BF) for _ in range(L)]
    a = (torch.randn(R, K, device=DEV) * 0.02).to(BF)
    t = torch.empty(T, R, device=DEV, dtype=BF)
    for narrow, rows, Ss in ((3, 64, (3, 4)), (2, 128, (3, 4, 6, 7)), (1, 256, (4, 5, 6, 7, 8, 14))):
        for nst in (2, 3, 4, 5):
            if nst > 2 and ((rows + 64) * 128 * nst > 160 * 1024 or (os.environ.get("PRODUCT_ONLY") and (rows, nst) not in ((64, 4), (256, 3)))):
                continue
            row = []
            for S in Ss:
                scratch = torch.empty(S * T * R, device=DEV, dtype=torch.float32)
                with lib.env(A3V_SKINNY_STAGES=nst, A3V_SKINNY_NARROW=narrow):
                    try:
                        for i in range(4):
                            ops.gemm_nt_splitk(xs[i % L], a, t, scratch, S)
                    except Exception:        # a rows / stages pair the product build does not carry (make EXTRA=-DA3V_ABLATION has them all)
                        row.append(f"S={S:2d}: not built")
                        continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    n = 40
                    e0.record()
                    for i in range(n):
                        ops.gemm_nt_splitk(xs[i % L], a, t, scratch, S)
                    e1.record()
                    torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                row.append(f"S={S:2d}: {us:6.1f} us {mb / us:4.2f} TB/s")
            print(f"K={K:5d} ({mb:3.0f} MB x {L})  rows={rows:3d} stages={nst}   " + "   ".join(row), flush=True)
