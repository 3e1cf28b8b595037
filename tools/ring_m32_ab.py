#!/usr/bin/env python3
"""(EXPERIMENTS=1 build) ring kernel on v_mfma_f32_16x16x32_bf16 (the product form) against the same kernel on 32x32x16 (dbg 9) and,
for a like-for-like comparison of the MFMA shape alone, against the ring with direct (unstaged) epilogue stores (dbg 11), on the
LoRA step's shapes; interleaved rounds in one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import lib, ops  # noqa: E402

assert lib.has_experiments(), "build with `make -C a3vlm_amd/csrc EXPERIMENTS=1`"
dev = "cuda"
T = lib.EPI_TILE_256PP
V = {"ring": 5, "ring_direct": 11, "ring32": 9}
shapes = [(8728, 12288, 4096), (8728, 4096, 4096), (8728, 22016, 4096), (8728, 4096, 11008), (8728, 4096, 12352), (8728, 4096, 22080),
          (8728, 11008, 4160), (8192, 8192, 8192)]


def ev(fn, reps=6):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for _ in range(40):
    x @ x
torch.cuda.synchronize()
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    best = {k: 1e9 for k in V}
    for rnd in range(3):
        for k, d in V.items():
            best[k] = min(best[k], ev(lambda: ops.gemm_nt(a, w, o, epilogue=T | (d << 24))))
    fl = 2.0 * M * N * K
    print(f"{str((M, N, K)):>22} " + "  ".join(f"{k} {best[k]:7.1f} us {fl / best[k] / 1e6:7.1f} TF" for k in V), flush=True)
    del a, w, o
