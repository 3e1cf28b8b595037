#!/usr/bin/env python3
"""A few launches of one adapter-sized NT product (a3v_gemm_nt_splitk + a3v_splitk_reduce) for PMC passes.  usage: skinny_one.py M N K S reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
M, N, K, S, reps = (int(v) for v in sys.argv[1:6])
x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
a = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
t = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
scratch = torch.empty(S * M * N, device="cuda", dtype=torch.float32)
big = torch.empty(512 << 20, device="cuda", dtype=torch.uint8)
for _ in range(reps):
    big.fill_(1)                          # push the operand out of the Infinity Cache between launches
    ops.gemm_nt_splitk(x, a, t, scratch, S)
torch.cuda.synchronize()
