#!/usr/bin/env python3
"""Cycle anatomy of the ping-pong GEMM (needs the library built with EXTRA=-DA3V_ABLATION).
Stamps of block 0: group 0 (wave 0): 0 L-start 1 reads-done 2 after-barrier 3 mfma-done 4 dma-landed 5 after-barrier
                   group 1 (wave 4): 0 L-start 1 reads-done 2 dma-landed 3 after-barrier 4 mfma(+dma issue)-done 5 after-barrier"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops, lib
M, N, K = 8192, 8192, 8192
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
buf = torch.zeros(2 * 64 * 8, dtype=torch.int64, device="cuda")
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for _ in range(3):
    rc = lib.load().a3v_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), N, M, N, K, buf.data_ptr(), None, 0,
                                lib.EPI_TILE_256PP | (dbg << 24), 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
torch.cuda.synchronize()
st = buf.cpu().view(2, 64, 8)
for g in range(2):
    print(f"group {g}: per K-tile deltas (cycles) for t=20..27; columns: s0->s1, s1->s2, s2->s3, s3->s4, s4->s5, period")
    for t in range(20, 28):
        r = st[g, t]
        d = [int(r[i + 1] - r[i]) for i in range(5)] + [int(st[g, t + 1, 0] - r[0])]
        print("   ", d)
import statistics
for g in range(2):
    per = [int(st[g, t + 1, 0] - st[g, t, 0]) for t in range(8, 60)]
    cols = [[int(st[g, t, i + 1] - st[g, t, i]) for t in range(8, 60)] for i in range(5)]
    print(f"group {g} median period {statistics.median(per)}  segment medians {[statistics.median(c) for c in cols]}")
