#!/usr/bin/env python3
"""Cycle anatomy of the ring ping-pong GEMM (dbg 6 = stamped build of gemm_nt_bf16_ring_kernel; dbg 10: its 32x32x16 form),
block 0, waves 0 and 4.  Per K-tile (the block's LAST tile): 0 L-start (DMA issue + fragment reads follow) 1 reads returned
2 counted vmcnt passed 3 barrier passed 4 MFMA done 5 second counted vmcnt passed (group 0 only) 6 barrier passed.
Per tile: k-loop entry, k-loop exit, next prologue issued, epilogue issued."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import lib, ops
DBG = int(sys.argv[1]) if len(sys.argv) > 1 else 6
shapes = [(1024, 2048, 8192, 0), (1024, 2048, 4096, 0), (8192, 8192, 8192, 0), (8728, 22016, 4096, ops.EPI_SWIGLU), (8728, 22016, 4096, 0), (8728, 4096, 11008, 0), (4616, 4096, 1024, 0)]
for (M, N, K, epi) in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    buf = torch.zeros(2 * 64 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        buf.zero_()
        rc = lib.load().a3v_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), N if not epi else N // 2, M, N, K, buf.data_ptr(), None, 0,
                                    epi | lib.EPI_TILE_256PP | (DBG << 24), 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    torch.cuda.synchronize()
    st = buf.cpu().view(2, 64, 8)
    nk = K // 64
    print(f"shape {M}x{N}x{K} epi {epi}")
    for g in range(2):
        hi = min(60, nk - 1)
        per = [int(st[g, t + 1, 0] - st[g, t, 0]) for t in range(4, hi)]
        cols = [[int(st[g, t, i + 1] - st[g, t, i]) for t in range(4, hi)] for i in range(6)]
        print(f"  group {g}: median K-tile period {statistics.median(per)}  segments issue+read {statistics.median(cols[0])} vmcnt {statistics.median(cols[1])} "
              f"barrier {statistics.median(cols[2])} mfma {statistics.median(cols[3])} vmcnt2 {statistics.median(cols[4])} barrier2 {statistics.median(cols[5])}")
        tl = st[g, :, 7]
        for n in range(12):
            r = [int(tl[n * 5 + k]) for k in range(4)]
            nxt = int(tl[(n + 1) * 5]) if n < 11 else 0
            if r[0] == 0:
                break
            print(f"     tile {n}: k-loop {r[1] - r[0]} ({(r[1] - r[0]) / nk:.0f}/K-tile)  prologue issue {r[2] - r[1]}  epilogue {r[3] - r[2]}"
                  + (f"  to next k-loop entry {nxt - r[3]}  tile total {nxt - r[0]}" if nxt else ""))
