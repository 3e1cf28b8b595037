#!/usr/bin/env python3
"""Cycle anatomy of the one-wave-per-SIMD GEMM (A3V_GEMM_W4=2 = stamped build), block 0, all four waves, the block's LAST tile.
Per 32-k sub-stage: 0 start, 1 DMA issued + counted vmcnt passed, 2 barrier passed, 3 reads + MFMAs issued, 4 lgkmcnt(0) passed."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import lib
os.environ["A3V_GEMM_W4"] = "2"
__import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
shapes = [(8192, 8192, 8192), (8728, 22016, 4096)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    buf = torch.zeros(4 * 64 * 8 + 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        buf.zero_()
        rc = lib.load().a3v_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), N, M, N, K, buf.data_ptr(), None, 0,
                                    lib.EPI_TILE_256PP, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.load().a3v_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), N, M, N, K, buf.data_ptr(), None, 0,
                               lib.EPI_TILE_256PP, 0, torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = -(-tiles // 256)
    print(f"  stamped kernel {us:.1f} us, {rounds} tile rounds, {us / rounds / (K // 32) * 1e3:.1f} ns per sub-stage if tiles cost nothing else")
    for mode in ("2", "9", "8", "10"):
        os.environ["A3V_GEMM_W4"] = mode
        __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
        for _ in range(3):
            lib.load().a3v_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), N, M, N, K, buf.data_ptr(), None, 0,
                                   lib.EPI_TILE_256PP, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        pr = buf.cpu()[4 * 64 * 8:]
        cyc, rt = int(pr[2] - pr[0]), int(pr[3] - pr[1])
        print(f"  clock probe mode {mode}: block 0 ran {cyc} s_memtime ticks in {rt * 10} ns = {cyc / (rt * 10) :.3f} ticks/ns")
    os.environ["A3V_GEMM_W4"] = "2"
    __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
    lib.load().a3v_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), N, M, N, K, buf.data_ptr(), None, 0,
                           lib.EPI_TILE_256PP, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = buf.cpu()[:4 * 64 * 8].view(4, 64, 8)
    ns = K // 32
    print(f"shape {M}x{N}x{K}")
    for wv in range(4):
        hi = min(60, ns - 1)
        per = [int(st[wv, t + 1, 0] - st[wv, t, 0]) for t in range(8, hi)]
        cols = [[int(st[wv, t, i + 1] - st[wv, t, i]) for t in range(8, hi)] for i in range(4)]
        print(f"  wave {wv}: median sub-stage period {statistics.median(per)} (x2 = {2 * statistics.median(per)} per K-tile)  issue+vmcnt {statistics.median(cols[0])} "
              f"barrier {statistics.median(cols[1])} reads+mfma {statistics.median(cols[2])} lgkm {statistics.median(cols[3])}   max period {max(per)}")
        if wv == 0:
            for n in range(11):
                r = [int(st[wv, n, k]) for k in (5, 6, 7)]
                nx = int(st[wv, n + 1, 5])
                if r[0] == 0 or nx == 0:
                    break
                print(f"     tile {n}: wait vmcnt(0)+barrier {r[1] - r[0]}  k-loop {r[2] - r[1]} ({(r[2] - r[1]) / ns:.0f}/sub-stage)  barrier+prologue+epilogue {nx - r[2]}  total {nx - r[0]}")
