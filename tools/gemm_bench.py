#!/usr/bin/env python3
"""GPU micro-benchmark of a3v_gemm_nt tile configurations on the bench step's GEMM shapes.
Interleaved rounds in ONE process (guide rule 24); random data (rule 25)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops, lib

dev = "cuda"
shapes = [(8728, 12288, 4096, 0), (8728, 4096, 4096, ops.EPI_RESIDUAL), (8728, 22016, 4096, ops.EPI_SWIGLU),
          (8728, 4096, 11008, ops.EPI_RESIDUAL), (4616, 3072, 1024, 0), (4616, 1024, 1024, 0), (4616, 4096, 1024, 0),
          (4616, 1024, 4096, 0), (8192, 8192, 8192, 0), (4096, 4096, 4096, 0)]
cfgs = {"t128": lib.EPI_TILE_128, "pp": lib.EPI_TILE_256PP, "auto": 0}
if os.environ.get("PP_ONLY"):
    cfgs = {"pp": lib.EPI_TILE_256PP, "pp_s1": lib.EPI_TILE_256PP | (8 << 24)}
    shapes = [(8192, 8192, 8192, 0), (8728, 22016, 4096, ops.EPI_SWIGLU), (8728, 12288, 4096, 0), (8728, 4096, 4096, ops.EPI_RESIDUAL), (8728, 4096, 11008, ops.EPI_RESIDUAL)]
if os.environ.get("ABLATE"):
    cfgs = {"pp": lib.EPI_TILE_256PP, "pp_nodma": lib.EPI_TILE_256PP | (1 << 24), "pp_nord": lib.EPI_TILE_256PP | (2 << 24),
            "pp_none": lib.EPI_TILE_256PP | (3 << 24)}
    shapes = [(8192, 8192, 8192, 0), (8728, 22016, 4096, 0)]
rounds, reps = 5, 4
for (M, N, K, epi) in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    ncol = N // 2 if epi & ops.EPI_SWIGLU else N
    outs = {k: torch.zeros(M, ncol, device=dev, dtype=torch.bfloat16) for k in cfgs}
    e = epi & ~ops.EPI_RESIDUAL
    times = {k: [] for k in cfgs}
    for r in range(rounds):
        for k, flag in cfgs.items():
            res = outs[k] if epi & ops.EPI_RESIDUAL else None
            if res is not None:
                outs[k].zero_()
            ops.gemm_nt(a, w, outs[k], residual=res, epilogue=e | flag)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.gemm_nt(a, w, outs[k], residual=res, epilogue=e | flag)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e-3 / reps)
    fl = 2.0 * M * N * K
    row = {"M": M, "N": N, "K": K}
    for k in cfgs:
        t = sorted(times[k])
        row[k + "_tf_med"] = round(fl / t[len(t) // 2] / 1e12, 1)
        row[k + "_tf_best"] = round(fl / t[0] / 1e12, 1)
    if not (epi & ops.EPI_RESIDUAL):
        if "t128" in outs and "pp" in outs and "auto" in outs:
            row["bit_equal"] = bool(torch.equal(outs["t128"], outs["auto"])) and bool(torch.equal(outs["t128"], outs["pp"]))
    if "pp_s1" in outs and "pp" in outs:
        d = (outs["pp_s1"].float() - outs["pp"].float()).abs().max().item()
        row["s1_vs_pp_maxabs"] = round(d, 5)
        row["pp_absmax"] = round(outs["pp"].float().abs().max().item(), 3)
    print(json.dumps(row), flush=True)
