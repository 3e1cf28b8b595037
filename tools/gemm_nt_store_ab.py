#!/usr/bin/env python3
"""Non-temporal stores in the GEMM fast epilogue forms (A3V_GEMM_NT_STORE, read per launch) on the 7B shapes, interleaved rounds."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
dev = "cuda"; BF = torch.bfloat16
T = 8728
a0 = torch.randn(T, 4096, device=dev, dtype=BF); w0 = torch.randn(4096, 4096, device=dev, dtype=BF) * 0.02; o0 = torch.zeros(T, 4096, device=dev, dtype=BF)
t0 = time.time()
while time.time() - t0 < 0.6:
    ops.gemm_nt(a0, w0, o0); torch.cuda.synchronize()
cases = [("nt qkv plain", "nt", T, 12288, 4096, None), ("nt wo res_f32", "nt", T, 4096, 4096, "resf"), ("nt w13 plain (train)", "nt", T, 22016, 4096, None),
         ("nt w13 swiglu", "nt", T, 22016, 4096, "swiglu"), ("nt w2 residual", "nt", T, 4096, 11008, "res"),
         ("nn dgrad w13", "nn", T, 4096, 22016, None), ("nn dgrad w2", "nn", T, 11008, 4096, None), ("tn wgrad w13 acc", "tn", 22016, 4096, T, "resf"),
         ("tn wgrad qkv acc", "tn", 12288, 4096, T, "resf")]
for (name, fam, M, N, K, epi) in cases:
    if fam == "nt":
        a = torch.randn(M, K, device=dev, dtype=BF); w = torch.randn(N, K, device=dev, dtype=BF) * 0.02
    elif fam == "nn":
        a = torch.randn(M, K, device=dev, dtype=BF); w = torch.randn(K, N, device=dev, dtype=BF) * 0.02
    else:
        a = torch.randn(K, M, device=dev, dtype=BF); w = torch.randn(K, N, device=dev, dtype=BF) * 0.02
    if epi == "res":
        out = torch.zeros(M, N, device=dev, dtype=BF); kw = dict(residual=out)
    elif epi == "resf":
        out = torch.zeros(M, N, device=dev, dtype=torch.float32); kw = dict(residual=out, epilogue=ops.EPI_RES_F32)
    elif epi == "swiglu":
        out = torch.zeros(M, N // 2, device=dev, dtype=BF); kw = dict(epilogue=ops.EPI_SWIGLU)
    else:
        out = torch.zeros(M, N, device=dev, dtype=BF); kw = {}
    f = {"nt": ops.gemm_nt, "nn": ops.gemm_nn, "tn": ops.gemm_tn}[fam]
    ts = {"0": [], "1": []}
    for r in range(6):
        for v in ("0", "1"):
            os.environ["A3V_GEMM_NT_STORE"] = v
            __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
            f(a, w, out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): f(a, w, out, **kw)
            e1.record(); torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / 4)
    fl = 2.0 * M * N * K
    m = {v: sorted(t)[3] for v, t in ts.items()}
    print(json.dumps({"case": name, "plain_us": round(m["0"] * 1e3, 1), "nt_us": round(m["1"] * 1e3, 1), "nt_tf": round(fl / m["1"] / 1e9, 1), "ratio": round(m["0"] / m["1"], 3)}), flush=True)
os.environ.pop("A3V_GEMM_NT_STORE", None)
__import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
