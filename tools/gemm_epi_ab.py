#!/usr/bin/env python3
"""Fast (interior-tile) epilogue forms against the general epilogue (A3V_GEMM_FAST_EPI=0): bit equality for every output kind
and kernel family on shapes with interior AND ragged tiles, then interleaved timing on the 7B shapes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops, lib

dev = "cuda"
BF = torch.bfloat16


def both(fn):
    outs = []
    for fast in ("1", "0"):
        os.environ["A3V_GEMM_FAST_EPI"] = fast
        __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
        outs.append(fn())
    os.environ["A3V_GEMM_FAST_EPI"] = "1"
    __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
    return outs


ok = True
g = torch.Generator(device=dev).manual_seed(5)
for (M, N, K) in [(1000, 776, 256), (2048, 1024, 512), (8728, 4096, 1024), (300, 264, 128)]:
    a = torch.randn(M, K, device=dev, dtype=BF, generator=g)
    w = torch.randn(N, K, device=dev, dtype=BF, generator=g) * 0.05
    res = torch.randn(M, N, device=dev, dtype=BF, generator=g)
    resf = torch.randn(M, N, device=dev, dtype=torch.float32, generator=g)
    for tile in (0, lib.EPI_TILE_256PP, lib.EPI_TILE_128):
        def plain():
            o = torch.full((M, N), 3.0, device=dev, dtype=BF)
            ops.gemm_nt(a, w, o, epilogue=tile)
            return o

        def residual():
            o = res.clone()
            ops.gemm_nt(a, w, o, residual=o, epilogue=tile)
            return o

        def res_f32():
            o = resf.clone()
            ops.gemm_nt(a, w, o, residual=o, epilogue=tile | ops.EPI_RES_F32)
            return o

        def out_f32():
            o = torch.full((M, N), 3.0, device=dev, dtype=torch.float32)
            ops.gemm_nt(a, w, o, epilogue=tile | ops.EPI_OUT_F32)
            return o

        def swiglu():
            o = torch.full((M, N // 2), 3.0, device=dev, dtype=BF)
            ops.gemm_nt(a, w, o, epilogue=tile | ops.EPI_SWIGLU)
            return o
        cases = {"plain": plain, "residual": residual, "res_f32": res_f32, "out_f32": out_f32}
        if N % 32 == 0:
            cases["swiglu"] = swiglu
        for name, fn in cases.items():
            x, y = both(fn)
            good = torch.equal(x, y)
            ok = ok and good
            print(json.dumps({"shape": [M, N, K], "tile": tile >> 16, "kind": name, "equal": good}), flush=True)
# TN / NN families (weight / input gradients): plain bf16, fp32 out, fp32 accumulate
for (M, N, K) in [(1024, 768, 1000), (4096, 4096, 2184), (520, 264, 512), (4352, 2048, 1024)]:
    at = torch.randn(K, M, device=dev, dtype=BF, generator=g)
    wt = torch.randn(K, N, device=dev, dtype=BF, generator=g) * 0.05
    a = torch.randn(M, K, device=dev, dtype=BF, generator=g)
    accf = torch.randn(M, N, device=dev, dtype=torch.float32, generator=g)

    def tn(dtype, acc=False):
        def run():
            o = accf.clone() if acc else torch.zeros(M, N, device=dev, dtype=dtype)
            ops.gemm_tn(at, wt, o, residual=o if acc else None, epilogue=ops.EPI_RES_F32 if acc else (ops.EPI_OUT_F32 if dtype == torch.float32 else 0))
            return o
        return run

    def nn(dtype):
        def run():
            o = torch.zeros(M, N, device=dev, dtype=dtype)
            ops.gemm_nn(a, wt, o, epilogue=ops.EPI_OUT_F32 if dtype == torch.float32 else 0)
            return o
        return run
    cases = {"tn_plain": tn(BF), "tn_f32": tn(torch.float32), "tn_accumulate": tn(torch.float32, True)}
    if K % 64 == 0:
        cases["nn_plain"] = nn(BF)
        cases["nn_f32"] = nn(torch.float32)
    for name, fn in cases.items():
        x, y = both(fn)
        good = torch.equal(x, y)
        ok = ok and good
        print(json.dumps({"shape": [M, N, K], "kind": name, "equal": good}), flush=True)
print("EQUALITY", "OK" if ok else "FAILED", flush=True)

# timing
T = 8728
shapes = [("nt plain qkv-shape", "nt", T, 12288, 4096, 0), ("nt residual wo", "nt", T, 4096, 4096, "res"), ("nt swiglu w13", "nt", T, 22016, 4096, ops.EPI_SWIGLU),
          ("nt residual w2", "nt", T, 4096, 11008, "res"), ("nt res_f32 wo (train)", "nt", T, 4096, 4096, "resf"),
          ("nt bias+gelu vit c_fc", "nt", 4616, 4096, 1024, "bias_gelu"), ("nt bias+residual vit c_proj", "nt", 4616, 1024, 4096, "bias_res"),
          ("nn dgrad w13", "nn", T, 4096, 22016, 0), ("nn dgrad qkv", "nn", T, 4096, 12288, 0),
          ("tn wgrad w13 accumulate", "tn", 22016, 4096, T, "resf"), ("tn wgrad wo accumulate", "tn", 4096, 4096, T, "resf")]
for (name, fam, M, N, K, epi) in shapes:
    if fam == "nt":
        a = torch.randn(M, K, device=dev, dtype=BF); w = torch.randn(N, K, device=dev, dtype=BF) * 0.02
    elif fam == "nn":
        a = torch.randn(M, K, device=dev, dtype=BF); w = torch.randn(K, N, device=dev, dtype=BF) * 0.02
    else:
        a = torch.randn(K, M, device=dev, dtype=BF); w = torch.randn(K, N, device=dev, dtype=BF) * 0.02
    if epi == "res":
        out = torch.zeros(M, N, device=dev, dtype=BF); kw = dict(residual=out, epilogue=0)
    elif epi == "resf":
        out = torch.zeros(M, N, device=dev, dtype=torch.float32); kw = dict(residual=out, epilogue=ops.EPI_RES_F32)
    elif epi == "bias_gelu":
        out = torch.zeros(M, N, device=dev, dtype=BF); kw = dict(bias=torch.randn(N, device=dev, dtype=BF), epilogue=ops.EPI_GELU)
    elif epi == "bias_res":
        out = torch.zeros(M, N, device=dev, dtype=BF); kw = dict(bias=torch.randn(N, device=dev, dtype=BF), residual=out, epilogue=0)
    elif epi == ops.EPI_SWIGLU:
        out = torch.zeros(M, N // 2, device=dev, dtype=BF); kw = dict(epilogue=epi)
    else:
        out = torch.zeros(M, N, device=dev, dtype=BF); kw = dict(epilogue=0)
    f = {"nt": ops.gemm_nt, "nn": ops.gemm_nn, "tn": ops.gemm_tn}[fam]
    times = {"1": [], "0": []}
    for r in range(6):
        for fast in ("1", "0"):
            os.environ["A3V_GEMM_FAST_EPI"] = fast
            __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
            f(a, w, out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                f(a, w, out, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[fast].append(e0.elapsed_time(e1) * 1e-3 / 4)
    fl = 2.0 * M * N * K
    tf = {k: fl / sorted(v)[3] / 1e12 for k, v in times.items()}
    print(json.dumps({"case": name, "fast_tf": round(tf["1"], 1), "general_tf": round(tf["0"], 1), "ratio": round(tf["1"] / tf["0"], 3)}), flush=True)
os.environ["A3V_GEMM_FAST_EPI"] = "1"
__import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
