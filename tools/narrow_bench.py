#!/usr/bin/env python3
"""a3v_gemm_nt_narrow (one pass) against a3v_gemm_nt_splitk + a3v_splitk_reduce on the eight adapter projections of a 7B LoRA layer
(t = x A^T forward, dt = dy B backward; M = 8728 tokens, 64 = padded rank): values vs an fp32 product of the same bf16 operands, and
time per call over a rotation of operand buffers larger than the Infinity Cache."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
from a3vlm_amd.train import skinny_slices
dev = "cuda"
M = int(os.environ.get("M", 8728))
cus = torch.cuda.get_device_properties(0).multi_processor_count
for (name, K, ld_extra) in [("t_qkv / t_wo / t_w13 (x 71 MB)", 4096, 64), ("t_w2 (act 192 MB)", 11008, 64), ("dt_qkv (dy 214 MB)", 12288, 64),
                            ("dt_w13 (dy 385 MB)", 22016, 64), ("dt_wo / dt_w2 (dy 71 MB)", 4096, 64)]:
    nbuf = max(2, int(600e6 // (M * (K + ld_extra) * 2)) + 1)
    fulls = [torch.randn(M, K + ld_extra, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]       # [x | t]: the product's output sits behind its operand
    w = torch.zeros(64, K, device=dev, dtype=torch.bfloat16)
    w[:48] = torch.randn(48, K, device=dev, dtype=torch.bfloat16) * 0.05
    S = skinny_slices(M, 64, K, cus)
    scratch = torch.empty(max(S, 1) * M * 64, device=dev, dtype=torch.float32)
    a0, t0 = fulls[0][:, :K], fulls[0][:, K:]
    ref = (a0.float() @ w.float().t())
    ops.gemm_nt_narrow(a0, w, t0)
    got_n = t0.float().clone()
    ops.gemm_nt_splitk(a0, w, t0, scratch, S)
    got_s = t0.float().clone()
    tf = torch.empty(M, 64, device=dev, dtype=torch.float32)
    ops.gemm_nt_narrow(a0, w, tf)
    sc = float(ref.abs().max())
    res = dict(shape=name, M=M, K=K, S=S, err_narrow=float((got_n - ref).abs().max()) / sc, err_splitk=float((got_s - ref).abs().max()) / sc,
               err_narrow_f32=float((tf - ref).abs().max()) / sc, narrow_vs_splitk_max=float((got_n - got_s).abs().max()) / sc,
               zero_rows_exact=bool((got_n[:, 48:] == 0).all()))
    def timeit(f):
        for i in range(nbuf): f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(3 * nbuf): f(i % nbuf)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (3 * nbuf) * 1e3
    res["narrow_us"] = round(timeit(lambda i: ops.gemm_nt_narrow(fulls[i][:, :K], w, fulls[i][:, K:])), 1)
    res["splitk_us"] = round(timeit(lambda i: ops.gemm_nt_splitk(fulls[i][:, :K], w, fulls[i][:, K:], scratch, S)), 1)
    res["narrow_tb_s"] = round(M * K * 2 / res["narrow_us"] / 1e6, 2)
    print(json.dumps(res), flush=True)
    del fulls
# ragged M and row clamp
for M2 in (33, 511, 1000, 8729):
    K = 4096
    a = torch.randn(M2, K, device=dev, dtype=torch.bfloat16); w = torch.randn(64, K, device=dev, dtype=torch.bfloat16) * 0.05
    out = torch.full((M2 + 3, 64), 7.0, device=dev, dtype=torch.bfloat16)
    ops.gemm_nt_narrow(a, w, out[:M2])
    ref = a.float() @ w.float().t()
    print(json.dumps(dict(M=M2, err=float((out[:M2].float() - ref).abs().max()) / float(ref.abs().max()), rows_past_M_untouched=bool((out[M2:] == 7.0).all()))), flush=True)
