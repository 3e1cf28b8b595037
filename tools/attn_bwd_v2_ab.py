#!/usr/bin/env python3
"""A/B of the attention backward: A3V_ATTN_BWD_V2 bit masks (bit 0: one-pass dK + dV kernel, bit 1: pipelined dQ kernel) against the
round-1 kernels (0).  Checks bit-equality of dq / dk / dv (same per-accumulator MFMA order) and times the whole backward."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import lib, ops  # noqa: E402

dev = "cuda"
BF = torch.bfloat16
variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
L = lib.load()


def setv(v):
    os.environ["A3V_ATTN_BWD_V2"] = str(v)
    L.a3v_reload_env()


cases = [(8, 1091, 32, 32, 128, True), (8, 2182, 32, 32, 128, True), (8, 1967, 32, 32, 128, True), (2, 517, 8, 2, 128, True), (2, 300, 4, 4, 64, True), (2, 577, 16, 16, 64, False),
         (1, 64, 2, 2, 128, True), (1, 33, 2, 1, 128, True), (3, 129, 4, 4, 128, False)]
for (B, S, H, Hkv, hd, causal) in cases[:int(os.environ.get('A3V_AB_CASES', '99'))]:
    torch.manual_seed(S)
    sp = (S + 63) // 64 * 64
    q = torch.randn(B, S, H, hd, device=dev, dtype=BF)
    kc = torch.randn(B, Hkv, sp, hd, device=dev, dtype=BF)
    vt = torch.randn(B, Hkv, hd, sp, device=dev, dtype=BF)
    v = vt.transpose(2, 3)[:, :, :S].permute(0, 2, 1, 3).contiguous()            # [B, S, Hkv, hd]
    o = torch.empty_like(q)
    lse = torch.empty(B, H, S, device=dev)
    st = (S * H * hd, H * hd, hd, Hkv * sp * hd, sp * hd, hd, Hkv * hd * sp, hd * sp, sp, S * H * hd, H * hd, hd)
    ops.attention_lse(q, kc, vt, o, lse, B, S, S, H, Hkv, hd, st, causal)
    do = torch.randn_like(q) * 0.1
    D = torch.empty(B, S, H, device=dev)
    ws = torch.empty(256, dtype=torch.uint8, device=dev)
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    cs = precompute_cos_sin(hd, 2 * sp, 10000.0, None).to(dev)
    N = (H + 2 * Hkv) * hd
    res, times, ptimes = {}, {}, {}

    def timed(f):
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3 * 1e3)
        return round(sorted(ts)[2], 1)

    for var in variants:
        setv(var)
        dq = torch.full_like(q, float("nan"))
        dk = torch.full((B, Hkv, S, hd), float("nan"), device=dev, dtype=BF)
        dv = torch.full_like(dk, float("nan"))
        f = lambda: ops.attention_bwd(q, kc, Hkv * sp * hd, sp * hd, v, S * Hkv * hd, Hkv * hd, hd, o, do, lse, D, dq, dk, dv, B, S, H, Hkv, hd, causal, workspace=ws)
        f()
        packed = torch.full((B * S, N + 8), 7.0, dtype=BF, device=dev)
        fp = lambda: ops.attention_bwd_packed(q, kc, Hkv * sp * hd, sp * hd, v, S * Hkv * hd, Hkv * hd, hd, o, do, lse, D, packed[:, :N], cs, B, S, H, Hkv, hd, causal, 0)
        fp()
        torch.cuda.synchronize()
        res[var] = (dq.clone(), dk.clone(), dv.clone(), packed.clone())
        times[var] = timed(f)
        ptimes[var] = timed(fp)
    base = res[variants[0]]
    eq = {}
    for var in variants[1:]:
        eq[var] = [bool(torch.equal(a, b)) for a, b in zip(res[var], base)]
        if not all(eq[var]):
            eq[var] += [float((a.float() - b.float()).abs().max()) for a, b in zip(res[var], base)]
    finite = all(bool(torch.isfinite(t).all()) for t in base)
    print(json.dumps(dict(B=B, S=S, H=H, Hkv=Hkv, hd=hd, causal=causal, us=times, packed_us=ptimes, equal_to_first=eq, finite=finite)), flush=True)
os.environ.pop("A3V_ATTN_BWD_V2", None)
L.a3v_reload_env()
