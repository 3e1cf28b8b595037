#!/usr/bin/env python3
"""A/B of the two hd-128 prefill kernels inside one process (A3V_ATTN_W64 is read per launch): 64-row waves, one wave per SIMD
(attn_prefill_w64_kernel) against the 32-row kernel.  Equality up to bf16 rounding of P / the order of the fp32 sums on odd shapes
(ragged tails, Sq != Sk, GQA, non-causal, spikes that force the rescale path), then interleaved timing rounds on random data."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
from a3vlm_amd import lib as L
dev = "cuda"
def setv(v):
    os.environ["A3V_ATTN_W64"] = v
    L.load().a3v_reload_env()
def mk(B, Sq, Sk, H, Hkv, hd, spike=False, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    sp = (Sk + 63) // 64 * 64
    q = torch.randn(B, Sq, H, hd, device=dev, dtype=torch.bfloat16, generator=g)
    k = torch.randn(B, Hkv, sp, hd, device=dev, dtype=torch.bfloat16, generator=g)
    vt = torch.randn(B, Hkv, hd, sp, device=dev, dtype=torch.bfloat16, generator=g)
    if spike:   # late keys with large scores: rows outgrow the lazy-rescale bound several times
        for j in range(3, Sk, 97):
            k[:, :, j] *= (4.0 + (j % 5))
    st = (Sq*H*hd, H*hd, hd, Hkv*sp*hd, sp*hd, hd, Hkv*hd*sp, hd*sp, sp, Sq*H*hd, H*hd, hd)
    return q, k, vt, st, sp
def ref(q, k, vt, Sq, Sk, H, Hkv, hd, causal):
    qf = q.float().permute(0, 2, 1, 3)                                   # B H Sq hd
    kf = k.float()[:, :, :Sk].repeat_interleave(H // Hkv, dim=1)         # B H Sk hd
    vf = vt.float()[:, :, :, :Sk].repeat_interleave(H // Hkv, dim=1)     # B H hd Sk
    s = qf @ kf.transpose(-1, -2) / hd ** 0.5
    if causal:
        i = torch.arange(Sq, device=dev)[:, None] + (Sk - Sq); j = torch.arange(Sk, device=dev)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ vf.transpose(-1, -2)
    return o.permute(0, 2, 1, 3), lse
ok = True
for (B, Sq, Sk, H, Hkv, causal, spike) in [(2, 100, 100, 4, 4, True, False), (2, 129, 129, 4, 2, True, True), (1, 300, 300, 2, 2, False, False),
                                          (2, 64, 200, 4, 4, True, False), (1, 257, 321, 8, 2, True, True), (2, 1091, 1091, 4, 4, True, True), (1, 191, 191, 3, 3, False, True),
                                          (1, 128, 128, 2, 2, True, False), (1, 127, 127, 2, 2, True, False), (1, 65, 1000, 2, 1, True, False)]:
    hd = 128
    q, k, vt, st, sp = mk(B, Sq, Sk, H, Hkv, hd, spike)
    outs, lses = {}, {}
    for v in ("1", "0"):
        setv(v)
        outs[v] = torch.zeros_like(q); lses[v] = torch.zeros(B, H, Sq, device=dev)
        ops.attention_lse(q, k, vt, outs[v], lses[v], B, Sq, Sk, H, Hkv, hd, st, causal)
    torch.cuda.synchronize()
    ro, rl = ref(q, k, vt, Sq, Sk, H, Hkv, hd, causal)
    e1 = float((outs["1"].float() - ro).abs().max()); e0 = float((outs["0"].float() - ro).abs().max())
    l1 = float((lses["1"] - rl).abs().max()); l0 = float((lses["0"] - rl).abs().max())
    d = float((outs["1"].float() - outs["0"].float()).abs().max())
    good = e1 <= max(2.5 * e0, 2e-2) and l1 <= max(2.5 * l0, 2e-3) and e1 == e1
    ok &= good
    print(json.dumps(dict(B=B, Sq=Sq, Sk=Sk, H=H, Hkv=Hkv, causal=causal, spike=spike, err_w64=round(e1, 5), err_32=round(e0, 5), lse_w64=round(l1, 6), lse_32=round(l0, 6),
                          w64_vs_32=round(d, 5), ok=good)), flush=True)
print("PARITY", "ok" if ok else "FAILED", flush=True)
if "--no-time" in sys.argv: sys.exit(0 if ok else 1)
for (B, S, H, causal) in [(8, 1091, 32, True), (8, 2182, 32, True), (8, 1967, 32, True), (8, 1091, 40, True), (8, 2048, 32, False)]:
    hd = 128
    q, k, vt, st, sp = mk(B, S, S, H, H, hd)
    o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev)
    times = {"1": [], "0": []}
    for r in range(5):
        for v in ("1", "0"):
            setv(v)
            f = lambda: ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, causal)
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5 * 1e-3)
    fl = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    print(json.dumps(dict(B=B, S=S, H=H, causal=causal, w64_us=round(med["1"] * 1e6, 1), k32_us=round(med["0"] * 1e6, 1), w64_tf=round(fl / med["1"] / 1e12, 1),
                          k32_tf=round(fl / med["0"] / 1e12, 1), speedup=round(med["0"] / med["1"], 3))), flush=True)
