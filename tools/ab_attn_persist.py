#!/usr/bin/env python3
"""A/B of the persistent prefill attention walk (A3V_ATTN_PERSIST, read per launch) against the one-block-per-unit launch inside one
process: bit-equality of O and the LSE on ragged / GQA / Sq != Sk shapes (incl. repeated launches: the unit counters must come back
to zero), then interleaved timing rounds on random data."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from a3vlm_amd import ops
from a3vlm_amd.lib import load
dev = "cuda"


def setv(v, grp=None):
    os.environ["A3V_ATTN_PERSIST"] = v
    if grp is None:
        os.environ.pop("A3V_ATTN_HEAD_GROUP", None)
    else:
        os.environ["A3V_ATTN_HEAD_GROUP"] = str(grp)
    load().a3v_reload_env()


def mk(B, Sq, Sk, H, Hkv, hd):
    sp = (Sk + 63) // 64 * 64
    q = torch.randn(B, Sq, H, hd, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, Hkv, sp, hd, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(B, Hkv, hd, sp, device=dev, dtype=torch.bfloat16)
    st = (Sq*H*hd, H*hd, hd, Hkv*sp*hd, sp*hd, hd, Hkv*hd*sp, hd*sp, sp, Sq*H*hd, H*hd, hd)
    return q, k, vt, st


ok = True
for (B, Sq, Sk, H, Hkv, hd, causal) in [(8, 1091, 1091, 32, 32, 128, True), (3, 700, 1000, 32, 8, 128, True), (8, 1091, 1091, 40, 40, 128, True),
                                        (5, 333, 333, 32, 32, 128, True), (40, 577, 577, 16, 16, 64, False), (24, 577, 577, 16, 16, 64, True),
                                        (16, 130, 130, 32, 32, 128, True), (2, 2182, 2182, 32, 32, 128, True), (16, 129, 200, 32, 4, 128, True)]:
    q, k, vt, st = mk(B, Sq, Sk, H, Hkv, hd)
    res = {}
    for v in ("0", "1"):
        setv(v)
        o = torch.full_like(q, float("nan")); lse = torch.full((B, H, Sq), float("nan"), device=dev)
        for _ in range(3):                          # repeated launches on the same stream: slots / counters reused
            ops.attention_lse(q, k, vt, o, lse, B, Sq, Sk, H, Hkv, hd, st, causal)
        torch.cuda.synchronize()
        res[v] = (o.clone(), lse.clone())
    eq = torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])
    ok &= eq
    print(json.dumps(dict(B=B, Sq=Sq, Sk=Sk, H=H, Hkv=Hkv, hd=hd, causal=causal, bit_identical=eq, finite=bool(torch.isfinite(res["1"][0].float()).all()))), flush=True)
# many launches back to back over > 256 counter slots, two streams
q, k, vt, st = mk(8, 1091, 1091, 32, 32, 128)
setv("0"); o0 = torch.empty_like(q); l0 = torch.empty(8, 32, 1091, device=dev); ops.attention_lse(q, k, vt, o0, l0, 8, 1091, 1091, 32, 32, 128, st, True)
setv("1")
s2 = torch.cuda.Stream()
outs = [torch.empty_like(q) for _ in range(4)]
ls = [torch.empty_like(l0) for _ in range(4)]
torch.cuda.synchronize()
for i in range(300):
    j = i & 3
    if j & 1:
        with torch.cuda.stream(s2):
            ops.attention_lse(q, k, vt, outs[j], ls[j], 8, 1091, 1091, 32, 32, 128, st, True)
    else:
        ops.attention_lse(q, k, vt, outs[j], ls[j], 8, 1091, 1091, 32, 32, 128, st, True)
torch.cuda.synchronize()
eq = all(torch.equal(x, o0) for x in outs) and all(torch.equal(x, l0) for x in ls)
ok &= eq
print(json.dumps(dict(check="300 launches on two streams", bit_identical=eq)), flush=True)

for (B, S, H, hd, causal, groups) in [(8, 1091, 32, 128, True, (None, 4, 8)), (4, 2048, 32, 128, True, (None, 4)), (8, 2048, 32, 128, True, (None,)),
                                      (8, 1967, 32, 128, True, (None,)), (8, 1091, 40, 128, True, (None,)), (40, 577, 16, 64, False, (None,))]:
    q, k, vt, st = mk(B, S, S, H, H, hd)
    o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev)
    variants = [("0", None)] + [("1", g) for g in groups]
    times = {v: [] for v in variants}
    for r in range(5):
        for v in variants:
            setv(*v)
            f = lambda: ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, causal)
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5 * 1e-3)
    fl = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    print(json.dumps(dict(B=B, S=S, H=H, hd=hd, causal=causal,
                          us={f"persist={v[0]},group={v[1]}": round(m * 1e6, 1) for v, m in med.items()},
                          tflops={f"persist={v[0]},group={v[1]}": round(fl / m / 1e12, 1) for v, m in med.items()})), flush=True)
print("PARITY", "ok" if ok else "FAILED")
