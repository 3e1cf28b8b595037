"""-m gpu: weight-only fp8 (OCP e4m3fn) decode path.  No reference oracle exists for fp8: the parity
statement is (i) the fp8 GEMV equals the bf16 arithmetic on the DEQUANTISED weights up to accumulation order, and (ii) the
quantised decode step stays within a stated, looser tolerance of the bf16 step (quantisation error of e4m3: 2^-4 relative
per weight, averaged down by the K-long dot products)."""
import math

import pytest
import torch
import torch.nn.functional as F

from a3vlm_amd import ops
from a3vlm_amd.model.LLM import llama_ens5 as plugin
from oracle.quant_fp8 import W8A8OracleDecoder, dequantize_rows_fp8, quantize_rows_fp8
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def rt(x):
    return x.to(BF).float()


def gen(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def assert_close(got, want, rtol, atol, what=""):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs()
    bad = err > atol + rtol * want.abs()
    if bad.any():
        i = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} out of tol; first at {i}: got {got[tuple(i)]:.6g} "
                             f"want {want[tuple(i)]:.6g}; max err {err.max():.4g}")


def test_quantizer_roundtrip():
    w = gen(300, 512, seed=1, scale=0.03)
    q, s = quantize_rows_fp8(w)
    assert q.dtype == torch.uint8 and q.shape == w.shape and s.shape == (300,)
    dq = dequantize_rows_fp8(q, s)
    assert float((dq - w).abs().max() / w.abs().max()) < 2 ** -4 + 1e-3          # e4m3: 3 mantissa bits
    assert float((dq.abs().amax(1) - w.abs().amax(1)).abs().max()) < 1e-6       # the row maximum is exactly representable


@pytest.mark.parametrize("M,N,K", [(8, 4096, 4096), (1, 256, 256), (16, 1000, 1024), (5, 12288, 4096), (8, 4096, 11008), (3, 64, 512)])
def test_gemm_skinny_fp8_vs_dequantised_bf16_math(M, N, K):
    a, w = rt(gen(M, K, seed=13)), gen(N, K, seed=14, scale=0.05)
    q, s = quantize_rows_fp8(w)
    wd = dequantize_rows_fp8(q, s)                       # what the kernel multiplies by (fp8 -> bf16 is exact, scale in fp32)
    ws = ops.gemm_skinny_workspace(M, N, K, DEV)
    ad, qd, sd = a.to(BF).to(DEV), q.to(DEV), s.to(DEV)
    want = (a @ q.view(torch.float8_e4m3fn).float().t()) * s[None, :]
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_skinny_fp8(ad, qd, sd, o32, ws, epilogue=ops.EPI_OUT_F32)
    err = float((o32.cpu() - rt(want)).abs().max() / want.abs().max())
    assert err < 2 ** -7, err
    res = rt(gen(M, N, seed=15))
    hb = res.to(BF).to(DEV).clone()
    ops.gemm_skinny_fp8(ad, qd, sd, hb, ws, residual=hb)
    assert float((hb.float().cpu() - rt(res + rt(want))).abs().max() / want.abs().max()) < 2 ** -6
    # and against the bf16 kernel on the dequantised weights rounded to bf16 (one extra weight rounding): loose
    ob = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm_skinny(ad, wd.to(BF).to(DEV), ob, ws)
    assert float((ob.float().cpu() - rt(want)).abs().max() / want.abs().max()) < 2e-2
    assert int(ws[:8192].view(torch.int32).abs().sum()) == 0


def test_gemm_skinny_fp8_swiglu():
    M, Fh, K = 8, 1024, 512
    a, w1, w3 = rt(gen(M, K, seed=20)), gen(Fh, K, seed=21, scale=0.05), gen(Fh, K, seed=22, scale=0.05)
    nb = Fh // 16
    w13 = torch.stack([w1.view(nb, 16, K), w3.view(nb, 16, K)], dim=1).reshape(2 * Fh, K)
    q, s = quantize_rows_fp8(w13)
    dq = dequantize_rows_fp8(q, s).view(nb, 2, 16, K)
    g, u = a @ dq[:, 0].reshape(Fh, K).t(), a @ dq[:, 1].reshape(Fh, K).t()
    want = rt(rt(F.silu(rt(g))) * rt(u))
    out = torch.empty(M, Fh, dtype=BF, device=DEV)
    ops.gemm_skinny_fp8(a.to(BF).to(DEV), q.to(DEV), s.to(DEV), out, ops.gemm_skinny_workspace(M, 2 * Fh, K, DEV), epilogue=ops.EPI_SWIGLU)
    assert float((out.float().cpu() - want).abs().max() / want.abs().max()) < 2 ** -5


@pytest.mark.parametrize("heads,kv,dim,B", [(4, 4, 512, 4), (8, 2, 1024, 8)])
def test_fp8_decode_step_vs_bf16(heads, kv, dim, B):
    args = plugin.ModelArgs(dim=dim, n_layers=3, n_heads=heads, n_kv_heads=kv, vocab_size=640, multiple_of=256, max_seq_len=192)
    oargs = ref_cpu.OracleArgs(dim=dim, n_layers=3, n_heads=heads, n_kv_heads=kv, vocab_size=640, multiple_of=256, max_seq_len=192)
    sd = ref_cpu.make_decoder_weights(oargs, seed=11, std=0.05)
    m = plugin.Transformer(args)
    m.load_state_dict(sd)
    m.to(BF).to(DEV)
    g = torch.Generator().manual_seed(2)
    T0, steps = 33, 5
    ex = torch.randint(3, 640, (B, T0 + steps), generator=g).to(DEV)
    ex[:, 0] = 1

    def run():
        lg = [m.forward_inference(ex[:, :T0], 0).float().clone()]
        for t in range(T0, T0 + steps):
            lg.append(m.forward_inference(ex[:, t:t + 1], t).float().clone())
        return lg
    base = run()
    m.quantize_decode_weights("fp8")
    q8 = run()
    assert torch.equal(base[0], q8[0])                                   # prefill does not use the fp8 images
    scale = max(float(b.abs().max()) for b in base)
    for i in range(1, steps + 1):
        err = float((base[i] - q8[i]).abs().max()) / scale
        assert 0 < err < 0.25, (i, err)                                  # really quantised; random N(0, 0.05) weights are a worst case for e4m3
    # oracle on the DEQUANTISED weights (what the fp8 step computes): tight(er) agreement
    from oracle.quant_fp8 import W8A8OracleDecoder, dequantize_rows_fp8, quantize_rows_fp8
    sdq = dict(sd)
    for k in list(sd):
        if k.startswith("layers.") and k.endswith((".wq.weight", ".wk.weight", ".wv.weight", ".wo.weight", ".w1.weight", ".w2.weight", ".w3.weight")):
            sdq[k] = dequantize_rows_fp8(*quantize_rows_fp8(sd[k].to(BF)))
    dec = ref_cpu.OracleDecoder(oargs, {k: v.to(BF) for k, v in sdq.items()})
    dec0 = ref_cpu.OracleDecoder(oargs, {k: v.to(BF) for k, v in sd.items()})
    dec0.forward_inference(ex[:, :T0].cpu(), 0)                          # bf16 prefill fills the cache, as on the GPU
    dec.k_cache, dec.v_cache, dec.cache_image_words = dec0.k_cache, dec0.v_cache, dec0.cache_image_words
    for i, t in enumerate(range(T0, T0 + steps)):
        want = dec.forward_inference(ex[:, t:t + 1].cpu(), t).float()
        assert float((q8[i + 1].cpu() - want).abs().max()) / scale < 5e-2, i
    m.quantize_decode_weights(None)
    again = run()
    assert torch.equal(again[1], base[1])


_W8A8Oracle = W8A8OracleDecoder.make(ref_cpu.OracleDecoder)


@pytest.mark.parametrize("heads,kv,dim,B,T0,L", [(4, 4, 512, 4, 33, 1), (8, 2, 1024, 3, 150, 1), (4, 4, 512, 4, 33, 3)])
def test_fp8_w8a8_prefill(heads, kv, dim, B, T0, L):
    """quantize_decode_weights("fp8", prefill=True): the multi-token forward runs every decoder GEMM on fp8 operands.
    (i) really quantised; (ii) explained by the W8A8 restatement of the oracle -- tightly for one block, and for three blocks
    (where random N(0, 0.05) weights, a worst case for e4m3, make the logits sums of amplified rounding noise) better than by
    the bf16 forward; (iii) the KV cache it leaves serves the following (weight-only fp8) decode steps; (iv) mode=None
    restores bf16.  The exact statements are at kernel level: test_gemm_nt_fp8 / test_quantize_rows_fp8."""
    args = plugin.ModelArgs(dim=dim, n_layers=L, n_heads=heads, n_kv_heads=kv, vocab_size=640, multiple_of=256, max_seq_len=256)
    oargs = ref_cpu.OracleArgs(dim=dim, n_layers=L, n_heads=heads, n_kv_heads=kv, vocab_size=640, multiple_of=256, max_seq_len=256)
    sd = ref_cpu.make_decoder_weights(oargs, seed=21, std=0.05)
    m = plugin.Transformer(args)
    m.load_state_dict(sd)
    m.to(BF).to(DEV)
    g = torch.Generator().manual_seed(4)
    ex = torch.randint(3, 640, (B, T0 + 2), generator=g)
    ex[:, 0] = 1
    exd = ex.to(DEV)
    base = m.forward_inference(exd[:, :T0], 0).float().clone()
    m.quantize_decode_weights("fp8", prefill=True)
    got = m.forward_inference(exd[:, :T0], 0).float().clone()
    nxt = [m.forward_inference(exd[:, t:t + 1], t).float().clone() for t in range(T0, T0 + 2)]
    def rms(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).norm() / b.norm())
    # random N(0, 0.05) weights are a worst case for e4m3 (logits are sums of noise): stated as relative RMS error
    e_bf = rms(got, base)
    dec = _W8A8Oracle(oargs, {k: v.to(BF) for k, v in sd.items()})
    want = dec.forward_inference(ex[:, :T0], 0).float()
    e_or, e_or_bf = rms(got, want), rms(want, base)
    print(f"w8a8 prefill: rms vs bf16 {e_bf:.4f}, vs W8A8 oracle {e_or:.4f} (oracle vs bf16 {e_or_bf:.4f})")
    assert 0 < e_bf < 0.4, e_bf
    assert e_or < 0.7 * e_bf and (L > 1 or e_or < 0.08), (e_or, e_bf)   # a bf16-level upstream difference flips ~6 % of the fp8 roundings
    for i, t in enumerate(range(T0, T0 + 2)):          # decode continues on the cache the fp8 prefill wrote
        w2 = dec.forward_inference(ex[:, t:t + 1], t).float()
        e_d = rms(nxt[i], w2)
        assert e_d < (0.12 if L == 1 else 0.25), (i, e_d)
    m.quantize_decode_weights(None)
    assert torch.equal(m.forward_inference(exd[:, :T0], 0).float(), base)


# ------------------------------------------------------------------ W8A8 prefill: activation quantiser + MX-scaled fp8 GEMM
def _deq(q, s):
    return q.cpu().view(torch.float8_e4m3fn).float() * s.cpu()[:, None]


@pytest.mark.parametrize("rows,dim,norm", [(37, 4096, False), (5, 11008, False), (64, 512, True), (19, 4096, True), (3, 264, False)])
def test_quantize_rows_fp8(rows, dim, norm):
    """scale = max|y| / 448 per row, q = fp8(y / scale) with y the bf16 rows or their bf16 RMSNorm (never stored): the
    dequantised value is within half an e4m3 step of y, the row maximum maps to +-448, a zero row stays finite."""
    g = torch.Generator().manual_seed(rows * 7 + dim)
    x = (torch.randn(rows, dim, generator=g) * torch.rand(rows, 1, generator=g) * 3).to(BF)
    x[rows // 2] = 0
    w = (1 + 0.1 * torch.randn(dim, generator=g)).to(BF)
    xd = x.to(DEV)
    y = x.float()
    if norm:
        yd = torch.empty_like(xd)
        ops.rmsnorm(xd, w.to(DEV), yd, 1e-5)
        y = yd.float().cpu()
    q = torch.zeros(rows, dim + 8, dtype=torch.uint8, device=DEV)
    s = torch.empty(rows, dtype=torch.float32, device=DEV)
    ops.quantize_rows_fp8(xd, q[:, :dim], s, w.to(DEV) if norm else None, 1e-5)
    assert int(q[:, dim:].sum()) == 0
    amax = y.abs().amax(dim=1)
    want_s = amax.clamp_min(1e-12) / 448
    assert torch.allclose(s.cpu(), want_s, rtol=1e-6, atol=0)
    d = _deq(q[:, :dim], s)
    assert torch.isfinite(d).all()
    step = torch.maximum(y.abs() * 2 ** -4, want_s[:, None] * 2 ** -10)     # half ulp of 3 mantissa bits / half a subnormal step
    assert bool(((d - y).abs() <= step * 1.001).all())
    nz = amax > 0
    assert torch.allclose(d.abs().amax(dim=1)[nz], amax[nz], rtol=1e-6, atol=0)


@pytest.mark.parametrize("M,N,K,epi", [(300, 520, 256, "plain"), (1091, 768, 1024, "residual"), (512, 1024, 128, "swiglu"),
                                       (260, 256, 384, "f32"), (8728, 512, 512, "plain")])
def test_gemm_nt_fp8(M, N, K, epi):
    """MX-scaled fp8 MFMA GEMM == the fp32 product of the dequantised operands (fp8 x fp8 products are exact in fp32), then
    the same rounding points as the bf16 kernel's epilogues; partial tiles in M and N, one to eight k-tiles."""
    from oracle.quant_fp8 import quantize_rows_fp8 as qhost
    a, w = gen(M, K, seed=71), gen(N, K, seed=72, scale=0.05)
    aq, sa = qhost(a)
    wq, sw = qhost(w)
    want = (aq.view(torch.float8_e4m3fn).float() * sa[:, None]) @ (wq.view(torch.float8_e4m3fn).float() * sw[:, None]).t()
    aqd, sad, wqd, swd = aq.to(DEV), sa.to(DEV), wq.to(DEV), sw.to(DEV)
    tol = dict(rtol=2 ** -7, atol=1e-3 * math.sqrt(K) * 0.05 + 1e-3)
    if epi == "plain":
        out = torch.empty(M, N, dtype=BF, device=DEV)
        ops.gemm_nt_fp8(aqd, sad, wqd, swd, out)
        assert_close(out, want, what="fp8 plain", **tol)
        out2 = torch.empty_like(out)
        ops.gemm_nt_fp8(aqd, sad, wqd, swd, out2)
        assert torch.equal(out, out2)
    elif epi == "residual":
        res = rt(gen(M, N, seed=73))
        out = res.to(BF).to(DEV)
        ops.gemm_nt_fp8(aqd, sad, wqd, swd, out, residual=out)
        # the product is rounded to bf16 before the add: a rounding flip is one bf16 step of the PRODUCT, whatever the sum is
        assert_close(out, res + rt(want), what="fp8 residual", rtol=2 ** -7, atol=tol["atol"] + 2 ** -8 * float(want.abs().max()))
    elif epi == "f32":
        res = gen(M, N, seed=74)
        out = res.to(DEV).clone()
        ops.gemm_nt_fp8(aqd, sad, wqd, swd, out, residual=out, epilogue=ops.EPI_RES_F32)
        assert_close(out.cpu() - res, want, what="fp8 f32 residual", **tol)
    else:
        # rows interleaved in blocks of 16: even block = gate, odd = up (A3V_EPI_SWIGLU)
        out = torch.empty(M, N // 2, dtype=BF, device=DEV)
        ops.gemm_nt_fp8(aqd, sad, wqd, swd, out, epilogue=ops.EPI_SWIGLU)
        wb = rt(want).view(M, N // 32, 2, 16)
        gate, up = wb[:, :, 0].reshape(M, N // 2), wb[:, :, 1].reshape(M, N // 2)
        assert_close(out, rt(torch.nn.functional.silu(gate)) * up, what="fp8 swiglu", rtol=2 ** -6, atol=tol["atol"])


@pytest.mark.parametrize("M,N,K,mode", [(4452, 4096, 2048, "residual"), (4452, 4096, 4096, "f32"), (8728, 4096, 1024, "plain")])
def test_gemm_nt_fp8_split_tail(M, N, K, mode):
    """18 x 16 (35 x 16) tiles on 256 CUs: the tile rows that fill whole rounds run as one launch, the remaining rows as split-K
    planes + the reduce epilogue -- same values as the un-split kernel up to fp32 summation order before the bf16 rounding.
    Reference: fp32 torch matmul of the dequantised operands on the device (exact products)."""
    from oracle.quant_fp8 import quantize_rows_fp8 as qhost
    a, w = gen(M, K, seed=81), gen(N, K, seed=82, scale=0.05)
    aq, sa = qhost(a)
    wq, sw = qhost(w)
    aqd, sad, wqd, swd = aq.to(DEV), sa.to(DEV), wq.to(DEV), sw.to(DEV)
    want = ((aqd.view(torch.float8_e4m3fn).float() * sad[:, None]) @ (wqd.view(torch.float8_e4m3fn).float() * swd[:, None]).t()).cpu()
    atol = 1e-3 * math.sqrt(K) * 0.05 + 1e-3 + 2 ** -8 * float(want.abs().max())
    if mode == "plain":
        out = torch.empty(M, N, dtype=BF, device=DEV)
        ops.gemm_nt_fp8(aqd, sad, wqd, swd, out)
        assert_close(out, want, rtol=2 ** -7, atol=atol, what="fp8 plain (rows in rounds + tail)")
    elif mode == "residual":
        res = rt(gen(M, N, seed=83))
        out = res.to(BF).to(DEV)
        ops.gemm_nt_fp8(aqd, sad, wqd, swd, out, residual=out)
        assert_close(out, res + rt(want), rtol=2 ** -7, atol=atol, what="fp8 residual split tail")
    else:
        res = gen(M, N, seed=84)
        out = res.to(DEV).clone()
        ops.gemm_nt_fp8(aqd, sad, wqd, swd, out, residual=out, epilogue=ops.EPI_RES_F32)
        assert_close(out.cpu() - res, want, rtol=2 ** -7, atol=atol, what="fp8 f32 residual split tail")


@pytest.mark.parametrize("M,N,K,exact", [(8728, 4096, 4096, False), (8728, 4096, 11008, False), (8728, 1280, 384, True), (4452, 4096, 2048, False),
                                         (1000, 776, 1152, True), (2048, 2048, 640, True)])
def test_gemm_nt_fp8_ring_forms_equal_the_two_stage_kernel(M, N, K, exact):
    """Round 5: the fp8 product runs on the ring kernel (persistent tile walk, three LDS rings, DMA stream across tiles, scales applied in
    front of the staged epilogues) on 256 x 256 tiles or, where whole rounds + a split-K tail cost more, on 192 x 256 tiles.  Same MFMA,
    same K order, same scale expression, epilogues that round identically: where the two-stage kernel runs un-split (`exact`: at most one
    round of tiles) the ring forms equal it BIT FOR BIT for plain / residual / fp32-stream / SwiGLU outputs, twice in a row; where it takes
    its split-K tail (another summation order) they stay within one bf16 step of it."""
    from a3vlm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(9)
    aq = torch.randint(0, 256, (M, K), device=DEV, generator=g, dtype=torch.uint8)
    wq = torch.randint(0, 256, (N, K), device=DEV, generator=g, dtype=torch.uint8)
    aq[(aq & 0x7f) >= 0x60] &= 0x9f      # |x| < 2: no NaN encodings, sums stay small
    wq[(wq & 0x7f) >= 0x60] &= 0x9f
    sa = torch.rand(M, device=DEV, generator=g) * 1e-2 + 1e-3
    sw = torch.rand(N, device=DEV, generator=g) * 1e-2 + 1e-3
    res_b = (torch.randn(M, N, device=DEV, generator=g)).to(BF)
    res_f = torch.randn(M, N, device=DEV, generator=g)

    def run(kind):
        if kind == "plain":
            o = torch.full((M, N), 3.0, dtype=BF, device=DEV)
            return ops.gemm_nt_fp8(aq, sa, wq, sw, o)
        if kind == "residual":
            o = res_b.clone()
            return ops.gemm_nt_fp8(aq, sa, wq, sw, o, residual=o)
        if kind == "res_f32":
            o = res_f.clone()
            return ops.gemm_nt_fp8(aq, sa, wq, sw, o, residual=o, epilogue=ops.EPI_RES_F32)
        o = torch.empty(M, N // 2, dtype=BF, device=DEV)
        return ops.gemm_nt_fp8(aq, sa, wq, sw, o, epilogue=ops.EPI_SWIGLU)

    kinds = ["plain", "residual", "res_f32"] + (["swiglu"] if N % 32 == 0 else [])
    for kind in kinds:
        res = {"residual": res_b, "res_f32": res_f}.get(kind)
        with lib.env(A3V_GEMM_FP8_RING="0"):
            want = run(kind)                                   # the two-stage kernel
        with lib.env(A3V_GEMM_FP8_192="2"):
            for rep in range(2):
                got = run(kind)
                assert (torch.equal(got, want) if exact else _within_tail_noise(got, want, res)), (kind, rep, int((got != want).sum()))
        got = run(kind)                                        # the default dispatch
        assert (torch.equal(got, want) if exact else _within_tail_noise(got, want, res)), kind


def _within_tail_noise(got, want, res=None):
    """one bf16 step of the PRODUCT (the residual is added after the product is rounded: |product| <= |sum| + |residual|)"""
    d = (got.float() - want.float()).abs()
    mag = want.float().abs() + (res.float().abs() if res is not None else 0.0)
    return bool((d <= 2 ** -7 * mag + 1e-6 * float(want.float().abs().max()) + 1e-30).all())


def test_gemm_nt_fp8_7b_size_scale_homogeneity():
    """BASELINE-size property (wo of Llama-2-7B at 8728 tokens: rows in rounds + split-K tail): doubling the activation scales
    doubles every output exactly (the scales enter once, in the epilogue, before the single bf16 rounding)."""
    T, N, K = 8728, 4096, 4096
    g = torch.Generator(device=DEV).manual_seed(6)
    aq = torch.randint(0, 256, (T, K), device=DEV, generator=g, dtype=torch.uint8)
    wq = torch.randint(0, 256, (N, K), device=DEV, generator=g, dtype=torch.uint8)
    aq[(aq & 0x7f) == 0x7f] = 0x38       # no NaN encodings (e4m3fn: S.1111.111)
    wq[(wq & 0x7f) == 0x7f] = 0x38
    sa = torch.rand(T, device=DEV, generator=g) * 1e-3 + 1e-4
    sw = torch.rand(N, device=DEV, generator=g) * 1e-3 + 1e-4
    o1 = torch.empty(T, N, dtype=BF, device=DEV)
    o2 = torch.empty_like(o1)
    ops.gemm_nt_fp8(aq, sa, wq, sw, o1)
    ops.gemm_nt_fp8(aq, sa * 2, wq, sw, o2)
    assert torch.isfinite(o1.float()).all() and float(o1.float().abs().max()) > 0
    assert torch.equal(o2.float(), o1.float() * 2)
