"""-m gpu: every HIP kernel, called through the C-ABI, against the CPU oracle ops on the same
seeded inputs.  fp32 path: fp32 round-off tolerances.  bf16 path: the oracle is evaluated on the
same bf16-rounded inputs with fp32 accumulation and the comparison allows 1-2 bf16 ulps
(rel 2^-8 .. 2^-7) -- the tolerance is written next to each check."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from a3vlm_amd import ops  # noqa: E402
from oracle import ref_cpu  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def rt(x):  # bf16 round trip in fp32
    return x.to(BF).float()


def assert_close(got, want, rtol, atol, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = err > bound
    if bad.any():
        i = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} out of tol; first at {i}: got {got[tuple(i)]:.6g} "
                             f"want {want[tuple(i)]:.6g}; max err {err.max():.4g}")


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (100, 132, 192), (8728 // 8, 256, 4096),
                                   (1, 128, 64), (300, 4096, 1024), (129, 32000 // 10 // 4 * 4, 256)])
def test_gemm_bf16_plain(M, N, K):
    a, w = rt(gen(M, K, seed=1)), rt(gen(N, K, seed=2, scale=0.05))
    out = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm_nt(a.to(BF).to(DEV), w.to(BF).to(DEV), out)
    want = a @ w.t()
    assert_close(out, want, rtol=2 ** -7, atol=1e-3 * math.sqrt(K) * 0.05, what=f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(300, 520, 128), (1000, 1300, 256), (257, 4096, 64), (2048, 2048, 1024), (515, 777 // 4 * 4, 192)])
def test_gemm_bf16_tile_configs_agree(M, N, K):
    """128x128 and 256x256 tile kernels accumulate every output in the same k order: bit-identical."""
    from a3vlm_amd import lib
    a, w = gen(M, K, seed=60).to(BF).to(DEV), (gen(N, K, seed=61, scale=0.05)).to(BF).to(DEV)
    bias = gen(N, seed=62).to(BF).to(DEV)
    o1 = torch.empty(M, N, dtype=BF, device=DEV)
    o2 = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm_nt(a, w, o1, bias=bias, epilogue=ops.EPI_GELU | lib.EPI_TILE_128)
    ops.gemm_nt(a, w, o2, bias=bias, epilogue=ops.EPI_GELU | lib.EPI_TILE_256)
    assert torch.equal(o1, o2)
    want = rt(F.gelu(rt(a.float().cpu() @ w.float().cpu().t() + bias.float().cpu())))
    o3 = torch.empty(M, N, dtype=BF, device=DEV)
    for _ in range(3):   # ping-pong kernel: repeat to give a race a chance to show
        o3.zero_()
        ops.gemm_nt(a, w, o3, bias=bias, epilogue=ops.EPI_GELU | lib.EPI_TILE_256PP)
        assert torch.equal(o1, o3)
        if lib.has_experiments():      # (`make EXPERIMENTS=1` builds only)
            o3.zero_()   # 32x32x16 MFMA form: different in-instruction summation order -> tolerance, not bits
            ops.gemm_nt(a, w, o3, bias=bias, epilogue=ops.EPI_GELU | lib.EPI_TILE_256PP32)
            assert_close(o3, want, rtol=2 ** -6, atol=4e-3, what="gemm pp32")
    want = rt(F.gelu(rt(a.float().cpu() @ w.float().cpu().t() + bias.float().cpu())))
    assert_close(o2, want, rtol=2 ** -6, atol=4e-3, what="gemm 256 tile")


def test_gemm_bf16_transpose_detecting():
    """A = [I | 0] against an ASYMMETRIC W: catches swapped row/col in the MFMA C-write."""
    M, N, K = 128, 256, 128
    a = torch.zeros(M, K)
    a[:, :M] = torch.eye(M)
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125
    out = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm_nt(a.to(BF).to(DEV), w.to(BF).to(DEV), out)
    want = w[:, :M].t()
    assert torch.equal(out.float().cpu(), rt(want))


def test_gemm_bf16_strided_operands():
    M, N, K = 200, 192, 128
    abig, wbig = rt(gen(M, K + 64, seed=3)), rt(gen(N, K + 128, seed=4, scale=0.1))
    obig = torch.zeros(M, N + 64, dtype=BF, device=DEV)
    ad, wd = abig.to(BF).to(DEV), wbig.to(BF).to(DEV)
    ops.gemm_nt(ad[:, 64:], wd[:, 128:], obig[:, 32:32 + N])
    want = abig[:, 64:] @ wbig[:, 128:].t()
    assert_close(obig[:, 32:32 + N], want, rtol=2 ** -7, atol=5e-3, what="strided gemm")
    assert float(obig[:, :32].abs().sum()) == 0 and float(obig[:, 32 + N:].abs().sum()) == 0


def test_gemm_bf16_epilogues():
    M, N, K = 150, 256, 128
    a, w = rt(gen(M, K, seed=5)), rt(gen(N, K, seed=6, scale=0.08))
    bias, res = rt(gen(N, seed=7)), rt(gen(M, N, seed=8))
    ad, wd = a.to(BF).to(DEV), w.to(BF).to(DEV)
    lin = a @ w.t()
    # bias + erf GELU: F.linear -> bf16, gelu -> bf16
    out = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm_nt(ad, wd, out, bias=bias.to(BF).to(DEV), epilogue=ops.EPI_GELU)
    want = rt(F.gelu(rt(lin + bias)))
    assert_close(out, want, rtol=2 ** -6, atol=4e-3, what="bias+gelu")
    ops.gemm_nt(ad, wd, out, bias=bias.to(BF).to(DEV), epilogue=ops.EPI_QUICKGELU)
    y = rt(lin + bias)
    assert_close(out, rt(y * torch.sigmoid(1.702 * y)), rtol=2 ** -6, atol=4e-3, what="bias+quickgelu")
    # bias + residual, in place on the residual buffer (x = x + linear(...))
    hbuf = res.to(BF).to(DEV).clone()
    ops.gemm_nt(ad, wd, hbuf, bias=bias.to(BF).to(DEV), residual=hbuf)
    assert_close(hbuf, rt(res + rt(lin + bias)), rtol=2 ** -6, atol=8e-3, what="bias+residual in place")
    # fp32 output of bf16-rounded logits
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_nt(ad, wd, o32, epilogue=ops.EPI_OUT_F32)
    assert_close(o32, rt(lin), rtol=2 ** -7, atol=4e-3, what="out f32")
    assert torch.equal(o32.cpu(), rt(o32.cpu())), "fp32 logits must be bf16-representable (output(h).float())"
    # fp32 residual stream
    r32 = gen(M, N, seed=9).to(DEV)
    o32b = torch.empty_like(r32)
    ops.gemm_nt(ad, wd, o32b, residual=r32, epilogue=ops.EPI_RES_F32)
    assert_close(o32b, r32.cpu() + rt(lin), rtol=2 ** -7, atol=4e-3, what="res f32")


def pack_w13(w1, w3):
    nb = w1.shape[0] // 16
    return torch.stack([w1.view(nb, 16, -1), w3.view(nb, 16, -1)], dim=1).reshape(2 * w1.shape[0], -1).contiguous()


@pytest.mark.parametrize("M", [4, 150])
def test_gemm_bf16_swiglu(M):
    F_, K = 192, 128
    a, w1, w3 = rt(gen(M, K, seed=10)), rt(gen(F_, K, seed=11, scale=0.1)), rt(gen(F_, K, seed=12, scale=0.1))
    w13 = pack_w13(w1, w3).to(BF).to(DEV)
    out = torch.empty(M, F_, dtype=BF, device=DEV)
    ops.gemm_nt(a.to(BF).to(DEV), w13, out, epilogue=ops.EPI_SWIGLU)
    want = rt(rt(F.silu(rt(a @ w1.t()))) * rt(a @ w3.t()))
    assert_close(out, want, rtol=2 ** -6, atol=4e-3, what="swiglu")
    if M <= 16:
        part = ops.gemm_skinny_workspace(M, 2 * F_, K, DEV)
        out2 = torch.empty(M, F_, dtype=BF, device=DEV)
        ops.gemm_skinny(a.to(BF).to(DEV), w13, out2, part, epilogue=ops.EPI_SWIGLU)
        assert_close(out2, want, rtol=2 ** -6, atol=4e-3, what="skinny swiglu")


@pytest.mark.parametrize("M,N,K", [(1, 256, 128), (8, 4096, 4096), (16, 1000, 1376), (5, 32000, 512), (8, 12288, 4096),
                                   (16, 1000, 1024), (9, 4096, 4096), (8, 4096, 11008), (2, 132, 256), (7, 64, 2048)])
def test_gemm_skinny(M, N, K):
    """K % 128 == 0 -> LDS-DMA split-K GEMV (uneven slices at K = 11008, 16-row A image at M > 8, ragged N);
    otherwise the direct-to-VGPR kernel.  The workspace counters are left zero, so it is reused across calls."""
    a, w = rt(gen(M, K, seed=13)), rt(gen(N, K, seed=14, scale=0.05))
    res = rt(gen(M, N, seed=15))
    ad, wd = a.to(BF).to(DEV), w.to(BF).to(DEV)
    part = ops.gemm_skinny_workspace(M, N, K, DEV)
    lin = a @ w.t()
    out = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm_skinny(ad, wd, out, part)
    assert_close(out, lin, rtol=2 ** -7, atol=2e-3 * math.sqrt(K) * 0.05 + 1e-3, what="skinny")
    hb = res.to(BF).to(DEV).clone()
    ops.gemm_skinny(ad, wd, hb, part, residual=hb)
    assert_close(hb, rt(res + rt(lin)), rtol=2 ** -6, atol=1e-2, what="skinny residual")
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_skinny(ad, wd, o32, part, epilogue=ops.EPI_OUT_F32)
    assert_close(o32, rt(lin), rtol=2 ** -7, atol=2e-3 * math.sqrt(K) * 0.05 + 1e-3, what="skinny f32 out")
    # split-K partials are summed in slice order by whichever block arrives last: bit-identical across runs
    o32b = torch.empty_like(o32)
    for _ in range(3):
        ops.gemm_skinny(ad, wd, o32b, part, epilogue=ops.EPI_OUT_F32)
        assert torch.equal(o32, o32b)
    assert int(part[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("M,F_,K", [(8, 11008, 4096), (3, 96, 512), (16, 2048, 1024)])
def test_gemm_skinny_swiglu_large(M, F_, K):
    a, w1, w3 = rt(gen(M, K, seed=20)), rt(gen(F_, K, seed=21, scale=0.03)), rt(gen(F_, K, seed=22, scale=0.03))
    w13 = pack_w13(w1, w3).to(BF).to(DEV)
    part = ops.gemm_skinny_workspace(M, 2 * F_, K, DEV)
    out = torch.empty(M, F_, dtype=BF, device=DEV)
    ops.gemm_skinny(a.to(BF).to(DEV), w13, out, part, epilogue=ops.EPI_SWIGLU)
    want = rt(rt(F.silu(rt(a @ w1.t()))) * rt(a @ w3.t()))
    assert_close(out, want, rtol=2 ** -6, atol=4e-3, what="skinny swiglu large")


@pytest.mark.parametrize("M,N,K", [(70, 96, 64), (64, 64, 16), (130, 200, 640)])
def test_gemm_f32(M, N, K):
    a, w, bias, res = gen(M, K, seed=16), gen(N, K, seed=17, scale=0.1), gen(N, seed=18), gen(M, N, seed=19)
    out = torch.empty(M, N, device=DEV)
    ops.gemm_nt(a.to(DEV), w.to(DEV), out, bias=bias.to(DEV), residual=res.to(DEV), epilogue=ops.EPI_GELU)
    want = res + F.gelu(a @ w.t() + bias)
    assert_close(out, want, rtol=1e-5, atol=2e-5, what="gemm f32")
    if N % 32 == 0:
        w1, w3 = w[: N // 2], w[N // 2:]
        o2 = torch.empty(M, N // 2, device=DEV)
        ops.gemm_nt(a.to(DEV), pack_w13(w1, w3).to(DEV), o2, epilogue=ops.EPI_SWIGLU)
        assert_close(o2, F.silu(a @ w1.t()) * (a @ w3.t()), rtol=1e-5, atol=2e-5, what="swiglu f32")


def test_gemm_rejects_bad_shapes():
    a = torch.zeros(4, 40, dtype=BF, device=DEV)
    w = torch.zeros(8, 40, dtype=BF, device=DEV)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(a, w, torch.empty(4, 8, dtype=BF, device=DEV))


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("dim", [64, 1024, 4096, 5120])
def test_rmsnorm(dim):
    x, w = gen(37, dim, seed=20, scale=2.0), 1 + 0.1 * gen(dim, seed=21)
    out = torch.empty(37, dim, device=DEV)
    ops.rmsnorm(x.to(DEV), w.to(DEV), out, 1e-5)
    assert_close(out, ref_cpu.rmsnorm(x, w, 1e-5), rtol=1e-5, atol=1e-6, what="rmsnorm f32")
    xb, wb = x.to(BF), w.to(BF)
    ob = torch.empty(37, dim, dtype=BF, device=DEV)
    ops.rmsnorm(xb.to(DEV), wb.to(DEV), ob, 1e-5)
    want = ref_cpu.rmsnorm(xb, wb, 1e-5)   # bf16 semantic of the reference (two roundings)
    mism = (ob.cpu() != want).float().mean().item()
    assert mism < 2e-3, f"bf16 rmsnorm: {mism:.2%} elements differ (fp32 reduction order only)"
    assert_close(ob, want, rtol=2 ** -7, atol=1e-6, what="rmsnorm bf16")
    # fp32 residual stream -> bf16 activations (autocast training)
    o2 = torch.empty(37, dim, dtype=BF, device=DEV)
    ops.rmsnorm(x.to(DEV), w.to(DEV), o2, 1e-5)
    assert_close(o2, ref_cpu.rmsnorm(x, w, 1e-5).to(BF), rtol=2 ** -7, atol=1e-6, what="rmsnorm f32->bf16")


def test_rmsnorm_strided_rows():
    B, S, dim = 3, 5, 128
    h = gen(B * S, dim, seed=22)
    hd_ = h.to(DEV)
    last = hd_.view(B, S, dim)[:, -1, :]
    out = torch.empty(B, dim, device=DEV)
    ops.rmsnorm(last, torch.ones(dim, device=DEV), out, 1e-5)
    assert_close(out, ref_cpu.rmsnorm(h.view(B, S, dim)[:, -1], torch.ones(dim), 1e-5), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dim,dtype", [(64, torch.float32), (1024, BF), (4096, BF), (5120, torch.float32)])
def test_layernorm_and_row_map(dim, dtype):
    rows = 29
    x, w, b = gen(rows, dim, seed=23, scale=3.0), 1 + 0.1 * gen(dim, seed=24), 0.1 * gen(dim, seed=25)
    xd, wd, bd = x.to(dtype).to(DEV), w.to(dtype).to(DEV), b.to(dtype).to(DEV)
    want = F.layer_norm(x.to(dtype).float(), (dim,), w.to(dtype).float(), b.to(dtype).float(), 1e-5)
    out = torch.empty(rows, dim, dtype=dtype, device=DEV)
    ops.layernorm(xd, wd, bd, out)
    tol = dict(rtol=1e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=2e-3)
    assert_close(out, want, what="layernorm", **tol)
    perm = torch.randperm(rows * 2, generator=torch.Generator().manual_seed(1))[:rows].to(torch.int32)
    big = torch.zeros(rows * 2, dim, dtype=dtype, device=DEV)
    ops.layernorm(xd, wd, bd, big, row_map=perm.to(DEV))
    assert_close(big[perm.long().to(DEV)], want, what="layernorm row_map", **tol)
    untouched = torch.ones(rows * 2, dtype=torch.bool)
    untouched[perm.long()] = False
    assert float(big[untouched.to(DEV)].abs().sum()) == 0
    xd2 = xd.clone()
    ops.layernorm(xd2, wd, bd, xd2)   # in place
    assert_close(xd2, want, what="layernorm in place", **tol)


# ------------------------------------------------------------------ RoPE + KV cache
@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("B,S,H,Hkv,hd,start", [(2, 5, 4, 2, 16, 0), (2, 1, 4, 2, 16, 7), (1, 130, 2, 2, 128, 3), (2, 70, 4, 4, 64, 64)])
def test_rope_kvcache(dtype, B, S, H, Hkv, hd, start):
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    Smax = 256
    width = (H + 2 * Hkv) * hd
    qkv = gen(B * S, width, seed=26).to(dtype)
    cs = precompute_cos_sin(hd, 512, 10000.0, None)
    fc = ref_cpu.precompute_freqs_cis(hd, 512)
    assert torch.equal(cs[..., 0], fc.real) and torch.equal(cs[..., 1], fc.imag)
    q = qkv[:, :H * hd].view(B, S, H, hd)
    k = qkv[:, H * hd:(H + Hkv) * hd].view(B, S, Hkv, hd)
    v = qkv[:, (H + Hkv) * hd:].view(B, S, Hkv, hd)
    oq, ok = ref_cpu.apply_rotary_emb(q, k, fc[start:start + S])
    qkv_d = qkv.to(DEV).clone()
    kc = torch.full((B, Hkv, Smax, hd), 7.0, dtype=dtype, device=DEV)
    vc = torch.full((B, Hkv, hd, Smax), 7.0, dtype=dtype, device=DEV)
    ops.rope_kvcache(qkv_d, qkv_d, kc, vc, cs.to(DEV), B, S, H, Hkv, hd, start, start)
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=2 ** -8, atol=1e-6)
    assert_close(qkv_d[:, :H * hd].view(B, S, H, hd), oq, what="rope q (in place)", **tol)
    assert_close(kc[:, :, start:start + S].permute(0, 2, 1, 3), ok, what="k cache", **tol)
    assert torch.equal(vc[:, :, :, start:start + S].permute(0, 3, 1, 2).cpu(), v), "v^T cache must be an exact copy"
    assert float((kc[:, :, :start] - 7).abs().sum()) == 0 and float((kc[:, :, start + S:] - 7).abs().sum()) == 0
    assert float((vc[..., :start] - 7).abs().sum()) == 0 and float((vc[..., start + S:] - 7).abs().sum()) == 0


# ------------------------------------------------------------------ attention
def oracle_attn(q, k, v, causal):
    """q [B,Sq,H,hd], k/v [B,Sk,Hkv,hd] -> [B,Sq,H,hd] via ref_cpu.sdpa + right-aligned mask."""
    n_rep = q.shape[2] // k.shape[2]
    kk = ref_cpu.repeat_kv(k, n_rep).transpose(1, 2)
    vv = ref_cpu.repeat_kv(v, n_rep).transpose(1, 2)
    m = ref_cpu.make_causal_mask(q.shape[1], k.shape[1]) if causal else None
    return ref_cpu.sdpa(q.transpose(1, 2), kk, vv, m).transpose(1, 2)


def run_attn(q, k, v, causal, dtype, Smax=None):
    B, Sq, H, hd = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    Smax = Smax or (Sk + 63) // 64 * 64
    qd = q.to(dtype).to(DEV).contiguous()
    kc = torch.zeros(B, Hkv, Smax, hd, dtype=dtype, device=DEV)
    vc = torch.full((B, Hkv, hd, Smax), float("nan"), dtype=dtype, device=DEV)   # poison beyond Sk
    kc[:, :, :Sk] = k.to(dtype).permute(0, 2, 1, 3).to(DEV)
    vc[:, :, :, :Sk] = v.to(dtype).permute(0, 2, 3, 1).to(DEV)
    out = torch.empty(B, Sq, H, hd, dtype=dtype, device=DEV)
    strides = (Sq * H * hd, H * hd, hd, Hkv * Smax * hd, Smax * hd, hd, Hkv * hd * Smax, hd * Smax, Smax, Sq * H * hd, H * hd, hd)
    scratch = None
    if Sq == 1 and dtype == BF:
        scratch = torch.empty(ops.attention_scratch_floats(B, H, hd, Sk), dtype=torch.float32, device=DEV)
    ops.attention(qd, kc, vc, out, B, Sq, Sk, H, Hkv, hd, strides, causal, scratch)
    return out


@pytest.mark.parametrize("B,Sq,Sk,H,Hkv,hd,causal", [
    (2, 6, 6, 4, 2, 16, True), (2, 1, 6, 4, 2, 16, False), (2, 2, 5, 4, 2, 16, True),
    (1, 70, 70, 2, 1, 128, True), (2, 33, 97, 2, 2, 64, False)])
def test_attention_f32(B, Sq, Sk, H, Hkv, hd, causal):
    q, k, v = gen(B, Sq, H, hd, seed=30), gen(B, Sk, Hkv, hd, seed=31), gen(B, Sk, Hkv, hd, seed=32)
    out = run_attn(q, k, v, causal, torch.float32)
    assert_close(out, oracle_attn(q, k, v, causal), rtol=1e-4, atol=2e-5, what="attn f32")


@pytest.mark.parametrize("B,Sq,Sk,H,Hkv,hd,causal", [
    (1, 128, 128, 2, 2, 128, True), (2, 300, 300, 4, 2, 128, True), (1, 100, 427, 2, 1, 128, True),
    (2, 577, 577, 4, 4, 64, False), (1, 257, 257, 2, 2, 64, False), (1, 64, 64, 1, 1, 64, True),
    (1, 1091, 1091, 2, 2, 128, True), (3, 50, 50, 2, 2, 128, False)])
def test_attention_prefill_bf16(B, Sq, Sk, H, Hkv, hd, causal):
    q, k, v = rt(gen(B, Sq, H, hd, seed=33)), rt(gen(B, Sk, Hkv, hd, seed=34)), rt(gen(B, Sk, Hkv, hd, seed=35))
    out = run_attn(q, k, v, causal, BF)
    want = oracle_attn(q, k, v, causal)
    # P is rounded to bf16 before P.V (as flash-attn / bf16 SDPA do): abs error ~ 2^-9 * |v|max
    assert_close(out, want, rtol=2 ** -6, atol=1.5e-2, what="attn prefill bf16")
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("lazy", ["1", "0"])
@pytest.mark.parametrize("shape", ["spike", "ramp"])
def test_attention_prefill_bf16_spike_rescale(shape, lazy):
    """Forces the online-softmax rescale, with the lazy reference (moves only when an exponent would pass 2^8) and with the rescale
    on every tile (A3V_ATTN_LAZY=0).  spike: one key row dominates late in the sequence (one big jump).  ramp: the scores grow a
    little with every KV tile, so the lazy form keeps a stale reference for a few tiles (P up to 2^8) and then moves it."""
    import os
    B, S, H, hd = 1, 512, 1, 128
    q, k, v = rt(gen(B, S, H, hd, seed=36)), rt(gen(B, S, H, hd, seed=37)), rt(gen(B, S, H, hd, seed=38))
    if shape == "spike":
        k[0, 200, 0] = q[0, 220, 0] * 4.0     # query 220+ see a huge score at key 200 (tile 3)
    else:
        u = torch.nn.functional.normalize(gen(hd, seed=39), dim=0)
        q[0, :, 0] += 6.0 * u                 # every query has a component along u ...
        k[0, :, 0] += rt(torch.arange(S)[:, None] / 64.0 * 0.8 * u[None, :])      # ... and the keys' grows by 0.8 per tile: +~3 in log2 per tile
        q, k = rt(q), rt(k)
    from a3vlm_amd import lib
    with lib.env(A3V_ATTN_LAZY=lazy):
        out = run_attn(q, k, v, True, BF)
    assert torch.isfinite(out.float()).all()
    assert_close(out, oracle_attn(q, k, v, True), rtol=2 ** -6, atol=1.5e-2, what=f"attn {shape} lazy={lazy}")


@pytest.mark.parametrize("form", ["two_phase", "wave"])
@pytest.mark.parametrize("B,Sk,H,Hkv,hd", [(2, 6, 4, 2, 64), (8, 1091, 4, 4, 128), (1, 4000, 2, 1, 128), (3, 129, 2, 2, 128), (40, 700, 8, 8, 128),
                                           (2, 513, 2, 2, 64)])
def test_attention_decode_bf16(B, Sk, H, Hkv, hd, form):
    """Both decode-attention kernels (the wave-streaming form is the decode step's default; here it is reached through the
    tuning switch of the stand-alone entry).  The V^T cache beyond Sk is NaN: a kernel that lets a masked column touch the sum fails."""
    import os
    q, k, v = rt(gen(B, 1, H, hd, seed=39)), rt(gen(B, Sk, Hkv, hd, seed=40)), rt(gen(B, Sk, Hkv, hd, seed=41))
    from a3vlm_amd import lib
    with lib.env(A3V_ATTN_DECODE_WAVE_STANDALONE="1" if form == "wave" else "0"):
        out = run_attn(q, k, v, False, BF, Smax=4096)
    assert_close(out, oracle_attn(q, k, v, False), rtol=2 ** -7, atol=4e-3, what=f"attn decode bf16 ({form})")


def test_attention_decode_wave_form_spike_rescale():
    """Online softmax across a wave's tiles and across the eight waves: one late key dominates."""
    import os
    B, Sk, H, hd = 2, 1500, 2, 128
    q, k, v = rt(gen(B, 1, H, hd, seed=43)), rt(gen(B, Sk, H, hd, seed=44)), rt(gen(B, Sk, H, hd, seed=45))
    k[:, 1400] = q[:, 0] * 3.0
    from a3vlm_amd import lib
    with lib.env(A3V_ATTN_DECODE_WAVE_STANDALONE="1"):
        out = run_attn(q, k, v, False, BF, Smax=2048)
    assert_close(out, oracle_attn(q, k, v, False), rtol=2 ** -7, atol=4e-3, what="attn decode wave spike")


def test_vt_pack():
    N, L, H, hd = 3, 77, 4, 64
    W = H * hd
    qkv = gen(N * L, 3 * W, seed=42).to(BF)
    Lpad = 128
    vt = torch.full((N, H, hd, Lpad), 5.0, dtype=BF, device=DEV)
    qd = qkv.to(DEV)
    ops.vt_pack(qd[:, 2 * W:], 3 * W, vt, N, L, H, hd, Lpad)
    v = qkv[:, 2 * W:].view(N, L, H, hd)
    assert torch.equal(vt[..., :L].cpu(), v.permute(0, 2, 3, 1))
    assert float(vt[..., L:].float().abs().sum()) == 0, "padding must be zero"


# ------------------------------------------------------------------ assembly / vision helpers
def test_embed_assemble_and_fill_rows():
    B, T, W, dim, V = 2, 5, 7, 64, 50
    table = gen(V, dim, seed=43)
    tok = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(2))
    h = torch.full((B * (T + W), dim), -1.0, device=DEV)
    ops.embed_assemble(tok.to(DEV), table.to(DEV), h, B, T, W, dim)
    hv = h.view(B, T + W, dim).cpu()
    emb = F.embedding(tok, table)
    assert torch.equal(hv[:, 0], emb[:, 0]) and torch.equal(hv[:, W + 1:], emb[:, 1:])
    assert float((hv[:, 1:W + 1] + 1).abs().sum()) == 0
    tag = gen(dim, seed=44)
    rows = torch.tensor([1, 4, 13], dtype=torch.int32, device=DEV)
    ops.fill_rows(tag.to(DEV), h, rows)
    assert torch.equal(h[rows.long()].cpu(), tag.expand(3, dim))
    hb = torch.zeros(B * T, dim, dtype=BF, device=DEV)
    ops.embed_assemble(tok.to(DEV), table.to(DEV), hb, B, T, 0, dim)   # fp32 table -> bf16 stream
    assert torch.equal(hb.view(B, T, dim).cpu(), emb.to(BF))


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_patch_embed_conv_as_gemm(dtype):
    N, P, g, width = 3, 14, 4, 64
    img = gen(N, 3, g * P, g * P, seed=45)
    wconv = gen(width, 3, P, P, seed=46, scale=0.05).to(dtype)
    K, Kpad = 3 * P * P, 640
    cols = torch.empty(N * g * g, Kpad, dtype=dtype, device=DEV)
    ops.patch_im2col(img.to(DEV), cols, P)      # fp32 image in, model dtype out
    w2 = torch.zeros(width, Kpad, dtype=dtype)
    w2[:, :K] = wconv.reshape(width, -1)
    out = torch.empty(N * g * g, width, dtype=dtype, device=DEV)
    ops.gemm_nt(cols, w2.to(DEV), out)
    want = F.conv2d(img.to(dtype).float(), wconv.float(), None, stride=P).flatten(2).permute(0, 2, 1).reshape(N * g * g, width)
    tol = dict(rtol=1e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=1e-2)
    assert_close(out, want, what="patch embed", **tol)


def test_vit_embed():
    N, T, W = 2, 16, 64
    patch, cls, pos = gen(N * T, W, seed=47), gen(W, seed=48), gen(T + 1, W, seed=49)
    x = torch.empty(N * (T + 1), W, device=DEV)
    ops.vit_embed(patch.to(DEV), cls.to(DEV), pos.to(DEV), x, N, T, W)
    want = torch.cat([cls.expand(N, 1, W), patch.view(N, T, W)], dim=1) + pos
    assert_close(x.view(N, T + 1, W), want, rtol=0, atol=0, what="vit embed")


@pytest.mark.parametrize("out_dtype", [torch.float32, BF])
def test_split_views(out_dtype):
    B, c = 2, 28
    img = gen(B, 3, 2 * c, 2 * c, seed=50)
    out = torch.empty(5 * B, 3, c, c, dtype=out_dtype, device=DEV)
    ops.split_views(img.to(DEV), out)
    want = ref_cpu.split_views(img, c).to(out_dtype)
    assert torch.equal(out[B:].cpu(), want[B:]), "quadrant crops are copies"
    # bicubic in fp16: one fp16 ulp of slack for the accumulation order
    assert_close(out[:B], want[:B], rtol=2 ** -9 if out_dtype == torch.float32 else 2 ** -7, atol=2e-3, what="bicubic view")


# ------------------------------------------------------------------ argmax / CE
def test_argmax_first_index_ties():
    B, V = 5, 32000
    lg = gen(B, V, seed=51)
    lg[0, 100] = lg[0, 20000] = 50.0
    lg[1, 31999] = 60.0
    lg[2, 0] = 60.0
    lg[3] = 1.0
    out = torch.empty(B, dtype=torch.long, device=DEV)
    ops.argmax(lg.to(DEV), out)
    assert out.cpu().tolist() == torch.argmax(lg, dim=-1).tolist()
    assert out[0].item() == 100 and out[3].item() == 0


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_cross_entropy(dtype):
    rows, V = 23, 32000
    lg = (gen(rows, V, seed=52) * 3).to(dtype)
    lab = torch.randint(1, V, (rows,), generator=torch.Generator().manual_seed(3))
    lab[::5] = 0
    row_loss = torch.empty(rows, device=DEV)
    dl = torch.empty(rows, V, dtype=dtype, device=DEV)
    nv = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.count_valid(lab.to(DEV), nv)
    assert int(nv.item()) == int((lab != 0).sum())
    ops.cross_entropy(lg.to(DEV), lab.to(DEV), row_loss, dl, nv, 1.0)
    lgf = lg.float().requires_grad_(True)
    want = F.cross_entropy(lgf, lab, ignore_index=0, reduction="none")
    assert_close(row_loss, want.detach(), rtol=1e-5, atol=1e-5, what="ce rows")
    F.cross_entropy(lgf, lab, ignore_index=0).backward()
    tol = dict(rtol=1e-4, atol=1e-8) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=1e-7)
    assert_close(dl, lgf.grad, what="ce grad", **tol)


@pytest.mark.parametrize("M,N,K,S", [(300, 64, 4096, 8), (64, 520, 8768, 16), (1000, 64, 22016, 4), (130, 96, 128, 2), (64, 64, 8768, 7)])
def test_gemm_nt_splitk(M, N, K, S):
    """split-K planes + ordered reduce == the fp32-accumulate / round-once result of the un-split GEMM; accumulate form adds
    into an fp32 destination; uneven slices (137 k-tiles over 16 / 7 slices)."""
    a, w = rt(gen(M, K, seed=31)), rt(gen(N, K, seed=32, scale=0.05))
    ad, wd = a.to(BF).to(DEV), w.to(BF).to(DEV)
    want = a @ w.t()
    scratch = torch.empty(S * M * N, dtype=torch.float32, device=DEV)
    ob = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm_nt_splitk(ad, wd, ob, scratch, S)
    assert_close(ob, want, rtol=2 ** -7, atol=2e-3 * math.sqrt(K) * 0.05 + 1e-3, what="splitk bf16")
    acc0 = gen(M, N, seed=33)
    of = acc0.to(DEV).clone()
    ops.gemm_nt_splitk(ad, wd, of, scratch, S, accumulate=True)
    assert_close(of, acc0 + rt(want), rtol=2 ** -7, atol=2e-3 * math.sqrt(K) * 0.05 + 1e-3, what="splitk f32 accumulate")
    ob2 = torch.empty_like(ob)
    ops.gemm_nt_splitk(ad, wd, ob2, scratch, S)
    assert torch.equal(ob, ob2)


@pytest.mark.parametrize("M,N,K,S", [(1000, 64, 22016, 4), (8728, 64, 4096, 3), (8728, 64, 4096, 7), (520, 48, 128, 2), (777, 64, 64, 1), (600, 64, 8768, 7),
                                     (512, 64, 192, 1)])
def test_skinny_nt_stages_write_the_planes_of_the_two_stage_kernel(M, N, K, S):
    """gemm_nt_skinny_kernel<64, 4> / <256, 3> (k-tiles kept in flight across the block barrier) against the two-stage kernels they replace
    for adapter-sized outputs: the split-K planes bit for bit (same fragments, same MFMA order per accumulator), ragged M, N < 64,
    slices of one k-tile (fewer k-tiles than stages), uneven slices, the library's own pick of the form; and the reduced bf16 result
    against the fp32 product."""
    from a3vlm_amd import lib as _lib
    a, w = rt(gen(M, K, seed=41)), rt(gen(N, K, seed=42, scale=0.05))
    ad, wd = a.to(BF).to(DEV), w.to(BF).to(DEV)
    planes, outs = {}, {}
    for narrow, nst in ((3, 2), (3, 4), (1, 3), (1, 2), (0, 0)):       # narrow 3 / 1: 64 / 256 rows per block; (0, 0): the library's own choice
        scratch = torch.full((S * M * N,), float("nan"), dtype=torch.float32, device=DEV)
        ob = torch.empty(M, N, dtype=BF, device=DEV)
        with _lib.env(A3V_SKINNY_STAGES=nst, A3V_SKINNY_NARROW=narrow):
            ops.gemm_nt_splitk(ad, wd, ob, scratch, S)
        torch.cuda.synchronize()
        planes[narrow, nst], outs[narrow, nst] = scratch.clone(), ob
    for key in planes:
        assert torch.equal(planes[key], planes[3, 2]), f"rows / stages {key}: planes differ"
        assert torch.equal(outs[key], outs[3, 2])
    assert_close(outs[3, 4], a @ w.t(), rtol=2 ** -7, atol=2e-3 * math.sqrt(K) * 0.05 + 1e-3, what="skinny bf16")


@pytest.mark.parametrize("M,N,K,lda,ldw", [(512, 512, 256, 512, 512), (4096, 1024, 1000, 4096, 1024), (264, 260, 70, 272, 264),
                                           (1024, 4096, 8728, 3 * 1024, 4096), (256, 256, 1, 256, 256)])
def test_gemm_tn(M, N, K, lda, ldw):
    """C = At^T @ Wt with both operands row-indexed by the contracted index (ds_read_tr fragments, swizzled [k][m] LDS tiles):
    equal to the NT kernel on the transposed operands to fp32-accumulation order; ragged K (buffer OOB zero fill), strided
    operands (a q-slice of a fused qkv gradient), partial tiles, and the three epilogues the weight-gradient step uses."""
    at_full, wt_full = rt(gen(K, lda, seed=41)), rt(gen(K, ldw, seed=42, scale=0.05))
    atd, wtd = at_full.to(BF).to(DEV)[:, :M], wt_full.to(BF).to(DEV)[:, :N]
    want = at_full[:, :M].t() @ wt_full[:, :N]
    tol = dict(rtol=2 ** -7, atol=2e-3 * math.sqrt(K) * 0.05 + 1e-3)
    ob = torch.empty(M, N, dtype=BF, device=DEV)
    ops.gemm_tn(atd, wtd, ob)
    assert_close(ob, want, what="tn bf16", **tol)
    of = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm_tn(atd, wtd, of, epilogue=ops.EPI_OUT_F32)
    assert_close(of, want, what="tn f32 store", **tol)
    # bit-identical to the NT kernel on materialised transposes when K is a whole number of 64-tiles (same k order per lane)
    if K % 64 == 0:
        ref = torch.empty_like(of)
        ops.gemm_nt(atd.t().contiguous(), wtd.t().contiguous(), ref, epilogue=ops.EPI_OUT_F32)
        assert_close(of, ref, rtol=1e-5, atol=1e-5, what="tn vs nt")
    acc0 = gen(M, N, seed=43)
    og = acc0.to(DEV).clone()
    ops.gemm_tn(atd, wtd, og, residual=og, epilogue=ops.EPI_RES_F32)
    assert_close(og.cpu() - acc0, want, what="tn f32 accumulate", **tol)   # product rounded once to bf16, then added in fp32


@pytest.mark.parametrize("B,S,H,Hkv,hd,K,start_pos", [(2, 150, 4, 4, 128, 256, 0), (3, 77, 4, 2, 64, 128, 5), (8, 1091, 8, 8, 128, 512, 0),
                                                      (1, 40, 2, 1, 128, 192, 3)])
def test_gemm_qkv_rope_equals_gemm_then_rope(B, S, H, Hkv, hd, K, start_pos):
    """qkv GEMM with RoPE + KV-cache write in the epilogue == a3v_gemm_nt then a3v_rope_kvcache, bit for bit (same rounding
    points: accumulator -> bf16 qkv -> fp32 rotation -> bf16); GQA, both head sizes, a cache offset, and the 8 x 1091-row
    case that takes the hybrid 256x256 + 128x128-tail dispatch (rows of the tail carry their global token index)."""
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    rows, N = B * S, (H + 2 * Hkv) * hd
    Smax = (start_pos + S + 63) // 64 * 64 + 64
    x = gen(rows, K, seed=51).to(BF).to(DEV)
    w = gen(N, K, seed=52, scale=0.05).to(BF).to(DEV)
    cs = precompute_cos_sin(hd, 2 * Smax, 10000.0, None).to(DEV)
    outs = []
    for fused in (False, True):
        qkv = torch.zeros(rows, N, dtype=BF, device=DEV)
        kc = torch.zeros(B, Hkv, Smax, hd, dtype=BF, device=DEV)
        vc = torch.zeros(B, Hkv, hd, Smax, dtype=BF, device=DEV)
        if fused:
            vr = torch.zeros(rows, Hkv * hd + 8, dtype=BF, device=DEV)
            ops.gemm_qkv_rope(x, w, qkv, kc, vc, cs, B, S, H, Hkv, hd, start_pos, start_pos + 2, v_rows=vr[:, :Hkv * hd])
            assert torch.equal(vr[:, :Hkv * hd], outs[0][3]) and float(vr[:, Hkv * hd:].float().abs().sum()) == 0
        else:
            ops.gemm_nt(x, w, qkv)
            ops.rope_kvcache(qkv, qkv, kc, vc, cs, B, S, H, Hkv, hd, start_pos, start_pos + 2)
        outs.append((qkv[:, :H * hd].clone(), kc, vc, qkv[:, (H + Hkv) * hd:].clone()))
    for a, b, what in zip(outs[0][:3], outs[1][:3], ("q", "k cache", "v^T cache")):
        assert torch.equal(a, b), what
    assert float(outs[1][0].float().abs().max()) > 0.1


@pytest.mark.parametrize("M,N,K,S", [(4096, 64, 8728, 8), (64, 1024, 1000, 4), (520, 64, 130, 3), (64, 64, 4096, 16), (2048, 32, 640, 1)])
def test_gemm_tn_splitk(M, N, K, S):
    """adapter-sized TN products through split-K planes: slices of uneven k-tile counts, idle waves on either side of the
    tile, a ragged last k-tile; store and accumulate forms of the reduce; deterministic."""
    at, wt = rt(gen(K, M, seed=61)), rt(gen(K, N, seed=62, scale=0.05))
    atd, wtd = at.to(BF).to(DEV), wt.to(BF).to(DEV)
    want = at.t() @ wt
    tol = dict(rtol=2 ** -7, atol=2e-3 * math.sqrt(K) * 0.05 + 1e-3)
    scratch = torch.full((S * M * N,), float("nan"), dtype=torch.float32, device=DEV)
    of = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm_tn_splitk(atd, wtd, of, scratch, S)
    assert_close(of, want, what="tn splitk store", **tol)
    acc0 = gen(M, N, seed=63)
    og = acc0.to(DEV).clone()
    ops.gemm_tn_splitk(atd, wtd, og, scratch, S, accumulate=True)
    assert_close(og.cpu() - acc0, want, what="tn splitk accumulate", **tol)
    of2 = torch.empty_like(of)
    ops.gemm_tn_splitk(atd, wtd, of2, scratch, S)
    assert torch.equal(of, of2)


@pytest.mark.parametrize("M,N,K,mode", [(300, 520, 256, "plain"), (512, 256, 64, "plain"), (1091, 1024, 1280, "residual"),
                                        (8728, 4096, 2048, "plain"), (4452, 4096, 4096, "f32res"), (264, 264, 192, "f32")])
def test_gemm_nn(M, N, K, mode):
    """C = A @ Wt with Wt row-indexed by the contracted index (input gradients on the forward weight image): row-major A through
    the NT staging, Wt through the transpose reads; equal to the NT kernel on the materialised W^T; the 35 x 16- and 18 x 16-tile
    cases take the rows-in-rounds + split-K-tail dispatch."""
    a = rt(gen(M, K, seed=91)).to(BF).to(DEV)
    wt = rt(gen(K, N + 8, seed=92, scale=0.05)).to(BF).to(DEV)[:, :N]
    want = (a.float() @ wt.float()).cpu()
    atol = 2e-3 * math.sqrt(K) * 0.05 + 1e-3 + 2 ** -8 * float(want.abs().max())
    if mode == "plain":
        out = torch.empty(M, N, dtype=BF, device=DEV)
        ops.gemm_nn(a, wt, out)
        assert_close(out, want, rtol=2 ** -7, atol=atol, what="nn plain")
        ref = torch.empty_like(out)
        ops.gemm_nt(a, wt.t().contiguous(), ref)
        assert float((out.float() - ref.float()).abs().max()) <= 2 ** -7 * float(want.abs().max())
        out2 = torch.empty_like(out)
        ops.gemm_nn(a, wt, out2)
        assert torch.equal(out, out2)
    elif mode == "residual":
        res = rt(gen(M, N, seed=93))
        out = res.to(BF).to(DEV)
        ops.gemm_nn(a, wt, out, residual=out)
        assert_close(out, res + rt(want), rtol=2 ** -7, atol=atol, what="nn residual")
    elif mode == "f32":
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        ops.gemm_nn(a, wt, out, epilogue=ops.EPI_OUT_F32)
        assert_close(out, want, rtol=2 ** -7, atol=atol, what="nn f32")
    else:
        res = gen(M, N, seed=94)
        out = res.to(DEV).clone()
        ops.gemm_nn(a, wt, out, residual=out, epilogue=ops.EPI_RES_F32)
        assert_close(out.cpu() - res, want, rtol=2 ** -7, atol=atol, what="nn f32 residual")


def test_weight_and_input_gradient_gemms_at_7b_size():
    """BASELINE-size check of the TN / NN forms on the w2 shapes of Llama-2-7B at the bench's token count (8728, not a multiple
    of 64): the transposed-operand kernels must agree with the NT kernel on materialised transposes (same products, same
    per-lane k order; only the rows that go through a split-K tail may differ in fp32 summation order)."""
    T, N, K = 8728, 4096, 11008
    g = torch.Generator(device=DEV).manual_seed(5)
    dy = (torch.randn(T, N, device=DEV, generator=g) * 0.05).to(BF)
    x = (torch.randn(T, K, device=DEV, generator=g) * 0.5).to(BF)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.02).to(BF)
    # weight gradient dW[N, K] = dy^T x
    Tp = (T + 63) // 64 * 64
    dyt = torch.zeros(N, Tp, dtype=BF, device=DEV)
    xt = torch.zeros(K, Tp, dtype=BF, device=DEV)
    ops.transpose(dy, dyt, T, N, Tp)
    ops.transpose(x, xt, T, K, Tp)
    ref = torch.empty(N, K, dtype=torch.float32, device=DEV)
    ops.gemm_nt(dyt, xt, ref, epilogue=ops.EPI_OUT_F32)
    got = torch.empty_like(ref)
    ops.gemm_tn(dy, x, got, epilogue=ops.EPI_OUT_F32)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2 ** -7 * scale and float((got - ref).abs().mean()) <= 1e-4 * scale
    # input gradient dX[T, K] = dy W
    wt = torch.empty(K, N, dtype=BF, device=DEV)
    ops.transpose(w, wt, N, K, N)
    ref2 = torch.empty(T, K, dtype=BF, device=DEV)
    ops.gemm_nt(dy, wt, ref2)
    got2 = torch.empty_like(ref2)
    ops.gemm_nn(dy, w, got2)
    s2 = float(ref2.float().abs().max())
    d2 = (got2.float() - ref2.float()).abs()
    assert float(d2.max()) <= 2 ** -6 * s2 and float((d2 > 0).float().mean()) < 0.02      # bf16 rounding flips only
    # homogeneity of the TN product (a size-independent property that is exact in floating point): (2 a)^T x == 2 (a^T x)
    o3 = torch.empty(N, K, dtype=torch.float32, device=DEV)
    ops.gemm_tn((dy.float() * 2).to(BF), x, o3, epilogue=ops.EPI_OUT_F32)
    assert torch.equal(o3, got * 2)


def test_gemm_qkv_rope_with_additive_term():
    """The LoRA form of the fused projection: delta is added to the bf16 linear output (rounded) before the rotation ==
    gemm_nt with the bf16 residual epilogue followed by rope_kvcache, bit for bit; delta may live in the buffer that receives
    the token-major v (same element read, then written, by one lane)."""
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    B, S, H, Hkv, hd, K = 3, 211, 4, 2, 128, 256
    rows, N = B * S, (H + 2 * Hkv) * hd
    Smax = 256
    x = gen(rows, K, seed=101).to(BF).to(DEV)
    w = gen(N, K, seed=102, scale=0.05).to(BF).to(DEV)
    delta = gen(rows, N, seed=103, scale=0.3).to(BF).to(DEV)
    cs = precompute_cos_sin(hd, 2 * Smax, 10000.0, None).to(DEV)
    qkv = delta.clone()
    ops.gemm_nt(x, w, qkv, residual=qkv)                   # bf16(acc) + delta, rounded
    kc0 = torch.zeros(B, Hkv, Smax, hd, dtype=BF, device=DEV)
    vc0 = torch.zeros(B, Hkv, hd, Smax, dtype=BF, device=DEV)
    ops.rope_kvcache(qkv, qkv, kc0, vc0, cs, B, S, H, Hkv, hd, 0, 0)
    buf = delta.clone()
    qrot = torch.zeros(rows, H * hd, dtype=BF, device=DEV)
    kc1, vc1 = torch.zeros_like(kc0), torch.zeros_like(vc0)
    ops.gemm_qkv_rope(x, w, qrot, kc1, vc1, cs, B, S, H, Hkv, hd, 0, 0, v_rows=buf[:, (H + Hkv) * hd:], delta=buf)
    assert torch.equal(qrot, qkv[:, :H * hd]) and torch.equal(kc1, kc0) and torch.equal(vc1, vc0)
    assert torch.equal(buf[:, (H + Hkv) * hd:], qkv[:, (H + Hkv) * hd:])
    assert torch.equal(buf[:, :(H + Hkv) * hd], delta[:, :(H + Hkv) * hd])       # the q / k columns of delta are only read


@pytest.mark.parametrize("M,N,K,mode", [(22016, 4096, 1100, "store"), (4456, 4096, 2048, "acc"), (4456, 4096, 2048, "bf16")])
def test_gemm_tn_split_tail(M, N, K, mode):
    """dW-shaped TN products whose tile count leaves a partial round (86 x 16 / 18 x 16 tiles on 256 CUs): the rows beyond whole
    rounds go through split-K planes + the reduce epilogue; same values as one plain launch up to fp32 summation order."""
    g = torch.Generator(device=DEV).manual_seed(M + K)
    at = (torch.randn(K, M, device=DEV, generator=g) * 0.1).to(BF)
    wt = (torch.randn(K, N, device=DEV, generator=g) * 0.5).to(BF)
    want = (at.float().t() @ wt.float())
    scale = float(want.abs().max())
    if mode == "store":
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        ops.gemm_tn(at, wt, out, epilogue=ops.EPI_OUT_F32)
        assert float((out - want).abs().max()) <= 2 ** -7 * scale
    elif mode == "acc":
        acc0 = torch.randn(M, N, device=DEV, generator=g)
        out = acc0.clone()
        ops.gemm_tn(at, wt, out, residual=out, epilogue=ops.EPI_RES_F32)
        assert float((out - acc0 - want).abs().max()) <= 2 ** -7 * scale + 1e-5
    else:
        out = torch.empty(M, N, dtype=BF, device=DEV)
        ops.gemm_tn(at, wt, out)
        assert float((out.float() - want).abs().max()) <= 2 ** -7 * scale


# ------------------------------------------------------------------ epilogue forms
def _with_general_epilogue(fn):
    """fn() once with the interior-tile fast epilogues and once with A3V_GEMM_FAST_EPI=0 (every tile through the general form)."""
    from a3vlm_amd import lib
    outs = []
    for flag in ("1", "0"):
        with lib.env(A3V_GEMM_FAST_EPI=flag):
            outs.append(fn())
    return outs


@pytest.mark.parametrize("M,N,K", [(1000, 776, 256), (2048, 1024, 512), (300, 264, 128)])
@pytest.mark.parametrize("tile", ["auto", "pp", "t128"])
def test_gemm_fast_epilogue_forms_equal_general_form(M, N, K, tile):
    """The LDS-staged epilogues of tiles inside C (plain / bias+activation / bf16 residual / fp32 residual stream / fp32 output /
    SwiGLU) are bit-identical to the general per-lane epilogue, on shapes that mix interior and ragged tiles."""
    from a3vlm_amd import lib
    t = {"auto": 0, "pp": lib.EPI_TILE_256PP, "t128": lib.EPI_TILE_128}[tile]
    a, w = gen(M, K, seed=70).to(BF).to(DEV), gen(N, K, seed=71, scale=0.05).to(BF).to(DEV)
    bias, res, resf = gen(N, seed=72).to(BF).to(DEV), gen(M, N, seed=73).to(BF).to(DEV), gen(M, N, seed=74).to(DEV)

    def run(kind):
        def f():
            if kind == "swiglu":
                o = torch.full((M, N // 2), 3.0, dtype=BF, device=DEV)
                return ops.gemm_nt(a, w, o, epilogue=t | ops.EPI_SWIGLU)
            if kind in ("res_f32", "out_f32"):
                o = resf.clone() if kind == "res_f32" else torch.full((M, N), 3.0, device=DEV)
                return ops.gemm_nt(a, w, o, residual=o if kind == "res_f32" else None,
                                   epilogue=t | (ops.EPI_RES_F32 if kind == "res_f32" else ops.EPI_OUT_F32))
            o = res.clone() if "residual" in kind else torch.full((M, N), 3.0, dtype=BF, device=DEV)
            return ops.gemm_nt(a, w, o, bias=bias if "bias" in kind else None, residual=o if "residual" in kind else None,
                               epilogue=t | (ops.EPI_GELU if "gelu" in kind else 0) | (ops.EPI_QUICKGELU if "quick" in kind else 0))
        return f
    kinds = ["plain", "bias", "bias_gelu", "bias_quick", "residual", "bias_residual", "res_f32", "out_f32"] + (["swiglu"] if N % 32 == 0 else [])
    for kind in kinds:
        fast, general = _with_general_epilogue(run(kind))
        assert torch.equal(fast, general), kind


def test_gemm_ring_ragged_rows_over_several_tiles_per_block():
    """Persistent ring kernel on a shape with more tiles than CUs AND a ragged last tile row (M = 8728 = 34 x 256 + 24): blocks
    walk interior tiles (fast epilogue) and edge tiles (general epilogue, divergent lanes) in one launch.  An earlier build of the
    staged epilogues spilled ~100 dwords per lane in this kernel and produced wrong tiles exactly here; the bench shapes never
    showed it because the hybrid dispatch sends the ragged rows to the 128x128 kernel."""
    from a3vlm_amd import lib
    for (M, N, K) in [(8728, 4096, 1024), (8728, 3072, 256)]:
        a, w = gen(M, K, seed=80).to(BF).to(DEV), gen(N, K, seed=81, scale=0.05).to(BF).to(DEV)
        want = a.float() @ w.float().t()
        for dbg in ((0, 7) if lib.has_experiments() else (0,)):     # ring (default) and, in experiment builds, the two-stage kernel
            o = torch.full((M, N), 3.0, dtype=BF, device=DEV)
            ops.gemm_nt(a, w, o, epilogue=lib.EPI_TILE_256PP | (dbg << 24))
            assert_close(o, want, rtol=2 ** -7, atol=1e-3 * math.sqrt(K) * 0.05, what=f"ring ragged {M}x{N}x{K} dbg {dbg}")
        r = gen(M, N, seed=82).to(DEV)
        o = r.clone()
        ops.gemm_nt(a, w, o, residual=o, epilogue=lib.EPI_TILE_256PP | ops.EPI_RES_F32)
        assert_close(o, r.cpu() + rt(want.cpu()), rtol=2 ** -7, atol=0.04, what="ring ragged res_f32")   # 1 bf16 ulp of the product at |y| < 8


@pytest.mark.parametrize("M,N,K", [(4608, 4096, 128), (4608, 4096, 192), (8728, 4096, 320), (2048, 12288, 256), (6000, 5120 + 256, 704)])
def test_gemm_ring_stream_runs_on_across_the_tiles_of_a_block(M, N, K):
    """The persistent ring kernel keeps its DMA stream going ACROSS a block's tiles: the last two K-tile iterations of a tile fetch
    K-tiles 0 / 1 of the block's next tile (no prologue burst, no block barrier between tiles), the A parity and the W slot run on,
    and the epilogue stages through the one W slot no piece is in flight to.  Shapes with several tiles per block and short K loops
    (nk = 2, 3, 5, 4, 11: many boundaries per launch, odd and even nk, a ragged last tile row / column) must equal the 128 x 128
    kernel bit for bit (same MFMA, same K order) for every output kind of the fast epilogue, twice in a row (races show as flakes)."""
    from a3vlm_amd import lib
    a, w = gen(M, K, seed=90).to(BF).to(DEV), gen(N, K, seed=91, scale=0.05).to(BF).to(DEV)
    res_b, res_f = gen(M, N, seed=92).to(BF).to(DEV), gen(M, N, seed=93).to(DEV)

    def run(tile, kind):
        if kind == "plain":
            o = torch.full((M, N), 3.0, dtype=BF, device=DEV)
            return ops.gemm_nt(a, w, o, epilogue=tile)
        if kind == "residual":
            o = torch.empty(M, N, dtype=BF, device=DEV)
            return ops.gemm_nt(a, w, o, residual=res_b, epilogue=tile | ops.EPI_RESIDUAL)
        if kind == "res_f32":
            o = res_f.clone()
            return ops.gemm_nt(a, w, o, residual=o, epilogue=tile | ops.EPI_RES_F32)
        o = torch.empty(M, N, dtype=torch.float32, device=DEV)
        return ops.gemm_nt(a, w, o, epilogue=tile | ops.EPI_OUT_F32)

    for kind in ("plain", "residual", "res_f32", "out_f32"):
        want = run(lib.EPI_TILE_128, kind)
        for rep in range(2):
            got = run(lib.EPI_TILE_256PP, kind)
            assert torch.equal(got, want), (kind, rep, int((got != want).sum()))


@pytest.mark.parametrize("M,N,K", [(8728, 4096, 320), (8728, 4096, 128), (4616, 1024, 192), (1000, 520, 704), (8728, 1280, 64), (192 * 40 + 5, 4096, 256)])
def test_gemm_ring_192_row_tiles_equal_the_small_tile_kernel(M, N, K):
    """Round 5: the ring kernel's 192 x 256 tile form (six 16-row MFMA tiles per wave, 12-KiB A halves, 7 DMA pieces per wave and LOAD
    interval with their own counted waits) -- M = 8728 x N = 4096 is 2.875 rounds of these, no split-K planes for the rows beyond whole
    rounds of 256 x 256.  Same MFMA and K order as the 128 x 128 kernel: equal bit for bit for every output kind of the fast epilogue
    (plain / bf16 residual incl. its 2 + 1 chunk split at six row tiles / fp32 stream / fp32 out / SwiGLU backward) and through the
    general epilogue (bias + GELU), over several tiles per block, short and odd K loops (nk = 1 ... 11), ragged rows and columns,
    twice in a row."""
    from a3vlm_amd import lib
    a, w = gen(M, K, seed=60).to(BF).to(DEV), gen(N, K, seed=61, scale=0.05).to(BF).to(DEV)
    res_b, res_f = gen(M, N, seed=62).to(BF).to(DEV), gen(M, N, seed=63).to(DEV)
    bias = gen(N, seed=64).to(BF).to(DEV)
    gu = gen(M, 2 * N, seed=65).to(BF).to(DEV)

    def run(tile, kind):
        if kind == "plain":
            o = torch.full((M, N), 3.0, dtype=BF, device=DEV)
            return ops.gemm_nt(a, w, o, epilogue=tile)
        if kind == "residual":
            o = torch.empty(M, N, dtype=BF, device=DEV)
            return ops.gemm_nt(a, w, o, residual=res_b, epilogue=tile | ops.EPI_RESIDUAL)
        if kind == "res_f32":
            o = res_f.clone()
            return ops.gemm_nt(a, w, o, residual=o, epilogue=tile | ops.EPI_RES_F32)
        if kind == "out_f32":
            o = torch.empty(M, N, dtype=torch.float32, device=DEV)
            return ops.gemm_nt(a, w, o, epilogue=tile | ops.EPI_OUT_F32)
        if kind == "swiglu_bwd":
            o = torch.full((M, 2 * N), 5.0, dtype=BF, device=DEV)
            return ops.gemm_nt(a, w, o, residual=gu, epilogue=tile | lib.EPI_SWIGLU_BWD)
        o = torch.empty(M, N, dtype=BF, device=DEV)
        return ops.gemm_nt(a, w, o, bias=bias, residual=res_b, epilogue=tile | ops.EPI_GELU)

    kinds = ["plain", "residual", "res_f32", "out_f32", "gelu"] + (["swiglu_bwd"] if N % 8 == 0 else [])
    for kind in kinds:
        want = run(lib.EPI_TILE_128, kind)
        for rep in range(2):
            got = run(lib.EPI_TILE_192PP, kind)
            assert torch.equal(got, want), (kind, rep, int((got != want).sum()))


@pytest.mark.parametrize("M,F,K,tile", [(600, 1024, 256, "auto"), (600, 1024, 256, "pp"), (8728, 1280, 192, "auto"), (8728, 1280, 192, "pp"),
                                          (300, 528, 128, "t128"), (4608, 2816, 4160, "auto")])
def test_gemm_swiglu_backward_epilogue_equals_gemm_then_swiglu_bwd(M, F, K, tile):
    """A3V_EPI_SWIGLU_BWD (the input-gradient GEMM of w2 applies the SwiGLU backward to its own bf16-rounded product and writes
    d(gate) | d(up) into the w1|w3 gradient rows; llama_ens5.py:213-217 backward) == a3v_gemm_nt followed by a3v_swiglu_bwd, bit for
    bit: interior tiles of the ring kernel (fast form), ragged last tile rows / columns (general form), the 128 x 128 kernel, the
    hybrid dispatch with its tail rows; gu and the output inside wider rows (the K-extended buffers of the LoRA step)."""
    from a3vlm_amd import lib
    t = {"auto": 0, "pp": lib.EPI_TILE_256PP, "t128": lib.EPI_TILE_128}[tile]
    dy = gen(M, K, seed=131).to(BF).to(DEV)
    w = gen(F, K, seed=132, scale=0.05).to(BF).to(DEV)
    gu = gen(M, 2 * F, seed=133).to(BF).to(DEV)
    dact = torch.empty(M, F, dtype=BF, device=DEV)
    ops.gemm_nt(dy, w, dact, epilogue=t)
    want = torch.full((M, 2 * F + 64), 5.0, dtype=BF, device=DEV)
    ops.swiglu_bwd(gu, dact, want[:, :2 * F], F, interleaved=False)
    got = torch.full((M, 2 * F + 64), 5.0, dtype=BF, device=DEV)
    ops.gemm_nt(dy, w, got[:, :2 * F], residual=gu, epilogue=t | ops.EPI_SWIGLU_BWD)
    assert torch.equal(got, want), int((got != want).sum())
    # against the fp32 formula on the bf16-rounded product (the reference's autograd through F.silu(w1 x) * w3 x)
    g, u, da = gu[:, :F].float().cpu(), gu[:, F:].float().cpu(), rt(dy.float().cpu() @ w.float().cpu().t())
    sig = torch.sigmoid(g)
    assert_close(got[:, :F], da * u * (sig * (1 + g * (1 - sig))), rtol=2 ** -6, atol=2e-2, what="d gate")
    assert_close(got[:, F:2 * F], da * (g * sig), rtol=2 ** -6, atol=2e-2, what="d up")
    if K % 64 == 0 and tile == "auto":     # the NN form (full fine-tune: dX = dY . W on the forward image W [K, F])
        wt = w.t().contiguous()
        dact2 = torch.empty(M, F, dtype=BF, device=DEV)
        ops.gemm_nn(dy, wt, dact2)
        want2 = torch.full((M, 2 * F + 64), 5.0, dtype=BF, device=DEV)
        ops.swiglu_bwd(gu, dact2, want2[:, :2 * F], F, interleaved=False)
        got2 = torch.full((M, 2 * F + 64), 5.0, dtype=BF, device=DEV)
        ops.gemm_nn(dy, wt, got2[:, :2 * F], residual=gu, epilogue=ops.EPI_SWIGLU_BWD)
        assert torch.equal(got2, want2), int((got2 != want2).sum())


def test_gemm_one_wave_per_simd_kernel_equals_ring_kernel():
    """(Also the overlapped 8-wave form, A3V_GEMM_W4=20.)  The opt-in 4-wave (one wave per SIMD, 128 x 128 per wave, 5 x 32-KiB sub-stage ring) form of the NT kernel
    (A3V_GEMM_W4=1; DESIGN.md section 4: measured, slower than the ring kernel, kept for the record) accumulates in the same
    order through the same epilogues: bit-equal results on interior + ragged tiles, several tiles per block and the fp32 /
    residual / SwiGLU output kinds."""
    from a3vlm_amd import lib
    if not lib.has_experiments():
        pytest.skip("the experiment kernels are only in `make EXPERIMENTS=1` builds (not in the product library)")
    for (M, N, K) in [(8728, 3072, 256), (2048, 1024, 512), (520, 264, 128)]:
        a, w = gen(M, K, seed=83).to(BF).to(DEV), gen(N, K, seed=84, scale=0.05).to(BF).to(DEV)
        resf = gen(M, N, seed=85).to(DEV)

        def run(kind):
            if kind == "swiglu":
                o = torch.full((M, N // 2), 3.0, dtype=BF, device=DEV)
                return ops.gemm_nt(a, w, o, epilogue=lib.EPI_TILE_256PP | ops.EPI_SWIGLU)
            if kind == "res_f32":
                o = resf.clone()
                return ops.gemm_nt(a, w, o, residual=o, epilogue=lib.EPI_TILE_256PP | ops.EPI_RES_F32)
            o = torch.full((M, N), 3.0, dtype=torch.float32 if kind == "out_f32" else BF, device=DEV)
            return ops.gemm_nt(a, w, o, epilogue=lib.EPI_TILE_256PP | (ops.EPI_OUT_F32 if kind == "out_f32" else 0))
        for kind in ["plain", "res_f32", "out_f32"] + (["swiglu"] if N % 32 == 0 else []):
            outs = []
            for flag in ("0", "1", "20"):                                # ring, one wave per SIMD, overlapped 8-wave form
                with lib.env(A3V_GEMM_W4=flag):
                    outs.append(run(kind).clone())
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (M, N, K, kind)
        with lib.env(A3V_GEMM_W4="1"):
            o = run("plain")
        assert_close(o, a.float() @ w.float().t(), rtol=2 ** -7, atol=1e-3 * math.sqrt(K) * 0.05, what=f"w4 {M}x{N}x{K} vs fp32")


@pytest.mark.parametrize("M,N,K", [(1024, 768, 1000), (4352, 2048, 1024), (520, 264, 512)])
def test_gemm_tn_nn_fast_epilogue_forms_equal_general_form(M, N, K):
    at, wt = gen(K, M, seed=90).to(BF).to(DEV), gen(K, N, seed=91, scale=0.05).to(BF).to(DEV)
    a = gen(M, K, seed=92).to(BF).to(DEV)
    accf = gen(M, N, seed=93).to(DEV)

    def tn(dtype, acc):
        def f():
            o = accf.clone() if acc else torch.zeros(M, N, dtype=dtype, device=DEV)
            ops.gemm_tn(at, wt, o, residual=o if acc else None,
                        epilogue=ops.EPI_RES_F32 if acc else (ops.EPI_OUT_F32 if dtype == torch.float32 else 0))
            return o
        return f

    def nn(dtype):
        def f():
            o = torch.zeros(M, N, dtype=dtype, device=DEV)
            ops.gemm_nn(a, wt, o, epilogue=ops.EPI_OUT_F32 if dtype == torch.float32 else 0)
            return o
        return f
    cases = [tn(BF, False), tn(torch.float32, False), tn(torch.float32, True)] + ([nn(BF), nn(torch.float32)] if K % 64 == 0 else [])
    for f in cases:
        fast, general = _with_general_epilogue(f)
        assert torch.equal(fast, general)


@pytest.mark.parametrize("n", [4, 1003, 4096 * 4096 + 12, 200_000_001])
def test_sumsq_partials(n):
    """Partial sums of squares of a gradient bucket (the clip's norm = sqrt of the sum of all partials): against a float64 sum, and
    bit-identical across launches."""
    x = torch.randn(n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(n % 97))
    out = torch.full((ops.SUMSQ_SLOTS,), float("nan"), device=DEV)
    ops.sumsq_partials(x, out)
    got = out.double().sum()
    want = (x.double() ** 2).sum()
    assert abs(float(got - want)) <= 1e-6 * float(want)
    out2 = torch.empty_like(out)
    ops.sumsq_partials(x, out2)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("M,N,K", [(4616, 1024, 4096), (1024, 768, 2048), (600, 1024, 4096)])
def test_gemm_few_big_tiles_go_split_k_on_the_ring_kernel(M, N, K):
    """Small-N problems (the ViT's output projections: 76 big tiles for 256 CUs) run on the ring kernel split over K with the
    epilogue (bias, bf16 rounding, residual) applied by the reduce pass: against an fp32 reference, against the small-tile
    kernel (another summation order: 1 bf16 ulp of the product), and deterministic across launches."""
    from a3vlm_amd import lib
    a, w = rt(gen(M, K, seed=95)), rt(gen(N, K, seed=96, scale=0.03))
    bias, res = rt(gen(N, seed=97)), rt(gen(M, N, seed=98))
    ad, wd, bd = a.to(BF).to(DEV), w.to(BF).to(DEV), bias.to(BF).to(DEV)
    want = rt(res + rt(a @ w.t() + bias))
    outs = []
    for rep in range(2):
        h = res.to(BF).to(DEV).clone()
        ops.gemm_nt(ad, wd, h, bias=bd, residual=h)                      # auto dispatch
        outs.append(h)
    assert torch.equal(outs[0], outs[1])
    assert_close(outs[0], want, rtol=2 ** -6, atol=0.05, what="ring split-K, bias + residual")
    h128 = res.to(BF).to(DEV).clone()
    ops.gemm_nt(ad, wd, h128, bias=bd, residual=h128, epilogue=lib.EPI_TILE_128)
    assert_close(outs[0], h128.float().cpu(), rtol=2 ** -6, atol=0.05, what="vs the small-tile kernel")
    o32 = torch.empty(M, N, device=DEV)
    ops.gemm_nt(ad, wd, o32, epilogue=ops.EPI_OUT_F32)
    assert_close(o32, rt(a @ w.t()), rtol=2 ** -7, atol=0.02, what="ring split-K, fp32 out")


@pytest.mark.parametrize("Kt,R,N,S,ld_extra", [(8728, 64, 4096, 32, 0), (1000, 48, 1024, 5, 4096), (130, 16, 384, 3, 64), (64, 64, 128, 1, 0),
                                                (700, 32, 22016 // 8, 4, 4160 - 64)])
def test_gemm_tn_strip(Kt, R, N, S, ld_extra):
    """a3v_gemm_tn_strip (adapter weight gradients: out[R, N] = T^T X over the token rows; 64 x 128 tiles, transposing LDS reads on both
    operands with the 128- / 256-byte-row swizzles) against the fp32 product of the same bf16 operands and against the 256 x 256 TN
    split-K kernel: ragged last k-tile, R < 64 inside a wider row (the view the engine passes: dt[:, :n r] / t inside the K-extended
    buffers), store and accumulate, deterministic."""
    wide = rt(gen(Kt, 64 + ld_extra, seed=111))
    x = rt(gen(Kt, N, seed=112, scale=0.05))
    td_full = wide.to(BF).to(DEV)
    td = td_full[:, :R]                        # row stride 64 + ld_extra; columns R..63 hold other (finite) data
    xd = x.to(BF).to(DEV)
    want = wide[:, :R].t() @ x
    tol = dict(rtol=2 ** -7, atol=2e-3 * math.sqrt(Kt) * 0.05 + 1e-3)
    scratch = torch.full((S * R * N,), float("nan"), dtype=torch.float32, device=DEV)
    of = torch.full((R, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm_tn_strip(td, xd, of, scratch, S)
    assert_close(of, want, what="tn strip store", **tol)
    if R % 8 == 0:
        ref = torch.empty_like(of)
        S2 = min(S, (Kt + 63) // 64, 16)
        ops.gemm_tn_splitk(td, xd, ref, torch.empty(S2 * R * N, dtype=torch.float32, device=DEV), S2)
        assert_close(of, ref, rtol=2 ** -7, atol=tol["atol"], what="tn strip vs tn split-K")
    acc0 = gen(R, N, seed=113)
    og = acc0.to(DEV).clone()
    ops.gemm_tn_strip(td, xd, og, scratch, S, accumulate=True)
    assert_close(og.cpu() - acc0, want, what="tn strip accumulate", **tol)
    of2 = torch.empty_like(of)
    ops.gemm_tn_strip(td, xd, of2, scratch, S)
    assert torch.equal(of, of2)


def test_gemm_split_k_scratch_is_per_stream():
    """Two streams run hybrid-tail GEMMs (big tiles + split-K tail through the registered scratch) concurrently, many times: each stream
    has its own scratch (a3v_gemm_set_workspace_for keys registrations by device and stream; round 3 kept one process-global pointer
    that both tails wrote), so both results equal the serial ones bit for bit."""
    from a3vlm_amd import lib as _lib
    g = torch.Generator().manual_seed(77)
    M, N = 8 * 1091, 4096
    shapes = [(4096, 1), (11008, 2)]                         # the wo / w2 shapes of the 7B step: 2.19 tile rounds -> split-K tail
    data = []
    for K, seed in shapes:
        a = (torch.randn(M, K, generator=g)).to(BF).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
        r = torch.randn(M, N, generator=g).to(BF).to(DEV)
        ref = torch.empty(M, N, dtype=BF, device=DEV)
        ops.gemm_nt(a, w, ref, residual=r)
        data.append((a, w, r, ref))
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [torch.empty(M, N, dtype=BF, device=DEV) for _ in data]
    for rep in range(6):
        for o in outs:
            o.fill_(float("nan"))
        torch.cuda.synchronize()
        for st, (a, w, r, _), o in zip((s1, s2), data, outs):
            with torch.cuda.stream(st):
                for _ in range(3):
                    ops.gemm_nt(a, w, o, residual=r)
        torch.cuda.synchronize()
        for (_, _, _, ref), o in zip(data, outs):
            assert torch.equal(o, ref), rep
    # a process that runs GEMMs on many streams keeps at most GEMM_WS_MAX_STREAMS scratch buffers (least recently used dropped first,
    # its registration removed before the memory goes back to the allocator); an evicted stream simply gets a new one
    many = [torch.cuda.Stream() for _ in range(ops.GEMM_WS_MAX_STREAMS + 3)]
    (a, w, r, ref) = data[0]
    for rep in range(2):
        for st in many:
            o = torch.empty(M, N, dtype=BF, device=DEV)
            with torch.cuda.stream(st):
                ops.gemm_nt(a, w, o, residual=r)
            st.synchronize()
            assert torch.equal(o, ref)
            assert len(ops._gemm_ws) <= ops.GEMM_WS_MAX_STREAMS
    # the legacy registration binds to the first stream that uses it and is never handed to another one
    L = _lib.load()
    ws = torch.empty(96 << 20, dtype=torch.uint8, device=DEV)
    assert L.a3v_gemm_set_workspace(ws.data_ptr(), ws.numel()) == 0
    assert L.a3v_gemm_set_workspace(None, 0) == 0


def test_mfma_probe_reports_a_plausible_pipe_rate():
    """tools/ubench/liba3v_probe.so (bench.py `roofline.mfma_pipe_measured`; not part of the product C-ABI): a bare MFMA stream lands between a third of and just above the
    nominal 2.5 PF/s, and random operands are never faster than constants by more than the run-to-run spread."""
    from tools.ubench.probe import probe_mfma_tflops
    const, rand = probe_mfma_tflops(4000)
    assert 800.0 < rand < 2700.0 and 800.0 < const < 2700.0, (const, rand)
    assert rand < const * 1.05, (const, rand)


@pytest.mark.parametrize("M,N,K,epi", [(8, 4096, 4096, "res"), (8, 4096, 11008, "res"), (8, 32000, 4096, "f32"), (3, 1024, 512, ""), (8, 22016, 4096, "swiglu"),
                                       (5, 10240, 5120, "swiglu"), (1, 64, 256, ""), (8, 12288, 4096, "")])
def test_gemv_k_slices_inside_the_block_equal_the_across_blocks_form(M, N, K, epi):
    """Round 4: the decode GEMVs without an RMSNorm prologue put the K slices of a 16-row tile into ONE block (gemv_kq_bf16_kernel: the partial
    accumulators meet in LDS, no split-K fix-up through HBM).  Same slice order => the same bits as the across-blocks kernel
    (A3V_GEMV_KQ=0) whenever the slice counts agree, and both within bf16 rounding of the fp32 product."""
    from a3vlm_amd import lib as _l2
    g = torch.Generator().manual_seed(N + K + M)
    a = torch.randn(M, K, generator=g).to(BF).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
    flags = {"res": ops.EPI_RESIDUAL, "f32": ops.EPI_OUT_F32, "swiglu": ops.EPI_SWIGLU, "": 0}[epi]
    No = N // 2 if epi == "swiglu" else N
    res = torch.randn(M, No, generator=g).to(BF).to(DEV) if epi == "res" else None
    ws = ops.gemm_skinny_workspace(M, N, K, DEV)
    outs = {}
    for mode in ("2", "0"):
        with _l2.env(A3V_GEMV_KQ=mode):
            out = torch.zeros(M, No, device=DEV, dtype=torch.float32 if epi == "f32" else BF)
            ops.gemm_skinny(a, w, out, ws, residual=res, epilogue=flags)
            torch.cuda.synchronize()
            outs[mode] = out.float().cpu()
    y = a.float().cpu() @ w.float().cpu().t()
    if epi == "swiglu":
        y = y.view(M, N // 32, 2, 16)
        gq, uq = y[:, :, 0].to(BF).float(), y[:, :, 1].to(BF).float()
        y = (torch.nn.functional.silu(gq).to(BF).float() * uq).reshape(M, N // 2)
    if res is not None:
        y = y.to(BF).float() + res.float().cpu()
    for mode in ("2", "0"):
        err = float((outs[mode] - y).abs().max() / y.abs().max())
        assert err < 2e-2, (mode, err)
    same_plan = not (epi == "swiglu" and L_skinny_split(M, N, K) > 6)
    if same_plan:
        assert torch.equal(outs["2"], outs["0"])
    assert int((ws.view(torch.int32)[: 4096] != 0).sum()) == 0          # the arrival counters are left at zero by both forms


def L_skinny_split(M, N, K):
    from a3vlm_amd import lib as _l3
    return int(_l3.load().a3v_gemm_skinny_split(M, N, K))


def test_wave_reductions_on_the_valu_equal_the_shuffle_forms_bit_for_bit():
    """csrc/a3v_common.h wave_sum / wave_max (round 4: v_permlane32_swap, v_permlane16_swap, DPP row_ror / quad_perm -- no LDS round
    trips) pair the same lanes in the same order as the __shfl_xor butterflies they replaced: identical bits on random data, on data
    with infinities / huge dynamic range, and on denormals."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4096, 64, generator=g)
    x[100:200] *= torch.logspace(-30, 30, 100)[:, None]
    x[300:310, ::7] = float("inf")
    x[320:330, 3] = float("-inf")
    x[400:420] *= 1e-41
    from tools.ubench.probe import probe_wave_reduce
    assert probe_wave_reduce(x.to(DEV).contiguous()) == 0
