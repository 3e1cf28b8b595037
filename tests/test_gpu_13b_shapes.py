"""-m gpu: the MFMA kernels at the Llama-2-13B shapes of BASELINE configs[3] (dim 5120, ffn 13824, 40 heads of 128, 8 x 1091 =
8728 token rows; reference geometry defaults LLM/llama_ens5.py:35-37, ffn rule :196-200) against the CPU oracle.

These are the tile-tail / split-K / hybrid-dispatch cases the 7B shapes do not reach: N in {5120, 15360, 27648}, K in {5120, 13824},
M = 8728 = 34 x 256 + 24.  The full products are too big for the CPU oracle to finish in seconds, so the device result is computed
in full and the oracle (fp32 matmul of the same bf16-rounded operands, ref_cpu's linear / sdpa arithmetic) checks a ROW SAMPLE that
contains the first tile, the ragged last tile rows, both sides of the hybrid dispatch's row split and random rows in between, over
ALL columns (so every N tile incl. the N tail and every k-tile is covered)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from a3vlm_amd import ops  # noqa: E402
from oracle import ref_cpu  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
ROWS = 8 * 1091
DIM, FFN, H, HD = 5120, 13824, 40, 128


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def row_sample(M, seed, n=160):
    g = torch.Generator().manual_seed(seed)
    fixed = list(range(0, 40)) + list(range(M - 40, M)) + [255, 256, 257, 8191, 8192, 8193, 8703, 8704, 8705]
    rnd = torch.randint(0, M, (n,), generator=g).tolist()
    return torch.tensor(sorted({r for r in fixed + rnd if 0 <= r < M}))


def close(got, want, K, what, scale=0.05):
    got, want = got.float().cpu(), want.float()
    err = (got - want).abs()
    bound = 1e-3 * math.sqrt(K) * scale + 2 ** -7 * want.abs()         # bf16 output rounding + fp32 accumulation-order noise
    bad = err > bound
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {float(err.max()):.4g}"


@pytest.mark.parametrize("N,K,epi", [(3 * DIM, DIM, "plain"), (DIM, DIM, "res_f32"), (2 * FFN, DIM, "swiglu"), (DIM, FFN, "res_f32"),
                                     (2 * FFN, DIM, "plain")])
def test_gemm_nt_13b_forward_linears(N, K, epi):
    """wqkv / wo / w1|w3 (+SwiGLU epilogue on the interleaved image) / w2 (fp32 residual stream) of a 13B block: x @ W^T
    (llama_ens5.py:63-90, 202-217)."""
    a = gen(ROWS, K, seed=1).to(BF)
    w = gen(N, K, seed=2, scale=0.05).to(BF)
    ad, wd = a.to(DEV), w.to(DEV)
    rows = row_sample(ROWS, 3)
    full = a[rows].float() @ w.float().t()
    if epi == "plain":
        out = torch.full((ROWS, N), float("nan"), dtype=BF, device=DEV)
        ops.gemm_nt(ad, wd, out)
        close(out[rows.to(DEV)], full, K, f"nt {N}x{K}")
    elif epi == "res_f32":
        res = gen(ROWS, N, seed=4)
        out = res.to(DEV).clone()
        ops.gemm_nt(ad, wd, out, residual=out, epilogue=ops.EPI_RES_F32)
        got = out[rows.to(DEV)].cpu() - res[rows]
        close(got, full.to(BF).float(), K, f"nt res_f32 {N}x{K}")
    else:
        # interleaved image: 16-row blocks of w1 and w3 alternate; out = silu(g) * u on bf16-rounded g, u
        F_ = N // 2
        w1, w3 = w[:F_], w[F_:]
        wi = torch.stack((w1.view(F_ // 16, 16, K), w3.view(F_ // 16, 16, K)), dim=1).reshape(N, K).contiguous()
        out = torch.full((ROWS, F_), float("nan"), dtype=BF, device=DEV)
        ops.gemm_nt(ad, wi.to(DEV), out, epilogue=ops.EPI_SWIGLU)
        g_ = (a[rows].float() @ w1.float().t()).to(BF).float()
        u_ = (a[rows].float() @ w3.float().t()).to(BF).float()
        want = torch.nn.functional.silu(g_).to(BF).float() * u_
        got = out[rows.to(DEV)].float().cpu()
        err = (got - want).abs()
        assert float((err / (want.abs() + 0.05)).max()) < 2 ** -5, float(err.max())       # two bf16 roundings in front of the product


@pytest.mark.parametrize("N,K", [(DIM, 3 * DIM), (DIM, DIM), (DIM, 2 * FFN), (FFN, DIM)])
def test_gemm_nn_13b_input_gradients(N, K):
    """dX = dY @ W on the forward weight image [K = out features, N = in features]."""
    dy = gen(ROWS, K, seed=5).to(BF)
    w = gen(K, N, seed=6, scale=0.05).to(BF)
    out = torch.full((ROWS, N), float("nan"), dtype=BF, device=DEV)
    ops.gemm_nn(dy.to(DEV), w.to(DEV), out)
    rows = row_sample(ROWS, 7)
    close(out[rows.to(DEV)], dy[rows].float() @ w.float(), K, f"nn {N}x{K}")


@pytest.mark.parametrize("M,N", [(3 * DIM, DIM), (DIM, DIM), (2 * FFN, DIM), (DIM, FFN)])
def test_gemm_tn_13b_weight_gradients(M, N):
    """dW = dY^T @ X straight from the token-major operands (contracted index = the 8728 token rows: a ragged last k-tile),
    fp32 store and fp32 accumulate (gradient accumulation over micro-steps), with the clip's sums of squares."""
    dy = gen(ROWS, M, seed=8).to(BF)
    x = gen(ROWS, N, seed=9, scale=0.05).to(BF)
    dyd, xd = dy.to(DEV), x.to(DEV)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    sq = torch.zeros(ops.gemm_tn_sumsq_slots(M, N), dtype=torch.float32, device=DEV)
    ops.gemm_tn(dyd, xd, out, epilogue=ops.EPI_OUT_F32, sumsq=sq)
    rows = row_sample(M, 10)
    want = dy[:, rows].float().t() @ x.float()
    close(out[rows.to(DEV)], want, ROWS, f"tn {M}x{N}")
    assert abs(float(sq.sum()) / float((out.double() ** 2).sum()) - 1.0) < 1e-4
    acc = out.clone()
    ops.gemm_tn(dyd, xd, acc, residual=acc, epilogue=ops.EPI_RES_F32)
    close(acc[rows.to(DEV)] - out[rows.to(DEV)], want.to(BF).float(), ROWS, f"tn accumulate {M}x{N}")


def test_qkv_rope_and_attention_13b_heads():
    """40 heads x 128 at B = 8, S = 1091: the fused qkv GEMM + RoPE + cache-write epilogue equals GEMM then RoPE kernel bit for
    bit, the rotated q / cached k of sampled (batch, head, position) triples equal the oracle's apply_rotary_emb on the oracle's
    projection, and the causal prefill attention over all 40 heads equals ref_cpu.sdpa on sampled batch rows / heads."""
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    B, S = 8, 1091
    N, Smax = 3 * DIM, 1152
    x = gen(ROWS, DIM, seed=11, scale=0.5).to(BF)
    w = gen(N, DIM, seed=12, scale=0.02).to(BF)
    xd, wd = x.to(DEV), w.to(DEV)
    cs = precompute_cos_sin(HD, 2 * Smax, 10000.0, None).to(DEV)
    qf = torch.zeros(ROWS, N, dtype=BF, device=DEV)
    kc = torch.zeros(B, H, Smax, HD, dtype=BF, device=DEV)
    vc = torch.zeros(B, H, HD, Smax, dtype=BF, device=DEV)
    ops.gemm_qkv_rope(xd, wd, qf, kc, vc, cs, B, S, H, H, HD, 0, 0)
    q2 = torch.zeros(ROWS, N, dtype=BF, device=DEV)
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.gemm_nt(xd, wd, q2)
    ops.rope_kvcache(q2, q2, kc2, vc2, cs, B, S, H, H, HD, 0, 0)
    assert torch.equal(qf[:, :H * HD], q2[:, :H * HD]) and torch.equal(kc, kc2) and torch.equal(vc, vc2)
    # oracle: projection + rotary on sampled rows (llama_ens5.py:114-118 call site; RoPE restated in ref_cpu)
    rows = row_sample(ROWS, 13, n=60)
    proj = (x[rows].float() @ w.float().t()).to(BF).float()
    fc = ref_cpu.precompute_freqs_cis(HD, 2 * Smax)
    pos = rows % S
    xq = proj[:, :H * HD].view(-1, 1, H, HD)
    xk = proj[:, H * HD:2 * H * HD].view(-1, 1, H, HD)
    for i, r in enumerate(rows.tolist()):
        oq, ok = ref_cpu.apply_rotary_emb(xq[i:i + 1], xk[i:i + 1], fc[pos[i]:pos[i] + 1])
        gq = qf[r, :H * HD].float().cpu().view(H, HD)
        gk = kc[r // S, :, r % S].float().cpu()
        assert float((gq - oq[0, 0]).abs().max()) < 2 ** -7 * float(oq.abs().max()) + 1e-3
        assert float((gk - ok[0, 0]).abs().max()) < 2 ** -7 * float(ok.abs().max()) + 1e-3
    # attention over the caches just written (all 40 heads on the device; oracle on 2 batch rows x 5 heads)
    att = torch.zeros(ROWS, H * HD, dtype=BF, device=DEV)
    ld = qf.stride(0)
    strides = (S * ld, ld, HD, H * Smax * HD, Smax * HD, HD, H * HD * Smax, HD * Smax, Smax, S * H * HD, H * HD, HD)
    ops.attention(qf, kc, vc, att, B, S, S, H, H, HD, strides, True)
    mask = ref_cpu.make_causal_mask(S, S)
    for b in (0, 7):
        for h in (0, 13, 26, 38, 39):
            q = qf[b * S:(b + 1) * S, h * HD:(h + 1) * HD].float().cpu()[None, None]
            k = kc[b, h, :S].float().cpu()[None, None]
            v = vc[b, h, :, :S].float().cpu().t()[None, None]
            want = ref_cpu.sdpa(q, k, v, mask)[0, 0]
            got = att[b * S:(b + 1) * S, h * HD:(h + 1) * HD].float().cpu()
            assert float((got - want).abs().max()) < 2 ** -6 * float(want.abs().max()) + 4e-3, (b, h)


@pytest.mark.parametrize("form", ["nt", "nn"])
def test_gemm_swiglu_backward_epilogue_13b(form):
    """The input gradient of w2 with the SwiGLU backward in its epilogue (A3V_EPI_SWIGLU_BWD) at the 13B shape: d(act) = dy [8728, 5120]
    . W2 [5120, 13824], then d(gate) | d(up) from the forward's gate | up rows.  NT form (LoRA: over the transposed frozen image, with
    the 64 adapter columns appended to K) and NN form (full fine-tune: on the forward image); a row sample against the fp32 formula on
    the bf16-rounded product (llama_ens5.py:213-217 backward through autocast), every column."""
    K = DIM + (64 if form == "nt" else 0)
    dy = gen(ROWS, K, seed=41).to(BF)
    w = gen(FFN, K, seed=42, scale=0.05).to(BF)                       # rows = ffn columns of d(act)
    gu = gen(ROWS, 2 * FFN, seed=43).to(BF)
    dyd, gud = dy.to(DEV), gu.to(DEV)
    out = torch.full((ROWS, 2 * FFN + 64), float("nan"), dtype=BF, device=DEV)
    if form == "nt":
        ops.gemm_nt(dyd, w.to(DEV), out[:, :2 * FFN], residual=gud, epilogue=ops.EPI_SWIGLU_BWD)
    else:
        ops.gemm_nn(dyd, w.t().contiguous().to(DEV), out[:, :2 * FFN], residual=gud, epilogue=ops.EPI_SWIGLU_BWD)
    assert bool(torch.isnan(out[:, 2 * FFN:].float()).all())          # nothing written past the [d gate | d up] window
    rows = row_sample(ROWS, 44)
    da = (dy[rows].float() @ w.float().t()).to(BF).float()            # the bf16 d(act) the un-fused path stores
    g, u = gu[rows, :FFN].float(), gu[rows, FFN:].float()
    sig = torch.sigmoid(g)
    got = out[rows.to(DEV)].float().cpu()
    for what, have, want in (("d gate", got[:, :FFN], da * u * (sig * (1 + g * (1 - sig)))), ("d up", got[:, FFN:2 * FFN], da * (g * sig))):
        err = (have - want).abs()
        # da itself carries the product's accumulation-order noise (one bf16 ulp of |da| where a rounding boundary is crossed)
        bound = 2 ** -6 * want.abs() + 2 ** -7 * (da.abs() + 1e-3 * math.sqrt(K) * 0.05) * (u.abs() + g.abs() + 1) + 1e-3
        assert not bool((err > bound).any()), f"{form} {what}: {int((err > bound).sum())} out of tolerance, max err {float(err.max()):.4g}"


def test_gemm_tn_strip_13b_adapter_gradients():
    """Adapter weight gradients at the 13B widths: dB^T = t^T . dy with dy [8728, 27648] (the fused w1|w3 group) and dA = dt^T . x with
    x [8728, 13824] (w2's input), R = 48 / 16 real adapter columns inside the 64-wide K-extension block (model/peft.py:58-159)."""
    for N, R, seed in ((2 * FFN, 32, 51), (FFN, 16, 52), (3 * DIM, 48, 53)):
        t_full = gen(ROWS, 64, seed=seed).to(BF)
        x = gen(ROWS, N, seed=seed + 10, scale=0.05).to(BF)
        td, xd = t_full.to(DEV)[:, :R], x.to(DEV)
        S = 16
        out = torch.full((R, N), float("nan"), dtype=torch.float32, device=DEV)
        ops.gemm_tn_strip(td, xd, out, torch.empty(S * R * N, dtype=torch.float32, device=DEV), S)
        want = t_full[:, :R].float().t() @ x.float()
        err = (out.cpu() - want).abs()
        bound = 2 ** -7 * want.abs() + 2e-3 * math.sqrt(ROWS) * 0.05 + 1e-3
        assert not bool((err > bound).any()), f"strip R={R} N={N}: max err {float(err.max()):.4g}"
