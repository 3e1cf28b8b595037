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
from a3vlm_amd.quant import dequantize_rows_fp8, quantize_rows_fp8
from oracle import ref_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def rt(x):
    return x.to(BF).float()


def gen(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


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
    from a3vlm_amd.quant import dequantize_rows_fp8, quantize_rows_fp8
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
