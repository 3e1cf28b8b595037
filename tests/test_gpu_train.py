"""-m gpu: backward kernels and the full training step against torch autograd over the CPU oracle
(oracle/ref_cpu.py is built from differentiable torch ops, so its autograd IS the reference gradient:
engine_finetune.py:44-68 semantics).  fp32 parity path: 1e-3 relative; bf16 compute: stated per check."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from a3vlm_amd import ops  # noqa: E402
from a3vlm_amd.model.LLM import llama_ens5 as plugin  # noqa: E402
from a3vlm_amd.model.meta import MetaModel  # noqa: E402
from a3vlm_amd.train import TrainEngine  # noqa: E402
from a3vlm_amd.util import promote_trainable_params_to_fp32  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from oracle.gen_golden import synth_image  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def gen(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def relerr(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-12))


# ------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_transpose_and_cast(dtype):
    x = gen(3, 70, 130, seed=1).to(dtype)
    xd = x.to(DEV)
    out = torch.full((3, 130, 128), 9.0, dtype=dtype, device=DEV)
    ops.transpose(xd, out, 70, 130, 128, batch=3, bs_src=70 * 130, bs_dst=130 * 128)
    assert torch.equal(out[:, :, :70].cpu(), x.transpose(1, 2))
    assert float(out[:, :, 70:].float().abs().sum()) == 0
    # vector path (bf16, C % 8 == 0, Rpad % 8 == 0): ragged R, strided source rows, several tiles
    for (R, C, Rp, ld) in [(70, 136, 128, 136), (8728 // 8, 256, 1152, 320), (5, 64, 64, 64), (200, 4096, 256, 4096)]:
        z = gen(R, ld, seed=7).to(dtype)
        zd = z.to(DEV)
        o2 = torch.full((C, Rp), 3.0, dtype=dtype, device=DEV)
        ops.transpose(zd[:, :C], o2, R, C, Rp)
        assert torch.equal(o2[:, :R].cpu(), z[:, :C].t()) and float(o2[:, R:].float().abs().sum()) == 0
    y = gen(37, 64, seed=2)
    yb = torch.empty(37, 64, dtype=BF, device=DEV)
    ops.cast(y.to(DEV), yb)
    assert torch.equal(yb.cpu(), y.to(BF))


@pytest.mark.parametrize("act", [torch.float32, BF])
@pytest.mark.parametrize("rows,dim", [(45, 512), (19, 4096), (9, 6144), (33, 130), (8, 1028)])
def test_rmsnorm_bwd(act, rows, dim):
    """vector kernel (dim % 4 == 0; 4 or 8 float4 per thread) and the scalar fallback (dim = 130)"""
    x = gen(rows, dim, seed=3, scale=2.0).requires_grad_(True)
    w = (1 + 0.1 * gen(dim, seed=4)).requires_grad_(True)
    dy = gen(rows, dim, seed=5).to(act)
    y = ref_cpu.rmsnorm(x, w, 1e-5)
    y.backward(dy.float())
    dh0 = gen(rows, dim, seed=6)
    dh = dh0.to(DEV).clone()
    dw = torch.zeros(dim, device=DEV)
    ops.rmsnorm_bwd(x.detach().to(DEV), w.detach().to(DEV), dy.to(DEV), dh, dw, 1e-5)
    assert relerr(dh.cpu() - dh0, x.grad) < 1e-4
    assert relerr(dw, w.grad) < 1e-4
    if dim % 4 == 0:
        # the form that also emits the bf16 copy of the updated stream gradient (the next GEMMs' operand): same dh, same dw, and
        # the copy is exactly the bf16 rounding of it -- into a wider buffer (row stride != dim), as the adapter path has it
        dh2, dw2 = dh0.to(DEV).clone(), torch.zeros(dim, device=DEV)
        wide = torch.full((rows, dim + 64), 7.0, dtype=BF, device=DEV)
        ops.rmsnorm_bwd(x.detach().to(DEV), w.detach().to(DEV), dy.to(DEV), dh2, dw2, 1e-5, dh_lowp=wide[:, :dim])
        assert torch.equal(dh2, dh) and torch.allclose(dw2, dw, rtol=1e-5, atol=1e-5)      # (dw: partial sums added in no fixed order)
        assert torch.equal(wide[:, :dim], dh.to(BF)) and bool((wide[:, dim:] == 7.0).all())


@pytest.mark.parametrize("act", [torch.float32, BF])
def test_layernorm_bwd(act):
    rows, dim = 40, 256
    x = gen(rows, dim, seed=7, scale=2.0).to(act)
    w, b = 1 + 0.1 * gen(dim, seed=8), 0.1 * gen(dim, seed=9)
    xr = x.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (dim,), wr, br, 1e-5).backward(gen(rows, dim, seed=10))
    perm = torch.randperm(rows * 2, generator=torch.Generator().manual_seed(1))[:rows].to(torch.int32)
    dy_big = torch.zeros(rows * 2, dim)
    dy_big[perm.long()] = gen(rows, dim, seed=10)
    dx = torch.empty(rows, dim, dtype=act, device=DEV)
    dw, db = torch.zeros(dim, device=DEV), torch.zeros(dim, device=DEV)
    ops.layernorm_bwd(x.to(DEV), w.to(DEV), dy_big.to(DEV), perm.to(DEV), dx, dw, db)
    tol = 1e-4 if act == torch.float32 else 1e-2
    assert relerr(dx, xr.grad) < tol and relerr(dw, wr.grad) < 1e-4 and relerr(db, br.grad) < 1e-4
    # forward with fp32 params writing fp32 rows (projector under autocast)
    yb = torch.zeros(rows * 2, dim, device=DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), yb, row_map=perm.to(DEV))
    assert relerr(yb[perm.long().to(DEV)], F.layer_norm(x.float(), (dim,), w, b, 1e-5)) < 1e-5


@pytest.mark.parametrize("inter", [False, True])
def test_swiglu_fwd_bwd(inter):
    rows, Fd = 33, 64
    g, u = gen(rows, Fd, seed=11).requires_grad_(True), gen(rows, Fd, seed=12).requires_grad_(True)
    da = gen(rows, Fd, seed=13)
    (F.silu(g) * u).backward(da)
    if inter:
        gu = torch.stack([g.detach().view(rows, Fd // 16, 16), u.detach().view(rows, Fd // 16, 16)], dim=2).reshape(rows, 2 * Fd)
    else:
        gu = torch.cat([g.detach(), u.detach()], dim=1)
    gud = gu.to(DEV)
    act = torch.empty(rows, Fd, device=DEV)
    ops.swiglu_fwd(gud, act, Fd, inter)
    assert relerr(act, F.silu(g) * u) < 1e-5
    dgu = torch.empty(rows, 2 * Fd, device=DEV)
    ops.swiglu_bwd(gud, da.to(DEV), dgu, Fd, inter)
    if inter:
        d = dgu.cpu().view(rows, Fd // 16, 2, 16)
        dg, du = d[:, :, 0].reshape(rows, Fd), d[:, :, 1].reshape(rows, Fd)
    else:
        dg, du = dgu.cpu()[:, :Fd], dgu.cpu()[:, Fd:]
    assert relerr(dg, g.grad) < 1e-5 and relerr(du, u.grad) < 1e-5


def test_rope_bwd_pack():
    B, S, H, Hkv, hd = 2, 7, 4, 2, 16
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    fc = ref_cpu.precompute_freqs_cis(hd, 64)
    q, k = gen(B, S, H, hd, seed=14).requires_grad_(True), gen(B, S, Hkv, hd, seed=15).requires_grad_(True)
    oq, ok = ref_cpu.apply_rotary_emb(q, k, fc[:S])
    dq, dk, dv = gen(B, S, H, hd, seed=16), gen(B, S, Hkv, hd, seed=17), gen(B, S, Hkv, hd, seed=18)
    (oq * dq).sum().backward(retain_graph=True)
    (ok * dk).sum().backward()
    dqkv = torch.empty(B * S, (H + 2 * Hkv) * hd, device=DEV)
    ops.rope_bwd_pack(dq.to(DEV), dk.permute(0, 2, 1, 3).contiguous().to(DEV), dv.permute(0, 2, 1, 3).contiguous().to(DEV), dqkv,
                      precompute_cos_sin(hd, 64, 10000.0, None).to(DEV), B, S, H, Hkv, hd, 0)
    d = dqkv.cpu()
    assert relerr(d[:, :H * hd].view(B, S, H, hd), q.grad) < 1e-5
    assert relerr(d[:, H * hd:(H + Hkv) * hd].view(B, S, Hkv, hd), k.grad) < 1e-5
    assert torch.equal(d[:, (H + Hkv) * hd:].view(B, S, Hkv, hd), dv)


@pytest.mark.parametrize("dtype,B,S,H,Hkv,hd,causal,mfma", [
    (torch.float32, 2, 9, 4, 2, 16, True, False), (torch.float32, 1, 70, 2, 2, 64, False, False),
    (BF, 2, 40, 2, 1, 128, True, False), (torch.float32, 1, 33, 2, 2, 128, True, False),
    (BF, 2, 40, 2, 1, 128, True, True), (BF, 1, 300, 4, 2, 128, True, True), (BF, 2, 200, 2, 2, 64, False, True),
    (BF, 1, 577, 2, 2, 64, False, True), (BF, 2, 129, 2, 2, 128, True, True),
    # (round 5) the lengths of geometry R / the reference recipe / configs[4]: S = 1091, 1967, 2048 (GQA), 2182 -- the one-pass dK + dV
    # kernel and the dQ kernel against oracle autograd at 9 ... 18 key blocks, not only against each other
    (BF, 1, 1091, 2, 2, 128, True, True), (BF, 1, 1967, 2, 2, 128, True, True), (BF, 1, 2048, 4, 2, 128, True, True),
    (BF, 1, 2182, 2, 2, 128, True, True)])
def test_attention_lse_and_bwd(dtype, B, S, H, Hkv, hd, causal, mfma):
    q = gen(B, S, H, hd, seed=19).to(dtype).float().requires_grad_(True)
    k = gen(B, S, Hkv, hd, seed=20).to(dtype).float().requires_grad_(True)
    v = gen(B, S, Hkv, hd, seed=21).to(dtype).float().requires_grad_(True)
    do = gen(B, S, H, hd, seed=22).to(dtype).float()
    n_rep = H // Hkv
    kk, vv = ref_cpu.repeat_kv(k, n_rep).transpose(1, 2), ref_cpu.repeat_kv(v, n_rep).transpose(1, 2)
    mask = ref_cpu.make_causal_mask(S, S) if causal else None
    out = ref_cpu.sdpa(q.transpose(1, 2), kk, vv, mask).transpose(1, 2)
    out.backward(do)
    spad = (S + 63) // 64 * 64
    qd = q.detach().to(dtype).to(DEV).contiguous()
    kc = torch.zeros(B, Hkv, spad, hd, dtype=dtype, device=DEV)
    vc = torch.zeros(B, Hkv, hd, spad, dtype=dtype, device=DEV)
    kc[:, :, :S] = k.detach().to(dtype).permute(0, 2, 1, 3).to(DEV)
    vc[:, :, :, :S] = v.detach().to(dtype).permute(0, 2, 3, 1).to(DEV)
    o = torch.empty(B, S, H, hd, dtype=dtype, device=DEV)
    lse = torch.empty(B, H, S, device=DEV)
    strides = (S * H * hd, H * hd, hd, Hkv * spad * hd, spad * hd, hd, Hkv * hd * spad, hd * spad, spad, S * H * hd, H * hd, hd)
    ops.attention_lse(qd, kc, vc, o, lse, B, S, S, H, Hkv, hd, strides, causal)
    sc = torch.matmul(q.detach().transpose(1, 2), kk.detach().transpose(-1, -2)) / math.sqrt(hd)
    if causal:
        sc = sc.masked_fill(~mask, float("-inf"))
    assert relerr(lse, torch.logsumexp(sc, dim=-1)) < (1e-5 if dtype == torch.float32 else 2e-2)
    vrows = v.detach().to(dtype).to(DEV).contiguous()      # [B,S,Hkv,hd]
    dq = torch.empty(B, S, H, hd, dtype=dtype, device=DEV)
    dk = torch.empty(B, Hkv, S, hd, dtype=dtype, device=DEV)
    dv = torch.empty(B, Hkv, S, hd, dtype=dtype, device=DEV)
    D = torch.empty(B, S, H, device=DEV)
    ws = torch.empty(ops.attention_bwd_workspace_bytes(B, S, H, Hkv, hd), dtype=torch.uint8, device=DEV) if mfma else None
    ops.attention_bwd(qd, kc, Hkv * spad * hd, spad * hd, vrows, S * Hkv * hd, Hkv * hd, hd, o, do.to(dtype).to(DEV), lse, D,
                      dq, dk, dv, B, S, H, Hkv, hd, causal, workspace=ws)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert relerr(dq, q.grad) < tol
    assert relerr(dk.permute(0, 2, 1, 3), k.grad) < tol
    assert relerr(dv.permute(0, 2, 1, 3), v.grad) < tol


def test_embed_bwd_and_rows_sum():
    B, T, W, dim, V = 2, 5, 3, 64, 30
    tok = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(4))
    tok[1, 2] = tok[0, 1]        # repeated token -> accumulation
    dh = gen(B * (T + W), dim, seed=23)
    dt_ = torch.zeros(V, dim, device=DEV)
    ops.embed_bwd(tok.to(DEV), dh.to(DEV), dt_, B, T, W, dim)
    want = torch.zeros(V, dim)
    dv = dh.view(B, T + W, dim)
    for b in range(B):
        for t in range(T):
            want[tok[b, t]] += dv[b, 0 if t == 0 else W + t]
    assert relerr(dt_, want) < 1e-6
    idx = torch.tensor([0, 3, 9], dtype=torch.int32)
    out = torch.ones(dim, device=DEV)
    ops.rows_sum(dh.to(DEV), idx.to(DEV), 3, out)
    assert relerr(out, 1 + dh[idx.long()].sum(0)) < 1e-6


# ------------------------------------------------------------------ full training step vs oracle autograd
TK = dict(dim=64, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=192, multiple_of=64, max_seq_len=1024)


def oracle_loss_and_grads(sd, vsd, ex, lab, img):
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    vtrain = {k: v.clone().requires_grad_(k.startswith(("visual_proj", "start_img", "end_img"))) for k, v in (vsd or {}).items()}
    oargs = ref_cpu.OracleArgs(**TK)
    dec = ref_cpu.OracleDecoder(oargs, sd)
    itok = None
    if img is not None:
        views = ref_cpu.encode_image(img, vtrain, vit_layers=2, vit_heads=4, n_views=1)
        itok = ref_cpu.assemble_image_tokens(views, vtrain["start_img"], vtrain["end_img"])
    loss = ref_cpu.meta_forward_loss(dec, ex, lab, itok)
    loss.backward()
    grads = {k: v.grad for k, v in sd.items()}
    grads.update({k: v.grad for k, v in vtrain.items() if v.requires_grad})
    return float(loss), grads


def build(with_visual, compute_dtype):
    args = plugin.ModelArgs(**TK, vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)
    m = plugin.Transformer(args, with_visual=with_visual)
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(**TK), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=64, layers=2, patch=14, grid=8, seed=1, std=0.05) if with_visual else None
    m.load_state_dict({**sd, **(vsd or {})})
    for n, p in m.named_parameters():
        p.requires_grad = not n.startswith("clip.")
    m.to(compute_dtype).to(DEV)
    promote_trainable_params_to_fp32(m)
    return m, sd, vsd


@pytest.mark.parametrize("with_visual,recompute", [(False, True), (True, True), (True, False)])
def test_train_step_fp32_matches_autograd(with_visual, recompute):
    m, sd, vsd = build(with_visual, torch.float32)
    g = torch.Generator().manual_seed(5)
    B, T = 2, 12
    ex = torch.randint(3, 192, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :4] = 0
    lab[1, 9:] = 0
    img = synth_image(B, size=112, seed=3) if with_visual else None
    want_loss, want = oracle_loss_and_grads(sd, vsd, ex, lab, img)
    eng = TrainEngine(m, torch.float32, recompute=recompute)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV) if with_visual else None)
    assert abs(float(loss) - want_loss) < 1e-3 * abs(want_loss)
    eng.backward(1.0)
    for name, p in m.get_trainable_params().items():
        assert p.grad is not None, name
        e = relerr(p.grad, want[name])
        assert e < 1e-3, (name, e)
    # accumulation + grad scale: a second backward with scale 0.5 adds half the gradient
    eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV) if with_visual else None)
    eng.backward(0.5)
    assert relerr(m.layers[0].feed_forward.w2.weight.grad, 1.5 * want["layers.0.feed_forward.w2.weight"]) < 1e-3
    assert relerr(m.tok_embeddings.weight.grad, 1.5 * want["tok_embeddings.weight"]) < 1e-3
    # zero_grad(set_to_none=True) then a new step: the big matrices are WRITTEN by their first weight-gradient GEMM (their
    # storage still holds the previous step's values), everything else is re-zeroed -> exactly one step's gradient again
    for p in m.parameters():
        p.grad = None
    eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV) if with_visual else None)
    eng.backward(1.0)
    for name, p in m.get_trainable_params().items():
        assert relerr(p.grad, want[name]) < 1e-3, name


def test_train_step_bf16_close_to_fp32_reference():
    """autocast-style step (bf16 activations/GEMMs, fp32 masters & grads): bf16-level agreement with the
    fp32 reference gradient (cosine similarity, since individual small entries carry bf16 noise)."""
    m, sd, vsd = build(True, BF)
    g = torch.Generator().manual_seed(6)
    B, T = 2, 16
    ex = torch.randint(3, 192, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :5] = 0
    img = synth_image(B, size=112, seed=4)
    want_loss, want = oracle_loss_and_grads(sd, vsd, ex, lab, img)
    eng = TrainEngine(m, BF)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV))
    assert abs(float(loss) - want_loss) < 2e-2 * abs(want_loss)
    eng.backward(1.0)
    for name, p in m.get_trainable_params().items():
        a, b = p.grad.float().cpu().flatten(), want[name].flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-20))
        assert cos > 0.99, (name, cos)


def test_train_step_bf16_tn_weight_gradients():
    """dim 256: every decoder matrix takes the a3v_gemm_tn weight-gradient path (ragged token count, fused qkv / w13 views,
    fresh store then accumulate).  Same gradients as the transpose + NT path to fp32 summation order, and bf16-level
    agreement with the oracle's autograd."""
    big = dict(dim=256, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=512, multiple_of=256, max_seq_len=256)
    oargs = ref_cpu.OracleArgs(**big)
    sd = ref_cpu.make_decoder_weights(oargs, seed=2, std=0.05)
    g = torch.Generator().manual_seed(8)
    B, T = 3, 37
    ex = torch.randint(3, 512, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :6] = 0
    got = {}
    for tn in (True, False):
        m = plugin.Transformer(plugin.ModelArgs(**big), with_visual=False)
        m.load_state_dict(sd)
        m.to(BF).to(DEV)
        promote_trainable_params_to_fp32(m)
        eng = TrainEngine(m, BF)
        eng.tn_wgrad = eng.fuse_qkv_rope = eng.nn_dgrad = tn      # also: fused qkv / RoPE / cache epilogue, NN input gradients
        # (packed_attn_bwd stays on in both: it moves a bf16 rounding point, see test_attention_bwd_packed_equals_bwd_then_rope_pack)
        for scale in (1.0, 0.5):
            eng.forward_loss(ex.to(DEV), lab.to(DEV), None)
            eng.backward(scale)
        got[tn] = {n: p.grad.float().cpu() for n, p in m.get_trainable_params().items()}
    for n in got[True]:
        assert relerr(got[True][n], got[False][n]) < 1e-5, n
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = ref_cpu.meta_forward_loss(ref_cpu.OracleDecoder(oargs, sdg), ex, lab, None)
    loss.backward()
    for n, gr in got[True].items():
        a, b = gr.flatten(), 1.5 * sdg[n].grad.flatten()
        assert float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-20)) > 0.99, n
        assert abs(float(a.norm() / b.norm()) - 1) < 0.05, n


def test_metamodel_loss_backward_drop_in(golden_dir=None):
    """loss, _ = model(examples, labels); loss.backward() -- the trainer's call sequence (engine_finetune.py:50-68)."""
    import os
    gd = os.path.join(os.path.dirname(__file__), "golden")
    mm = MetaModel("llama_ens5", os.path.join(gd, "tiny_params.json"), os.path.join(gd, "tokenizer.model"), with_visual=False, max_seq_len=64)
    from oracle.gen_golden import TINY
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=192, **TINY), seed=0, std=0.08)
    mm.llma.load_state_dict(sd)
    mm.to(DEV)
    mm.train_compute_dtype = torch.float32
    g = torch.Generator().manual_seed(7)
    ex = torch.randint(3, 192, (3, 14), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :5] = 0
    lab[:, 11:] = 0          # trailing pad columns are trimmed by MetaModel.forward
    loss, extra = mm(ex.to(DEV), lab.to(DEV))
    assert extra == {}
    (loss / 2).backward()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=192, **TINY), sdg)
    want = ref_cpu.meta_forward_loss(dec, ex, lab)
    (want / 2).backward()
    assert abs(float(loss) - float(want)) < 1e-3 * float(want)
    for name, p in mm.llma.named_parameters():
        assert relerr(p.grad, sdg[name].grad) < 1e-3, name
    opt = torch.optim.AdamW(mm.parameters(), lr=1e-3, betas=(0.9, 0.95))
    opt.step()
    opt.zero_grad()
    loss2, _ = mm(ex.to(DEV), lab.to(DEV))
    assert float(loss2) < float(loss)       # one AdamW step on the same batch reduces the loss


@pytest.mark.parametrize("B,S,H,Hkv,hd", [(2, 150, 4, 4, 128), (1, 77, 4, 2, 64), (2, 1091, 4, 4, 128)])
def test_attention_bwd_packed_equals_bwd_then_rope_pack(B, S, H, Hkv, hd):
    """a3v_attention_bwd_packed == a3v_attention_bwd + a3v_rope_bwd_pack: the only difference allowed is the rounding point
    (the packed form rotates the fp32 accumulator and rounds once; the two-pass form rounds dq / dk to bf16 first)."""
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    g = torch.Generator().manual_seed(B * 100 + S)
    sp = (S + 63) // 64 * 64
    q = (torch.randn(B, S, H, hd, generator=g)).to(BF).to(DEV)
    kc = torch.zeros(B, Hkv, sp, hd, dtype=BF, device=DEV)
    kc[:, :, :S] = torch.randn(B, Hkv, S, hd, generator=g).to(BF).to(DEV)
    v = torch.randn(B, S, Hkv, hd, generator=g).to(BF).to(DEV)
    vt = torch.zeros(B, Hkv, hd, sp, dtype=BF, device=DEV)
    vt[:, :, :, :S] = v.permute(0, 2, 3, 1)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, S, device=DEV)
    st = (S * H * hd, H * hd, hd, Hkv * sp * hd, sp * hd, hd, Hkv * hd * sp, hd * sp, sp, S * H * hd, H * hd, hd)
    ops.attention_lse(q, kc, vt, o, lse, B, S, S, H, Hkv, hd, st, True)
    do = torch.randn(B, S, H, hd, generator=g).to(BF).to(DEV)
    cs = precompute_cos_sin(hd, 2 * sp, 10000.0, None).to(DEV)
    D = torch.empty(B, S, H, device=DEV)
    N = (H + 2 * Hkv) * hd
    dq = torch.empty_like(q)
    dk = torch.empty(B, Hkv, S, hd, dtype=BF, device=DEV)
    dv = torch.empty_like(dk)
    ws = torch.empty(ops.attention_bwd_workspace_bytes(B, S, H, Hkv, hd), dtype=torch.uint8, device=DEV)
    ops.attention_bwd(q, kc, Hkv * sp * hd, sp * hd, v, S * Hkv * hd, Hkv * hd, hd, o, do, lse, D, dq, dk, dv, B, S, H, Hkv, hd, True, workspace=ws)
    ref = torch.zeros(B * S, N + 8, dtype=BF, device=DEV)
    ops.rope_bwd_pack(dq, dk, dv, ref[:, :N], cs, B, S, H, Hkv, hd, 0)
    got = torch.full((B * S, N + 8), 7.0, dtype=BF, device=DEV)
    ops.attention_bwd_packed(q, kc, Hkv * sp * hd, sp * hd, v, S * Hkv * hd, Hkv * hd, hd, o, do, lse, D, got[:, :N], cs, B, S, H, Hkv, hd, True, 0)
    assert float((got[:, N:].float() - 7.0).abs().max()) == 0                   # nothing written past the row
    a, b = got[:, :N].float(), ref[:, :N].float()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2 ** -6 * scale
    assert torch.equal(got[:, (H + Hkv) * hd:N], ref[:, (H + Hkv) * hd:N])       # dv is not rotated: identical


@pytest.mark.gpu
def test_fused_adamw_grad_scale_equals_scaled_gradients():
    """FusedAdamW.step(grad_scale=coef) (a3v_adamw_scaled: the clip coefficient applied as the gradient is read) is the same
    update, bit for bit, as grad.mul_(coef) followed by step() (util/clip_grad.py:187-193 + engine_finetune.py:63)."""
    import copy
    from a3vlm_amd.optim import FusedAdamW
    from a3vlm_amd.dp import clip_grad_norm
    g = torch.Generator(device=DEV).manual_seed(5)
    shapes = [(257, 131), (64,), (1000, 7)]
    pa = [torch.nn.Parameter(torch.randn(*s, device=DEV, generator=g)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = FusedAdamW(pa, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1)
    ob = FusedAdamW(pb, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1)
    for it in range(3):
        grads = [torch.randn(*s, device=DEV, generator=g) * (10.0 if it == 0 else 1e-4) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad = gr.clone()
            q.grad = gr.clone()
        na, coef = clip_grad_norm(pa, 1.0, defer=True)
        assert coef.dtype == torch.float32 and coef.is_cuda
        oa.step(grad_scale=coef)
        nb = clip_grad_norm(pb, 1.0)               # scales pb's gradients in place
        ob.step()
        assert torch.equal(na, nb)
        assert (float(coef) < 1.0) == (it == 0)
        for p, q in zip(pa, pb):
            assert torch.equal(p, q)
            assert torch.equal(oa.state[p]["exp_avg"], ob.state[q]["exp_avg"])
            assert torch.equal(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"])


@pytest.mark.gpu
def test_clip_grad_norm_on_flat_buffer_matches_per_parameter_form():
    from a3vlm_amd.dp import clip_grad_norm
    g = torch.Generator(device=DEV).manual_seed(6)
    flat = torch.randn(5000, device=DEV, generator=g)
    views = [flat[0:1200].view(30, 40), flat[1200:1264], flat[2048:5000].view(-1, 8)]
    flat[1264:2048] = 0                                    # padding between views is zero in the engine's buffer
    ps = [torch.nn.Parameter(torch.zeros_like(v)) for v in views]
    for p, v in zip(ps, views):
        p.grad = v
    ref = [torch.nn.Parameter(torch.zeros_like(v)) for v in views]
    for p, v in zip(ref, views):
        p.grad = v.clone()
    n_ref = clip_grad_norm(ref, 0.5)
    n_flat, coef = clip_grad_norm(ps, 0.5, flat=flat, defer=True)
    assert abs(float(n_flat) - float(n_ref)) <= 1e-5 * float(n_ref)
    assert abs(float(coef) - 0.5 / (float(n_ref) + 1e-6)) <= 1e-6
    before = flat.clone()
    n2 = clip_grad_norm(ps, 0.5, flat=flat)                # in-place form on the flat buffer
    assert torch.equal(n2, n_flat)
    assert torch.allclose(flat, before * coef, rtol=1e-6, atol=0)
    for p, q in zip(ps, ref):
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-8)


def test_train_step_with_hook_fed_encoder_streams_fp32():
    """The reference llama_ens5 configuration trains visual_proj over the concat [CLIP | ConvNeXt | DINOv2] and qformer_proj over the
    Q-Former tokens (LLM/llama_ens5.py:325-349, 436-458); the three extra encoders are frozen inputs.  One fp32 training step with
    ``extra_feat_dim`` and ``qformer_tokens`` set (features injected as the provider hooks would) against oracle autograd:
    loss and every trainable gradient, the two projectors' included."""
    X1, X2, Q = 48, 32, 5
    args = plugin.ModelArgs(**TK, vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1, extra_feat_dim=X1 + X2, qformer_tokens=Q)
    m = plugin.Transformer(args, with_visual=True)
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(**TK), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=64, layers=2, patch=14, grid=8, in_feat=64 + X1 + X2, with_qformer=True, seed=1, std=0.05)
    m.load_state_dict({**sd, **vsd})
    for n, p in m.named_parameters():
        p.requires_grad = not n.startswith("clip.")
    m.to(DEV)
    promote_trainable_params_to_fp32(m)
    assert m.image_words == (Q + 65 + 2)
    g = torch.Generator().manual_seed(6)
    B, T = 2, 11
    ex = torch.randint(3, 192, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :3] = 0
    img = synth_image(B, size=112, seed=3)
    qf = torch.randn(B, Q, 768, generator=g)
    extra = [torch.randn(B, 65, X1, generator=g), torch.randn(B, 65, X2, generator=g)]
    # oracle
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    vg = {k: v.clone().requires_grad_(k.startswith(("visual_proj", "qformer_proj", "start_img", "end_img"))) for k, v in vsd.items()}
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(**TK), sdg)
    views = ref_cpu.encode_image(img, vg, vit_layers=2, vit_heads=4, n_views=1, qformer_feats=qf, extra_feats=extra)
    want_loss = ref_cpu.meta_forward_loss(dec, ex, lab, ref_cpu.assemble_image_tokens(views, vg["start_img"], vg["end_img"]))
    want_loss.backward()
    want = {k: v.grad for k, v in sdg.items()}
    want.update({k: v.grad for k, v in vg.items() if v.requires_grad})
    eng = TrainEngine(m, torch.float32, recompute=False)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV), qformer_feats=qf.to(DEV), extra_feats=[e.to(DEV) for e in extra])
    assert abs(float(loss) - float(want_loss)) < 1e-3 * abs(float(want_loss))
    eng.backward(1.0)
    names = set(m.get_trainable_params())
    assert {"qformer_proj.0.weight", "qformer_proj.1.bias", "visual_proj.0.weight"} <= names
    for name, p in m.get_trainable_params().items():
        assert p.grad is not None, name
        assert relerr(p.grad, want[name]) < 1e-3, (name, relerr(p.grad, want[name]))
    with pytest.raises(ValueError):
        eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV))            # features missing and no provider hook attached


@pytest.mark.parametrize("accum", [1, 2])
def test_grad_square_sums_from_the_weight_gradient_epilogues(accum):
    """One rank: the clip's norm is assembled from (a) the partial sums the big weight-gradient GEMMs leave (a3v_gemm_tn_sumsq:
    fast epilogue tiles, ragged tiles, split-K reduce pass) and (b) a3v_sumsq_partials over what is left of each bucket -- equal to
    the norm of the flat buffer, over two steps (slots are re-zeroed), with a non-boundary accumulation micro-step ignored."""
    from a3vlm_amd.dp import GradSquareSums, clip_grad_norm
    big = dict(dim=512, n_layers=2, n_heads=4, n_kv_heads=4, vocab_size=1024, multiple_of=256, max_seq_len=512)
    oargs = ref_cpu.OracleArgs(**big)
    sd = ref_cpu.make_decoder_weights(oargs, seed=3, std=0.05)
    m = plugin.Transformer(plugin.ModelArgs(**big), with_visual=False)
    m.load_state_dict(sd)
    m.to(BF).to(DEV)
    promote_trainable_params_to_fp32(m)
    eng = TrainEngine(m, BF)
    sq = GradSquareSums(eng)
    assert eng.sumsq_sink is sq
    g = torch.Generator().manual_seed(9)
    B, T = 4, 301                                           # 1204 rows: whole tiles, ragged tiles and a split tail in the TN GEMMs
    params = [p for p in m.parameters() if p.requires_grad]
    for step in range(2):
        for micro in range(accum):
            ex = torch.randint(3, 1024, (B, T), generator=g)
            ex[:, 0] = 1
            sq.enabled = micro == accum - 1
            eng.forward_loss(ex.to(DEV), ex.to(DEV), None)
            eng.backward(1.0 / accum)
        assert sq._covered, "no weight-gradient GEMM handed over its sums"
        norm, coef = clip_grad_norm(params, 8.0, flat=eng.flat_grads(), defer=True, sumsq=sq)
        ref = torch.linalg.vector_norm(eng.flat_grads().double()).float()
        assert torch.allclose(norm, ref, rtol=2e-5), (step, float(norm), float(ref))
        for p in params:
            p.grad = None


@pytest.mark.gpu
@pytest.mark.parametrize("with_visual", [False, True])
def test_overlapped_optimizer_step_equals_in_line_step(with_visual):
    """FusedAdamW.step(overlap=True) runs the update on its own stream in forward order and the next forward waits bucket by
    bucket.  Two identical models, one stepped in line and one overlapped, fed the SAME gradients (the backward's atomics are not
    bit-reproducible, and Adam amplifies that): parameters, optimizer state, weight images and the next forward's loss stay
    bit-equal over three steps."""
    from a3vlm_amd.dp import clip_grad_norm
    from a3vlm_amd.optim import FusedAdamW
    big = dict(dim=512, n_layers=3, n_heads=4, n_kv_heads=4, vocab_size=1024, multiple_of=256, max_seq_len=512)
    g = torch.Generator().manual_seed(21)
    B, T = 2, 130
    exs = [torch.randint(3, 192 if with_visual else 1024, (B, T), generator=g) for _ in range(3)]
    img = torch.randn(B, 3, 112, 112, generator=g).to(DEV).to(BF) if with_visual else None
    side = []
    for overlap in (False, True):
        if with_visual:
            m, _, _ = build(True, BF)
        else:
            m = plugin.Transformer(plugin.ModelArgs(**big), with_visual=False)
            m.load_state_dict(ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(**big), seed=3, std=0.05))
            m.to(BF).to(DEV)
        promote_trainable_params_to_fp32(m)
        eng = TrainEngine(m, BF)
        params = [p for p in m.parameters() if p.requires_grad]
        side.append((m, eng, params, FusedAdamW(params, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1, engine=eng)))
    (_, ea, pa, oa), (_, eb, pb, ob) = side
    for ex in exs:
        ex = ex.to(DEV)
        ex[:, 0] = 1
        la = ea.forward_loss(ex, ex, img).clone()
        lb = eb.forward_loss(ex, ex, img).clone()          # waits, bucket by bucket, for the update still running on ob's stream
        assert not eb._weights_ready
        assert float(la) == float(lb), (float(la), float(lb))
        ea.backward(1.0)
        eb.backward(1.0)
        eb.flat_grads().copy_(ea.flat_grads())
        _, coef = clip_grad_norm(pa, 0.5, flat=ea.flat_grads(), defer=True)
        oa.step(grad_scale=coef)
        ob.step(grad_scale=coef, overlap=True)
        assert eb._weights_ready, "the overlapped step left no events for the next forward"
        oa.zero_grad(set_to_none=True)
        ob.zero_grad(set_to_none=True)
    eb.sync_optimizer()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
        assert torch.equal(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]) and float(oa.state[a]["step"]) == float(ob.state[b]["step"]) == 3
    assert torch.equal(ea._images()["qkv.1"], eb._images()["qkv.1"])


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,row0,ld", [(256, 192, 0, 256), (128, 128, 64, 320), (192, 64, 128, 324), (1024, 512, 512, 2048 + 64)])
def test_adamw_with_transposed_image_equals_adamw_then_transpose(rows, cols, row0, ld):
    """a3v_adamw_scaled_t (round 5): same fp32 update and forward image as a3v_adamw_scaled, bit for bit, and the parameter's columns of
    the transposed image [cols][ld] = the forward image transposed; nothing else of that image is touched; a negative scale is a no-op."""
    from a3vlm_amd import lib as _l
    lib = _l.load()
    g = torch.Generator().manual_seed(rows + cols)
    p = torch.randn(rows, cols, generator=g).to(DEV)
    gr = (torch.randn(rows, cols, generator=g) * 0.1).to(DEV)
    m = (torch.randn(rows, cols, generator=g) * 0.01).to(DEV)
    v = (torch.rand(rows, cols, generator=g) * 1e-3).to(DEV)
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    img, img2 = torch.zeros(rows, cols, device=DEV, dtype=BF), torch.zeros(rows, cols, device=DEV, dtype=BF)
    wt = torch.full((cols, ld), 7.0, device=DEV, dtype=BF)
    view = wt[:, row0:row0 + rows]
    st = torch.cuda.current_stream().cuda_stream
    for step, scale in ((1, 0.7), (2, 1.0), (3, -1.0)):
        gs = torch.tensor([scale], device=DEV)
        assert lib.a3v_adamw_scaled(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 1e-2, 0.9, 0.95, 1e-8, 0.1, step,
                                    img.data_ptr(), gs.data_ptr(), st) == 0
        assert lib.a3v_adamw_scaled_t(p2.data_ptr(), gr.data_ptr(), m2.data_ptr(), v2.data_ptr(), rows, cols, 1e-2, 0.9, 0.95, 1e-8, 0.1, step,
                                      img2.data_ptr(), view.data_ptr(), wt.stride(0), gs.data_ptr(), st) == 0
        torch.cuda.synchronize()
        assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2) and torch.equal(img, img2)
        assert torch.equal(view, img.t())
    rest = wt.clone()
    rest[:, row0:row0 + rows] = 7.0
    assert bool((rest == 7.0).all())
    # shapes the tiled kernel does not take are refused, not mangled
    assert lib.a3v_adamw_scaled_t(p2.data_ptr(), gr.data_ptr(), m2.data_ptr(), v2.data_ptr(), rows - 1, cols, 1e-2, 0.9, 0.95, 1e-8, 0.1, 1,
                                  img2.data_ptr(), view.data_ptr(), wt.stride(0), None, st) != 0


@pytest.mark.gpu
def test_full_fine_tune_nt_input_gradients_keep_their_transposed_images_current():
    """Full fine-tune default (round 5): input gradients on the NT kernel over W^T images that FusedAdamW re-writes in its own pass
    (a3v_adamw_scaled_t).  Two identical models -- NN input gradients on the forward images / NT on the transposed ones -- fed the
    SAME gradients: parameters and losses stay bit-equal over three steps, every transposed image equals its forward image after
    every update, and no a3v_transpose runs after the first backward built them."""
    from a3vlm_amd.dp import clip_grad_norm
    from a3vlm_amd.optim import FusedAdamW
    big = dict(dim=512, n_layers=2, n_heads=4, n_kv_heads=4, vocab_size=1024, multiple_of=256, max_seq_len=512)
    g = torch.Generator().manual_seed(33)
    B, T = 2, 96
    exs = [torch.randint(3, 1024, (B, T), generator=g) for _ in range(3)]
    side = []
    for nn in (True, False):
        m = plugin.Transformer(plugin.ModelArgs(**big), with_visual=False)
        m.load_state_dict(ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(**big), seed=5, std=0.05))
        m.to(BF).to(DEV)
        promote_trainable_params_to_fp32(m)
        eng = TrainEngine(m, BF)
        assert eng.nn_dgrad is False or "A3V_NN_DGRAD" in os.environ      # the default of a plain (non-ZeRO-1, non-LoRA) engine
        eng.nn_dgrad = nn
        params = [p for p in m.parameters() if p.requires_grad]
        side.append((eng, params, FusedAdamW(params, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1, engine=eng)))
    (ea, pa, oa), (eb, pb, ob) = side
    calls = {"n": 0}
    real = ops.transpose

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    ops.transpose = counting
    try:
        after_first = None
        for it, ex in enumerate(exs):
            ex = ex.to(DEV)
            ex[:, 0] = 1
            la, lb = ea.forward_loss(ex, ex, None).clone(), eb.forward_loss(ex, ex, None).clone()
            assert float(la) == float(lb)
            ea.backward(1.0)
            eb.backward(1.0)
            if it == 0:
                after_first = calls["n"]
                assert after_first > 0                     # the first backward built the transposed images
            rel = float((eb.flat_grads() - ea.flat_grads()).norm() / ea.flat_grads().norm())
            assert rel < 2e-3, rel                          # NN vs NT input gradients: same products, another accumulation order
            eb.flat_grads().copy_(ea.flat_grads())
            _, coef = clip_grad_norm(pa, 0.5, flat=ea.flat_grads(), defer=True)
            oa.step(grad_scale=coef)
            ob.step(grad_scale=coef)
            oa.zero_grad(set_to_none=True)
            ob.zero_grad(set_to_none=True)
            im = eb._images()
            for key in ("qkv.0", "wo.1", "w13.0", "w2.1", "out"):
                w, wt = im.store[key], im.store[key + ".t"]
                assert torch.equal(wt[:, :w.shape[0]], w.t()), key
        assert calls["n"] == after_first, "a transposed image was rebuilt by a3v_transpose after an optimizer step"
    finally:
        ops.transpose = real
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
