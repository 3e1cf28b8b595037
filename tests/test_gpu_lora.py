"""-m gpu: the LoRA plugin ``llama_ens5_peft`` (model/peft.py adapter semantics): forward / cached inference against the
oracle (fp32 parity path and bf16), a full training step (adapter + norm + projector gradients, frozen base) against oracle
autograd, trainability / checkpoint key names, and optimizer steps through MetaModel."""
import os

import numpy as np
import pytest
import torch

from a3vlm_amd.model.LLM import llama_ens5_peft as peft
from a3vlm_amd.model.meta import MetaModel
from a3vlm_amd.train import TrainEngine
from a3vlm_amd.util import promote_trainable_params_to_fp32
from oracle import ref_cpu
from oracle.gen_golden import synth_image

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
TK = dict(dim=128, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=192, multiple_of=64, max_seq_len=512)
RANK = 8


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def build(with_visual, dtype):
    args = peft.ModelArgs(**TK, lora_rank=RANK, vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)
    m = peft.Transformer(args, with_visual=with_visual)
    oargs = ref_cpu.OracleArgs(**TK)
    sd = ref_cpu.make_decoder_weights(oargs, seed=0, std=0.08)
    lsd = ref_cpu.make_lora_weights(oargs, RANK, seed=5, std_a=0.05, std_b=0.05)
    vsd = ref_cpu.make_vision_weights(128, width=64, layers=2, patch=14, grid=8, seed=1, std=0.05) if with_visual else {}
    res = m.load_state_dict({**sd, **lsd, **vsd}, strict=True)
    train = m.get_trainable_params()
    for n, p in m.named_parameters():
        p.requires_grad = n in train
    m.to(dtype).to(DEV)
    return m, oargs, sd, lsd, vsd


def test_trainable_set_and_keys():
    m, oargs, sd, lsd, vsd = build(True, torch.float32)
    tr = set(m.get_trainable_params())
    assert all(("lora_" in n or "norm" in n) for n in tr if n.startswith("layers."))
    assert "layers.0.attention.wq.lora_a.weight" in tr and "layers.1.feed_forward.w2.lora_b.weight" in tr
    assert "layers.0.attention.wq.weight" not in tr and "tok_embeddings.weight" not in tr and "output.weight" not in tr
    assert {"visual_proj.0.weight", "start_img", "end_img", "norm.weight"} <= tr and not any(n.startswith("clip.") for n in tr)
    assert m.is_peft and set(lsd) <= set(m.state_dict())
    fresh = peft.Transformer(peft.ModelArgs(**TK, lora_rank=RANK))
    assert float(fresh.layers[0].attention.wq.lora_b.weight.detach().abs().sum()) == 0.0          # peft.py:76
    assert 0.015 < float(fresh.layers[0].attention.wq.lora_a.weight.detach().std()) < 0.025         # trunc_normal_(std=.02), bounds +-2 absolute


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (BF, 4e-2)])
def test_lora_forward_and_decode_vs_oracle(dtype, tol):
    m, oargs, sd, lsd, _ = build(False, dtype)
    g = torch.Generator().manual_seed(3)
    B, T = 3, 21
    ex = torch.randint(3, 192, (B, T), generator=g)
    ex[:, 0] = 1
    cast = (lambda t: t.to(BF)) if dtype == BF else (lambda t: t)
    dec = ref_cpu.OracleDecoder(oargs, {k: cast(v) for k, v in {**sd, **lsd}.items()})
    want = dec.forward(ex).float()
    base_only = ref_cpu.OracleDecoder(oargs, {k: cast(v) for k, v in sd.items()}).forward(ex).float()
    assert float((want - base_only).abs().max()) > 10 * tol * float(want.abs().max()) / 10   # the adapters matter
    got = m(ex.to(DEV)).float()
    scale = float(want.abs().max())
    assert float((got.cpu() - want).abs().max()) / scale < tol
    w0 = dec.forward_inference(ex[:, :17], 0).float()
    l0 = m.forward_inference(ex[:, :17].to(DEV), 0).float().clone()
    assert float((l0.cpu() - w0).abs().max()) / scale < tol
    for t in range(17, 21):
        wt = dec.forward_inference(ex[:, t:t + 1], t).float()
        lt = m.forward_inference(ex[:, t:t + 1].to(DEV), t).float()
        assert float((lt.cpu() - wt).abs().max()) / scale < tol, t
        if dtype == torch.float32:
            assert (lt.argmax(-1).cpu() == wt.argmax(-1)).all()


@pytest.mark.parametrize("with_visual,recompute", [(False, True), (True, False)])
def test_lora_train_step_fp32_matches_autograd(with_visual, recompute):
    m, oargs, sd, lsd, vsd = build(with_visual, torch.float32)
    promote_trainable_params_to_fp32(m)
    g = torch.Generator().manual_seed(5)
    B, T = 2, 12
    ex = torch.randint(3, 192, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :4] = 0
    img = synth_image(B, size=112, seed=3) if with_visual else None
    tr = m.get_trainable_params()
    osd = {k: v.clone().requires_grad_(k in tr) for k, v in {**sd, **lsd}.items()}
    ovs = {k: v.clone().requires_grad_(k in tr) for k, v in vsd.items()}
    dec = ref_cpu.OracleDecoder(oargs, osd)
    itok = None
    if with_visual:
        views = ref_cpu.encode_image(img, ovs, vit_layers=2, vit_heads=4, n_views=1)
        itok = ref_cpu.assemble_image_tokens(views, ovs["start_img"], ovs["end_img"])
    want_loss = ref_cpu.meta_forward_loss(dec, ex, lab, itok)
    want_loss.backward()
    want = {k: v.grad for k, v in {**osd, **ovs}.items() if v.requires_grad}
    eng = TrainEngine(m, torch.float32, recompute=recompute)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV) if with_visual else None)
    assert abs(float(loss) - float(want_loss)) < 1e-3 * abs(float(want_loss))
    eng.backward(1.0)
    assert set(eng._views) == set(tr)                     # gradient storage only for the trainable parameters
    for name, p in tr.items():
        assert p.grad is not None, name
        assert relerr(p.grad, want[name]) < 2e-3, name
    for n, p in m.named_parameters():
        if n not in tr:
            assert p.grad is None, n
    # accumulation over a second micro-step
    eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV) if with_visual else None)
    eng.backward(0.5)
    n0 = "layers.0.attention.wq.lora_b.weight"
    assert relerr(tr[n0].grad, 1.5 * want[n0]) < 2e-3


def test_lora_bf16_step_and_optimizer(golden_dir):
    """autocast-style step on the bf16 path: cosine agreement of every adapter gradient with the fp32 oracle, then AdamW
    updates only the trainable parameters."""
    m, oargs, sd, lsd, vsd = build(True, BF)
    promote_trainable_params_to_fp32(m)
    g = torch.Generator().manual_seed(6)
    B, T = 2, 16
    ex = torch.randint(3, 192, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :5] = 0
    img = synth_image(B, size=112, seed=4)
    tr = m.get_trainable_params()
    osd = {k: v.clone().requires_grad_(k in tr) for k, v in {**sd, **lsd}.items()}
    ovs = {k: v.clone().requires_grad_(k in tr) for k, v in vsd.items()}
    dec = ref_cpu.OracleDecoder(oargs, osd)
    views = ref_cpu.encode_image(img, ovs, vit_layers=2, vit_heads=4, n_views=1)
    itok = ref_cpu.assemble_image_tokens(views, ovs["start_img"], ovs["end_img"])
    want_loss = ref_cpu.meta_forward_loss(dec, ex, lab, itok)
    want_loss.backward()
    eng = TrainEngine(m, BF)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), img.to(DEV))
    assert abs(float(loss) - float(want_loss)) < 2e-2 * abs(float(want_loss))
    eng.backward(1.0)
    for name, p in tr.items():
        a, b = p.grad.float().cpu().flatten(), {**osd, **ovs}[name].grad.flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-20))
        assert cos > 0.98, (name, cos)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, betas=(0.9, 0.95))
    opt.step()
    for n, p in m.named_parameters():
        changed = not torch.equal(before[n], p.detach())
        assert changed == (n in tr), n


def test_lora_bf16_tn_adapter_gradients_match_transposed_path():
    """dim 256, 150 tokens: every adapter weight gradient goes through a3v_gemm_tn_splitk on the token-major operands; same
    values as the transpose + NT split-K path up to fp32 summation order before the single bf16 rounding of the product."""
    big = dict(dim=256, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=320, multiple_of=256, max_seq_len=256)
    oargs = ref_cpu.OracleArgs(**big)
    sd = ref_cpu.make_decoder_weights(oargs, seed=3, std=0.05)
    lsd = ref_cpu.make_lora_weights(oargs, RANK, seed=6, std_a=0.05, std_b=0.05)
    g = torch.Generator().manual_seed(9)
    B, T = 3, 50
    ex = torch.randint(3, 320, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :7] = 0
    got = {}
    for tn in (True, False):
        m = peft.Transformer(peft.ModelArgs(**big, lora_rank=RANK), with_visual=False)
        m.load_state_dict({**sd, **lsd}, strict=True)
        train = m.get_trainable_params()
        for n, p in m.named_parameters():
            p.requires_grad = n in train
        m.to(BF).to(DEV)
        promote_trainable_params_to_fp32(m)
        eng = TrainEngine(m, BF)
        eng.tn_wgrad = tn
        for scale in (1.0, 0.5):
            eng.forward_loss(ex.to(DEV), lab.to(DEV), None)
            eng.backward(scale)
        got[tn] = {n: p.grad.float().cpu() for n, p in train.items()}
    assert any("lora_a" in n for n in got[True]) and any("lora_b" in n for n in got[True])
    for n in got[True]:
        assert relerr(got[True][n], got[False][n]) < 1e-2, n
        assert float(got[True][n].abs().max()) > 0


def test_lora_step_images_refresh_in_place_after_optimizer_step():
    """After an optimizer step the engine re-writes only the adapter rows / columns of its padded, block-diagonal images:
    the result must equal images built from scratch, for an optimizer that bumps Tensor._version and for one that does not."""
    from a3vlm_amd.optim import FusedAdamW
    m, oargs, sd, lsd, vsd = build(False, BF)
    promote_trainable_params_to_fp32(m)
    g = torch.Generator().manual_seed(11)
    ex = torch.randint(3, 192, (2, 20), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :4] = 0
    eng = TrainEngine(m, BF)
    params = [p for p in m.parameters() if p.requires_grad]
    # FusedAdamW(engine=eng): the multi-tensor launch writes every adapter's bf16 values into the group images itself (strided sinks)
    # and marks them current; without the engine (and for torch's optimizer) the engine re-writes them with a3v_lora_refresh
    for opt in (FusedAdamW(params, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0),
                FusedAdamW(params, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, engine=eng),
                torch.optim.AdamW(params, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, fused=True)):
        for _ in range(2):
            eng.forward_loss(ex.to(DEV), lab.to(DEV), None)
            eng.backward(1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            if getattr(opt, "engine", None) is eng:
                ver = eng._li_ver
                li = eng._lora_step_images()
                assert eng._li_ver == ver             # nothing was refreshed: the optimizer's own writes are what is compared below
            li = eng._lora_step_images()                  # refreshed in place
            fresh = TrainEngine(m, BF)._lora_step_images()  # built from scratch from the updated parameters
            assert set(li) == set(fresh)
            for k in li:
                assert torch.equal(li[k], fresh[k]), k


def test_lora_prefill_fused_qkv_matches_separate_kernels():
    """peft plugin, head_dim 64, prefill of 61 tokens + two decode steps: with the adapter term folded into the fused qkv / RoPE /
    cache GEMM the logits and caches are bit-identical to the separate kernels (same rounding points), and close to the oracle."""
    big = dict(dim=256, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=320, multiple_of=256, max_seq_len=256)
    oargs = ref_cpu.OracleArgs(**big)
    sd = ref_cpu.make_decoder_weights(oargs, seed=3, std=0.05)
    lsd = ref_cpu.make_lora_weights(oargs, RANK, seed=6, std_a=0.05, std_b=0.05)
    m = peft.Transformer(peft.ModelArgs(**big, lora_rank=RANK), with_visual=False)
    m.load_state_dict({**sd, **lsd}, strict=True)
    m.to(BF).to(DEV)
    g = torch.Generator().manual_seed(12)
    B, T0 = 3, 61
    ex = torch.randint(3, 320, (B, T0 + 2), generator=g)
    ex[:, 0] = 1
    exd = ex.to(DEV)
    got = {}
    for fuse in (True, False):
        m._fuse_qkv_rope = fuse
        lg = [m.forward_inference(exd[:, :T0], 0).float().clone()]
        for t in range(T0, T0 + 2):
            lg.append(m.forward_inference(exd[:, t:t + 1], t).float().clone())
        got[fuse] = (lg, [k.clone() for k in m._k_cache], [v.clone() for v in m._vt_cache])
    for a, b in zip(got[True][0], got[False][0]):
        assert torch.equal(a, b)
    for l in range(2):
        assert torch.equal(got[True][1][l][:, :, :T0 + 2], got[False][1][l][:, :, :T0 + 2])
        assert torch.equal(got[True][2][l][:, :, :, :T0 + 2], got[False][2][l][:, :, :, :T0 + 2])
    dec = ref_cpu.OracleDecoder(oargs, {k: v.to(BF) for k, v in {**sd, **lsd}.items()})
    want = dec.forward_inference(ex[:, :T0], 0).float()
    assert float((got[True][0][0].cpu() - want).abs().max()) / float(want.abs().max()) < 4e-2


def test_lora_kext_adapters_inside_the_gemms_match_oracle_and_separate_path():
    """head_dim 64: the adapters ride inside the four decoder GEMMs ([x | t] . [W | B]^T, K extended by the padded rank).
    Loss and every trainable gradient agree with the oracle's autograd (bf16 level) and with the separate-pass path; the
    extended weight images carry B in their tail columns after an optimizer step (refreshed in place)."""
    from a3vlm_amd.optim import FusedAdamW
    big = dict(dim=256, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=320, multiple_of=256, max_seq_len=256)
    oargs = ref_cpu.OracleArgs(**big)
    sd = ref_cpu.make_decoder_weights(oargs, seed=3, std=0.05)
    lsd = ref_cpu.make_lora_weights(oargs, RANK, seed=6, std_a=0.05, std_b=0.05)
    g = torch.Generator().manual_seed(13)
    B, T = 3, 47
    ex = torch.randint(3, 320, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :6] = 0
    got, losses = {}, {}
    for kext in (True, False):
        m = peft.Transformer(peft.ModelArgs(**big, lora_rank=RANK), with_visual=False)
        m.load_state_dict({**sd, **lsd}, strict=True)
        train = m.get_trainable_params()
        for n, p in m.named_parameters():
            p.requires_grad = n in train
        m.to(BF).to(DEV)
        promote_trainable_params_to_fp32(m)
        eng = TrainEngine(m, BF)
        eng.lora_kext = kext
        assert (eng._kext() > 0) == kext
        losses[kext] = float(eng.forward_loss(ex.to(DEV), lab.to(DEV), None))
        eng.backward(1.0)
        got[kext] = {n: p.grad.float().cpu().clone() for n, p in train.items()}
        if kext:      # one optimizer step, then the B blocks inside the extended images equal the updated adapters
            opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0)
            opt.step()
            eng.forward_loss(ex.to(DEV), lab.to(DEV), None)
            full = eng._images()["wo.0.x"]
            wb = m.layers[0].attention.wo.lora_b.weight
            assert torch.equal(full[:, 256:256 + RANK], wb.detach().to(BF)) and float(full[:, 256 + RANK:].float().abs().sum()) == 0
    tr = {n for n in got[True]}
    osd = {k: v.clone().requires_grad_(k in tr) for k, v in {**sd, **lsd}.items()}
    want_loss = ref_cpu.meta_forward_loss(ref_cpu.OracleDecoder(oargs, osd), ex, lab, None)
    want_loss.backward()
    assert abs(losses[True] - float(want_loss)) < 2e-2 * abs(float(want_loss)) and abs(losses[True] - losses[False]) < 1e-2 * abs(losses[False])
    for n in got[True]:
        a, b, c = got[True][n].flatten(), got[False][n].flatten(), osd[n].grad.flatten()
        cos = lambda x, y: float(torch.dot(x, y) / (x.norm() * y.norm() + 1e-20))
        assert cos(a, b) > 0.995 and cos(a, c) > 0.98, (n, cos(a, b), cos(a, c))


def test_lora_swiglu_backward_in_the_w2_dgrad_epilogue_equals_the_separate_pass():
    """The LoRA step (adapters inside the GEMMs) runs the SwiGLU backward in the epilogue of w2's input-gradient GEMM
    (A3V_EPI_SWIGLU_BWD: d(act) never reaches HBM).  Same arithmetic on the same bf16-rounded product, so the loss and every
    trainable gradient are IDENTICAL to the path with the separate a3v_swiglu_bwd pass (fuse_swiglu_bwd = False)."""
    big = dict(dim=256, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=320, multiple_of=256, max_seq_len=256)
    oargs = ref_cpu.OracleArgs(**big)
    sd = ref_cpu.make_decoder_weights(oargs, seed=3, std=0.05)
    lsd = ref_cpu.make_lora_weights(oargs, RANK, seed=6, std_a=0.05, std_b=0.05)
    g = torch.Generator().manual_seed(17)
    B, T = 5, 61                                   # 305 rows: ragged against the 256-row tiles
    ex = torch.randint(3, 320, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :6] = 0
    got, losses = {}, {}
    for fuse in (True, False):
        m = peft.Transformer(peft.ModelArgs(**big, lora_rank=RANK), with_visual=False)
        m.load_state_dict({**sd, **lsd}, strict=True)
        train = m.get_trainable_params()
        for n, p in m.named_parameters():
            p.requires_grad = n in train
        m.to(BF).to(DEV)
        promote_trainable_params_to_fp32(m)
        eng = TrainEngine(m, BF)
        eng.fuse_swiglu_bwd = fuse
        assert eng._kext() > 0
        losses[fuse] = float(eng.forward_loss(ex.to(DEV), lab.to(DEV), None))
        eng.backward(1.0)
        got[fuse] = {n: p.grad.float().cpu().clone() for n, p in train.items()}
    assert losses[True] == losses[False]
    exact = 0
    for n in got[True]:
        a, b = got[True][n], got[False][n]
        exact += int(torch.equal(a, b))
        # (the norm-weight gradients are summed with atomics: equal up to the order of fp32 additions, run to run, on either path)
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12, (n, float((a - b).abs().max()), float(b.abs().max()))
    assert exact >= len(got[True]) - 2 * big["n_layers"] - 1, exact      # every adapter gradient bit for bit
