"""-m gpu: ONE decoder block of the plugin at the FULL width the bench times (7B: dim 4096 / 32 heads / ffn 11008; 13B: 5120 / 40 / 13824;
B = 8, S = 1091 -> 8728 token rows), composed as the plugin dispatches it -- the ring GEMM's fused qkv + RoPE + cache epilogue, the hybrid
big-tile + split-K-tail dispatch of wo / w2, the SwiGLU epilogue, 32- / 40-head MFMA attention, residual epilogues chained, the one-call
decode step, and the training forward + backward of TrainEngine (full fine-tune and LoRA) -- against ``oracle/ref_cpu.py``'s block
(LLM/llama_ens5.py:220-249) on the same bf16-rounded weights.  Model-level fixtures stop at dim 512 and the 7B / 13B shapes were only
covered kernel by kernel; this file checks the composition at the size the driver times.  Same for one CLIP ViT block at width 1024.

The oracle side is one layer at ~3.5 TFLOP per forward (a few seconds on the host's cores); it runs in fp32 on the bf16-rounded
weights, the bounds: 3e-2 of max |logit|, 1e-3 on the loss (north_star), relative L2 <= 2e-2 per gradient tensor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from a3vlm_amd.model.LLM import llama_ens5 as plugin  # noqa: E402
from a3vlm_amd.model.LLM import llama_ens5_peft as peft  # noqa: E402
from a3vlm_amd.train import TrainEngine  # noqa: E402
from a3vlm_amd.util import promote_trainable_params_to_fp32  # noqa: E402
from oracle import ref_cpu  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
B, S, V = 8, 1091, 512
GEOM = {"7b": dict(dim=4096, n_heads=32, multiple_of=256), "13b": dict(dim=5120, n_heads=40, multiple_of=256)}


def _args(g):
    return dict(n_layers=1, vocab_size=V, max_seq_len=2048, **GEOM[g])


def _weights(g, seed):
    oargs = ref_cpu.OracleArgs(**_args(g))
    sd = ref_cpu.make_decoder_weights(oargs, seed=seed, std=0.02)
    return oargs, {k: v.to(BF).float() for k, v in sd.items()}              # the values the bf16 build multiplies with


def _tokens(seed, b=B, s=S):
    gen = torch.Generator().manual_seed(seed)
    ex = torch.randint(3, V, (b, s), generator=gen)
    ex[:, 0] = 1
    return ex


@pytest.mark.parametrize("g", ["7b", "13b"])
def test_block_forward_prefill_and_decode_at_full_width(g):
    oargs, sd = _weights(g, 11)
    m = plugin.Transformer(plugin.ModelArgs(**_args(g)))
    m.load_state_dict(sd)
    m.to(BF).to(DEV)
    assert m.ffn == {"7b": 11008, "13b": 13824}[g] and m.head_dim == 128
    ex = _tokens(1)
    dec = ref_cpu.OracleDecoder(oargs, sd)
    rows = [0, 3, 7]                                                          # batch rows the oracle runs (rows are independent)
    with torch.no_grad():
        want = dec.forward(ex[rows]).float()                                 # teacher-forced logits of every position
    got = m(ex.to(DEV)).float().cpu()[rows]
    scale = float(want.abs().max())
    err = float((got - want).abs().max()) / scale
    assert err < 3e-2, err
    noise = float((got - want).abs().max())
    top2 = torch.sort(want, dim=-1).values[..., -2:]
    decided = (top2[..., 1] - top2[..., 0]) > 2 * noise
    assert bool((got.argmax(-1) == want.argmax(-1))[decided].all()) and int(decided.sum()) > 0.3 * decided.numel()
    # cached inference: prefill of S - 2 tokens, then two decode steps through a3v_llama_decode_step (all 8 rows on the device)
    P = S - 2
    with torch.no_grad():
        w0 = dec.forward_inference(ex[rows, :P], 0).float()
    l0 = m.forward_inference(ex[:, :P].to(DEV), 0).float().cpu()[rows]
    assert float((l0 - w0).abs().max()) / scale < 3e-2
    for t in (P, P + 1):
        with torch.no_grad():
            wt = dec.forward_inference(ex[rows, t:t + 1], t).float()
        lt = m.forward_inference(ex[:, t:t + 1].to(DEV), t).float().cpu()[rows]
        assert float((lt - wt).abs().max()) / scale < 3e-2, t


LOSS_REL = 1e-3    # north_star: loss within 1e-3 relative of the reference's eager path
REL_L2 = 2e-2      # per-tensor bound on |g - g_oracle|_2 / |g_oracle|_2 (round 5; was cosine > 0.99, i.e. a relative L2 of 0.14: a ragged
                   # 24-row tile dropped from a weight gradient -- 0.3 % of the tokens, relative L2 ~ 5e-2 -- passed)


def _grad_check(tr, want, bound=REL_L2, what=""):
    worst = {}
    for name, p in tr.items():
        assert p.grad is not None, name
        a, b = p.grad.float().cpu().flatten(), want[name].float().flatten()
        worst[name] = float((a - b).norm() / (b.norm() + 1e-20))
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"{what} gradient relative L2 vs oracle, worst: {[(n, round(v, 4)) for n, v in top]}")
    bad = {n: v for n, v in worst.items() if not v < bound}
    assert not bad, (what, bad)


@pytest.mark.parametrize("g", ["7b", "13b"])
def test_block_full_fine_tune_step_at_full_width(g):
    """TrainEngine forward + backward (TN weight gradients, NN input gradients with the SwiGLU-backward epilogue, packed attention
    backward, bf16 residual stream) at 8728 rows against oracle autograd over the same block."""
    oargs, sd = _weights(g, 12)
    m = plugin.Transformer(plugin.ModelArgs(**_args(g)))
    m.load_state_dict(sd)
    for p in m.parameters():
        p.requires_grad = True
    m.to(BF).to(DEV)
    promote_trainable_params_to_fp32(m)
    ex = _tokens(2)
    lab = ex.clone()
    lab[:, :600] = 0
    lab[5, 900:] = 0
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want_loss = ref_cpu.meta_forward_loss(ref_cpu.OracleDecoder(oargs, osd), ex, lab, None)
    want_loss.backward()
    want = {k: v.grad for k, v in osd.items()}
    eng = TrainEngine(m, BF)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), None)
    print("full fine-tune", g, "loss", float(loss), "oracle", float(want_loss))
    assert abs(float(loss) - float(want_loss)) < LOSS_REL * abs(float(want_loss))
    eng.backward(1.0)
    _grad_check(m.get_trainable_params(), want, what=f"full fine-tune {g}")


@pytest.mark.parametrize("g,rank", [("7b", 16), ("13b", 32)])
def test_block_lora_step_at_full_width(g, rank):
    """The headline step's dispatch (adapters inside the GEMMs, NT input gradients over the transposed frozen images, strip weight
    gradients, multi-tensor sinks) at 8728 rows.  Rank 32 also covers the group widths pad64(r) != pad64(3 r)."""
    oargs, sd = _weights(g, 13)
    lsd = {k: v.to(BF).float() for k, v in ref_cpu.make_lora_weights(oargs, rank, seed=5, std_a=0.02, std_b=0.02).items()}
    m = peft.Transformer(peft.ModelArgs(**_args(g), lora_rank=rank))
    m.load_state_dict({**sd, **lsd}, strict=True)
    tr = m.get_trainable_params()
    for n, p in m.named_parameters():
        p.requires_grad = n in tr
    m.to(BF).to(DEV)
    promote_trainable_params_to_fp32(m)
    tr = m.get_trainable_params()
    ex = _tokens(3)
    lab = ex.clone()
    lab[:, :600] = 0
    osd = {k: v.clone().requires_grad_(k in tr) for k, v in {**sd, **lsd}.items()}
    want_loss = ref_cpu.meta_forward_loss(ref_cpu.OracleDecoder(oargs, osd), ex, lab, None)
    want_loss.backward()
    want = {k: v.grad for k, v in osd.items() if v.requires_grad}
    eng = TrainEngine(m, BF)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), None)
    print("lora", g, "loss", float(loss), "oracle", float(want_loss))
    assert abs(float(loss) - float(want_loss)) < LOSS_REL * abs(float(want_loss))
    eng.backward(1.0)
    _grad_check({k: v for k, v in tr.items() if v.requires_grad}, want, what=f"LoRA r{rank} {g}")


def test_block_lora_step_at_the_reference_recipe_length():
    """The reference trains at max_words 2048 (scripts/a3vlm_train.sh:45-55: 448^2 input -> 1455 image words + text): one 7B-width LoRA
    block step at S = 2048 (B = 4: 8192 rows; attention backward at 16 key blocks per query block instead of 9) against oracle autograd."""
    g, rank, b, s_len = "7b", 16, 4, 2048
    oargs, sd = _weights(g, 14)
    lsd = {k: v.to(BF).float() for k, v in ref_cpu.make_lora_weights(oargs, rank, seed=6, std_a=0.02, std_b=0.02).items()}
    m = peft.Transformer(peft.ModelArgs(**_args(g), lora_rank=rank))
    m.load_state_dict({**sd, **lsd}, strict=True)
    tr = m.get_trainable_params()
    for n, p in m.named_parameters():
        p.requires_grad = n in tr
    m.to(BF).to(DEV)
    promote_trainable_params_to_fp32(m)
    tr = m.get_trainable_params()
    ex = _tokens(4, b, s_len)
    lab = ex.clone()
    lab[:, :1455] = 0
    lab[2, 1900:] = 0
    osd = {k: v.clone().requires_grad_(k in tr) for k, v in {**sd, **lsd}.items()}
    want_loss = ref_cpu.meta_forward_loss(ref_cpu.OracleDecoder(oargs, osd), ex, lab, None)
    want_loss.backward()
    want = {k: v.grad for k, v in osd.items() if v.requires_grad}
    eng = TrainEngine(m, BF)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), None)
    print("lora S=2048 loss", float(loss), "oracle", float(want_loss))
    assert abs(float(loss) - float(want_loss)) < LOSS_REL * abs(float(want_loss))
    eng.backward(1.0)
    _grad_check({k: v for k, v in tr.items() if v.requires_grad}, want, what="LoRA r16 7b S=2048")


def test_vit_block_at_width_1024():
    """One CLIP ViT-L/14@336 resblock (width 1024, 16 heads, 577 tokens x 8 images = 4616 rows: the K / N = 1024 GEMM shapes, the
    bias + GELU epilogues and the hd-64 non-causal attention) through the plugin's clip_encode_image against the oracle's."""
    width, heads, layers, grid = 1024, 16, 1, 24
    vsd = ref_cpu.make_vision_weights(256, width=width, layers=layers, patch=14, grid=grid, seed=4, std=0.02)
    vsd = {k: v.to(BF).float() for k, v in vsd.items()}
    args = plugin.ModelArgs(dim=256, n_layers=1, n_heads=2, vocab_size=64, multiple_of=64, max_seq_len=1024, vit_width=width,
                            vit_layers=layers, vit_heads=heads, vit_crop=336, n_views=1)
    m = plugin.Transformer(args, with_visual=True)
    dsd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(dim=256, n_layers=1, n_heads=2, vocab_size=64, multiple_of=64, max_seq_len=1024),
                                       seed=1, std=0.05)
    m.load_state_dict({**dsd, **vsd})
    m.to(BF).to(DEV)
    gen = torch.Generator().manual_seed(7)
    img = torch.randn(B, 3, 336, 336, generator=gen).to(BF).float()
    with torch.no_grad():
        want = ref_cpu.clip_encode_image(img[:2], vsd, layers, heads).float()
    got = m.clip_encode_image(img.to(DEV).to(BF)).float().cpu().view(B, -1, width)[:2]
    assert got.shape == want.shape
    assert float((got - want).abs().max()) / float(want.abs().max()) < 3e-2
