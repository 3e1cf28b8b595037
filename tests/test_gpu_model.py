"""-m gpu: the HIP plugin / MetaModel end to end against (a) the golden vectors captured from the
reference (tests/golden, fp32 path, token ids bit-exact) and (b) the CPU oracle on seeded inputs
(bf16 path with stated tolerances), plus size-independent properties at larger geometry."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from a3vlm_amd.model.LLM import llama_ens5 as plugin  # noqa: E402
from a3vlm_amd.model.meta import MetaModel  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from oracle.gen_golden import TINY, VIT, convnext_tokens, extra_feature_inputs, synth_image  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
# north_star tolerance: 1e-3 relative on outputs of the fp32 path (logits scaled by their max)
REL = 1e-3


def rel_err(got, want):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want, dtype=np.float32)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12))


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "meta_tiny.json")) as f:
        j = json.load(f)
    return dict(dec=np.load(os.path.join(golden_dir, "decoder_tiny.npz")),
                meta=np.load(os.path.join(golden_dir, "meta_tiny.npz")),
                vis=np.load(os.path.join(golden_dir, "vision_tiny.npz")), j=j, dir=golden_dir)


def tiny_model(V, dtype=torch.float32, max_seq_len=64, with_visual=False, **vis):
    args = plugin.ModelArgs(vocab_size=V, **{**TINY, "max_seq_len": max_seq_len}, **vis)
    m = plugin.Transformer(args, with_visual=with_visual)
    oargs = ref_cpu.OracleArgs(vocab_size=V, **{**TINY, "max_seq_len": max_seq_len})
    sd = ref_cpu.make_decoder_weights(oargs, seed=0, std=0.08)
    return m, oargs, sd


def test_g4_forward_logits_fp32(gold):
    V = gold["j"]["vocab_size"]
    m, _, sd = tiny_model(V)
    m.load_state_dict(sd)
    m.to(DEV)
    ex = torch.from_numpy(gold["dec"]["g4_examples"]).to(DEV)
    out = m(ex)
    assert out.shape == (2, 12, V) and out.dtype == torch.float32
    assert rel_err(out, gold["dec"]["g4_logits"]) < REL
    assert (out.argmax(-1).cpu().numpy() == gold["dec"]["g4_logits"].argmax(-1)).all()


def test_g4_forward_inference_fp32(gold):
    V = gold["j"]["vocab_size"]
    m, _, sd = tiny_model(V)
    m.load_state_dict(sd)
    m.to(DEV)
    ex = torch.from_numpy(gold["dec"]["g4_examples"]).to(DEV)
    want = gold["dec"]["g4_inf_logits"]
    lg = [m.forward_inference(ex[:, :7], 0).clone()]
    for t in range(7, 11):
        lg.append(m.forward_inference(ex[:, t:t + 1], t).clone())
    for i, l in enumerate(lg):
        assert l.dtype == torch.float32
        assert rel_err(l, want[i]) < REL, f"step {i}"
        assert (l.argmax(-1).cpu().numpy() == want[i].argmax(-1)).all()
    # KV-cache contents (reference layout [B,S,Hkv,hd]) vs ours (K [B,Hkv,S,hd], V^T [B,Hkv,hd,S])
    k0 = m._k_cache[0][:, :, :11].permute(0, 2, 1, 3)
    v1 = m._vt_cache[1][:, :, :, :11].permute(0, 3, 1, 2)
    assert rel_err(k0, gold["dec"]["g4_kcache_l0"]) < REL
    assert rel_err(v1, gold["dec"]["g4_vcache_l1"]) < REL


def test_state_dict_names_match_reference(gold):
    V = gold["j"]["vocab_size"]
    m, _, _ = tiny_model(V)
    assert sorted(m.state_dict().keys()) == list(gold["dec"]["dec_state_keys"])
    assert sorted(m.get_trainable_params().keys()) == list(gold["dec"]["dec_trainable"])
    args = plugin.ModelArgs(vocab_size=V, **TINY, vit_width=VIT["width"], vit_layers=VIT["layers"], vit_heads=VIT["heads"],
                            extra_feat_dim=3072 + 1536, qformer_tokens=32)
    mv = plugin.Transformer(args, with_visual=True)
    vis_keys = sorted(k for k in mv.state_dict().keys() if k.startswith(("clip.", "visual_proj", "qformer_proj", "start_", "end_")))
    assert vis_keys == list(gold["vis"]["vis_state_keys"])
    assert not any(n.startswith("clip.") for n in mv.get_trainable_params())
    assert mv.image_words == 1455 and mv.image_size == 448


def meta_model(gold, dtype=torch.float32):
    mm = MetaModel("llama_ens5", os.path.join(gold["dir"], "tiny_params.json"),
                   os.path.join(gold["dir"], "tokenizer.model"), with_visual=False, max_seq_len=64)
    oargs = ref_cpu.OracleArgs(vocab_size=gold["j"]["vocab_size"], **TINY)
    sd = ref_cpu.make_decoder_weights(oargs, seed=0, std=0.08)
    mm.llma.load_state_dict(sd)
    return mm.to(dtype).to(DEV)


def test_g5_meta_forward_loss_fp32(gold):
    mm = meta_model(gold)
    mm.train_compute_dtype = torch.float32      # grad mode on: this goes through the training engine's forward
    g = gold["meta"]
    ex = torch.from_numpy(g["g5_examples"]).to(DEV)
    with torch.no_grad():                        # inference-path loss
        l0, _ = mm(ex, torch.from_numpy(g["g5_labels_a"]).to(DEV))
    assert abs(float(l0) - float(g["g5_loss_a"])) < REL * float(g["g5_loss_a"])
    la, _ = mm(ex, torch.from_numpy(g["g5_labels_a"]).to(DEV))
    assert abs(float(la) - float(g["g5_loss_a"])) < REL * float(g["g5_loss_a"])
    lb, _ = mm(torch.from_numpy(g["g5_examples_b"]).to(DEV), torch.from_numpy(g["g5_labels_b"]).to(DEV))
    assert abs(float(lb) - float(g["g5_loss_b"])) < REL * float(g["g5_loss_b"])
    lc, _ = mm(ex, torch.zeros_like(ex))
    assert float(lc) == 0.0


@pytest.mark.parametrize("key,max_gen,stops", [("gen12", 12, ()), ("gen48", 48, ()), ("genstop", 12, ("li", "ab"))])
def test_g6_generate_greedy_bit_exact_fp32(gold, key, max_gen, stops):
    mm = meta_model(gold)
    texts, ids = mm.generate(gold["j"]["prompts"], None, max_gen_len=max_gen, temperature=0.0,
                             additional_stop_symbols=stops, return_ids=True)
    assert ids == gold["j"][key + "_ids"]
    assert texts == gold["j"][key + "_text"]
    # the stop flag is polled every poll_every steps (default 4): identical outputs at the reference's cadence (1) and beyond
    for pe in (1, 3, 64):
        t2, i2 = mm.generate(gold["j"]["prompts"], None, max_gen_len=max_gen, temperature=0.0, additional_stop_symbols=stops,
                             return_ids=True, poll_every=pe)
        assert i2 == ids and t2 == texts, pe


def vision_model(gold, dtype=torch.float32):
    V = gold["j"]["vocab_size"]
    args = plugin.ModelArgs(vocab_size=V, **{**TINY, "max_seq_len": 1600}, vit_width=VIT["width"], vit_layers=VIT["layers"],
                            vit_heads=VIT["heads"], vit_patch=VIT["patch"], vit_crop=224, n_views=5,
                            extra_feat_dim=3072 + 1536, qformer_tokens=32)
    m = plugin.Transformer(args, with_visual=True)
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=VIT["width"], layers=VIT["layers"], patch=VIT["patch"], grid=VIT["grid"],
                                      in_feat=VIT["width"] + 3072 + 1536, with_qformer=True, seed=1, std=0.05)
    res = m.load_state_dict({**sd, **vsd}, strict=True)
    return m.to(dtype).to(DEV), sd, vsd


def test_g7_g8_vision_fp32(gold):
    v = gold["vis"]
    m, sd, vsd = vision_model(gold)
    B = 2
    img = synth_image(B).to(DEV)
    views = torch.empty(5 * B, 3, 224, 224, device=DEV)
    from a3vlm_amd import ops
    ops.split_views(img, views)
    feats = m.clip_encode_image(views[:3].contiguous()).view(3, 257, -1)
    assert rel_err(feats, v["g7_clip_feats"]) < REL
    qf, cnx, dino = extra_feature_inputs(5 * B)
    extra = [convnext_tokens(cnx).to(DEV), dino.to(DEV)]
    ex = torch.from_numpy(v["g8_examples"]).to(DEV)
    out = m(ex, img, qformer_feats=qf.to(DEV), extra_feats=extra)
    assert rel_err(out, v["g8_logits"]) < REL
    # the assembled image words inside the sequence buffer vs the reference's per-view tokens
    S = ex.shape[1] + m.image_words
    l0 = m.forward_inference(ex[:, :6], 0, img, qformer_feats=qf.to(DEV), extra_feats=extra).clone()
    assert m.cache_image_words == int(v["g8_cache_image_words"]) == 1455
    l1 = m.forward_inference(ex[:, 6:7], 6).clone()
    l2 = m.forward_inference(ex[:, 7:8], 7).clone()
    for i, l in enumerate((l0, l1, l2)):
        assert rel_err(l, v["g8_inf_logits"][i]) < REL, f"inference step {i}"
        assert (l.argmax(-1).cpu().numpy() == v["g8_inf_logits"][i].argmax(-1)).all()


def test_image_words_layout_fp32(gold):
    """encode_image_into writes [start | qformer | clip | end] x 5 views at rows 1..1455 (llama_ens5.py:471-479)."""
    v = gold["vis"]
    m, sd, vsd = vision_model(gold)
    B, T = 2, 4
    img = synth_image(B).to(DEV)
    qf, cnx, dino = extra_feature_inputs(5 * B)
    W = m.image_words
    S = T + W
    h = torch.zeros(B * S, 64, device=DEV)
    m._pack(check=True)
    m.encode_image_into(h, img, B, S, qf.to(DEV), [convnext_tokens(cnx).to(DEV), dino.to(DEV)])
    hv = h.view(B, S, 64)[:, 1:1 + W].cpu()
    views = torch.from_numpy(v["g8_views"])                       # [5, B, 289, 64]
    want = ref_cpu.assemble_image_tokens(list(views), vsd["start_img"], vsd["end_img"])
    assert rel_err(hv, want.numpy()) < REL
    assert float(h.view(B, S, 64)[:, 0].abs().sum()) == 0 and float(h.view(B, S, 64)[:, 1 + W:].abs().sum()) == 0


# ------------------------------------------------------------------ bf16 path vs the oracle
def test_decoder_bf16_vs_oracle(gold):
    """bf16 kernels (fp32 accumulate, reference rounding points) vs the oracle run in bf16 on CPU.
    Accumulation order differs, so agreement is at the bf16-ulp level: 3e-2 of max|logit|
    (the reference's own bf16-vs-fp32 deviation on this model, fixture g4_logits_bf16, is ~2e-2)."""
    V = gold["j"]["vocab_size"]
    m, oargs, sd = tiny_model(V)
    m.load_state_dict(sd)
    m.to(BF).to(DEV)
    ex = torch.from_numpy(gold["dec"]["g4_examples"])
    d = ref_cpu.OracleDecoder(oargs, {k: v.to(BF) for k, v in sd.items()})
    want = d.forward(ex).float()
    out = m(ex.to(DEV))
    assert out.dtype == BF
    assert rel_err(out, want.numpy()) < 3e-2
    assert rel_err(out, gold["dec"]["g4_logits_bf16"]) < 3e-2      # the reference's own bf16 run
    lg = m.forward_inference(ex[:, :7].to(DEV), 0)
    assert rel_err(lg, d.forward_inference(ex[:, :7], 0).numpy()) < 3e-2
    lg = m.forward_inference(ex[:, 7:8].to(DEV), 7)
    assert rel_err(lg, d.forward_inference(ex[:, 7:8], 7).numpy()) < 3e-2


def test_loss_bf16_within_1e2(gold):
    mm = meta_model(gold, BF)
    g = gold["meta"]
    with torch.no_grad():      # evaluation of the loss with bf16 weights (training wants fp32 masters: tests/test_gpu_train.py)
        la, _ = mm(torch.from_numpy(g["g5_examples"]).to(DEV), torch.from_numpy(g["g5_labels_a"]).to(DEV))
    assert abs(float(la) - float(g["g5_loss_a"])) < 1e-2 * float(g["g5_loss_a"])


# ------------------------------------------------------------------ properties at larger geometry
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (BF, 4e-2)])
def test_prefill_decode_consistency_medium(dtype, tol):
    """forward (teacher forced) == prefill + decode steps, hd=128, GQA, 3 layers, ragged sizes."""
    args = plugin.ModelArgs(dim=512, n_layers=3, n_heads=4, n_kv_heads=2, vocab_size=1000, multiple_of=64, max_seq_len=256)
    torch.manual_seed(0)
    m = plugin.Transformer(args).to(dtype).to(DEV)
    B, T = 3, 150
    ex = torch.randint(3, 1000, (B, T), device=DEV)
    full = m(ex).float()
    lg = m.forward_inference(ex[:, :147], 0).float().clone()
    scale = full.abs().max()
    assert ((lg - full[:, 146]).abs().max() / scale) < tol
    for t in range(147, 150):
        lg = m.forward_inference(ex[:, t:t + 1], t).float()
        assert ((lg - full[:, t]).abs().max() / scale) < tol, t


def test_batch_row_independence_bf16():
    """A row's logits do not depend on what else is in the batch (sharding contract of DP replicas)."""
    args = plugin.ModelArgs(dim=256, n_layers=2, n_heads=2, vocab_size=512, multiple_of=64, max_seq_len=128)
    torch.manual_seed(1)
    m = plugin.Transformer(args).to(BF).to(DEV)
    ex = torch.randint(3, 512, (4, 40), device=DEV)
    a = m(ex).clone()
    b = m(ex[1:3]).clone()
    assert torch.equal(a[1:3], b)


def test_vit_single_crop_geometry_bf16():
    """Geometry S (BASELINE-literal): one 336x336 crop, ViT patch 14 -> 577 + 2 image words; vs oracle."""
    args = plugin.ModelArgs(dim=128, n_layers=1, n_heads=2, vocab_size=256, multiple_of=64, max_seq_len=1024,
                            vit_width=128, vit_layers=2, vit_heads=2, vit_crop=336, n_views=1)
    m = plugin.Transformer(args, with_visual=True)
    assert m.image_words == 579 and m.image_size == 336
    oargs = ref_cpu.OracleArgs(dim=128, n_layers=1, n_heads=2, vocab_size=256, multiple_of=64, max_seq_len=1024)
    sd = ref_cpu.make_decoder_weights(oargs, seed=3, std=0.05)
    vsd = ref_cpu.make_vision_weights(128, width=128, layers=2, patch=14, grid=24, seed=4, std=0.05)
    m.load_state_dict({**sd, **vsd})
    B = 2
    img = synth_image(B, size=336, seed=9)
    ex = torch.randint(3, 256, (B, 10), generator=torch.Generator().manual_seed(5))
    ex[:, 0] = 1
    # fp32 path vs oracle
    m.to(DEV)
    views = ref_cpu.encode_image(img, vsd, vit_layers=2, vit_heads=2, n_views=1)
    itok = ref_cpu.assemble_image_tokens(views, vsd["start_img"], vsd["end_img"])
    want = ref_cpu.OracleDecoder(oargs, sd).forward(ex, itok)
    out = m(ex.to(DEV), img.to(DEV))
    assert rel_err(out, want.numpy()) < REL
    # bf16 path, same inputs
    m.to(BF)
    outb = m(ex.to(DEV), img.to(DEV))
    assert rel_err(outb, want.numpy()) < 5e-2


@pytest.mark.parametrize("heads,kv,dim,B", [(4, 4, 256, 5), (4, 2, 512, 8), (2, 1, 256, 1), (8, 8, 512, 11), (2, 2, 128, 2),
                                            (4, 2, 512, 27), (4, 4, 256, 32)])      # 17..32 rows: two row chunks inside the C call
def test_fused_decode_step_matches_per_kernel_path(heads, kv, dim, B):
    """a3v_llama_decode_step's fused form (RMSNorm folded into the consuming GEMV, RoPE + KV write in the QKV epilogue,
    attention combine in-kernel; dim = 128 has a single 128-wide K block, so the step falls back to the per-kernel sequence
    inside the same C call) vs the same step run kernel by kernel: same rounding points, so logits agree to bf16
    accumulation-order noise, the KV cache rows written at the decode positions included; and vs the bf16 oracle."""
    args = plugin.ModelArgs(dim=dim, n_layers=3, n_heads=heads, n_kv_heads=kv, vocab_size=640, multiple_of=128, max_seq_len=192)
    oargs = ref_cpu.OracleArgs(dim=dim, n_layers=3, n_heads=heads, n_kv_heads=kv, vocab_size=640, multiple_of=128, max_seq_len=192)
    sd = ref_cpu.make_decoder_weights(oargs, seed=11, std=0.06)
    m = plugin.Transformer(args)
    m.load_state_dict(sd)
    m.to(BF).to(DEV)
    g = torch.Generator().manual_seed(2)
    T0, steps = 37, 6
    ex = torch.randint(3, 640, (B, T0 + steps), generator=g)
    ex[:, 0] = 1
    dec = ref_cpu.OracleDecoder(oargs, {k: v.to(BF) for k, v in sd.items()})
    want = [dec.forward_inference(ex[:, :T0], 0).float()]
    for t in range(T0, T0 + steps):
        want.append(dec.forward_inference(ex[:, t:t + 1], t).float())
    exd = ex.to(DEV)
    outs = {}
    for mode in (False, True):
        m._per_kernel_decode = mode
        lg = [m.forward_inference(exd[:, :T0], 0).float().clone()]
        for t in range(T0, T0 + steps):
            lg.append(m.forward_inference(exd[:, t:t + 1], t).float().clone())
        outs[mode] = (lg, [k[:, :, :T0 + steps].clone() for k in m._k_cache], [v[:, :, :, :T0 + steps].clone() for v in m._vt_cache])
    scale = max(float(w.abs().max()) for w in want)
    for i in range(steps + 1):
        assert float((outs[False][0][i] - outs[True][0][i]).abs().max()) / scale < 2e-2, i
        assert float((outs[False][0][i].cpu() - want[i]).abs().max()) / scale < 4e-2, i
    for l in range(3):
        ks = float(outs[True][1][l].float().abs().max())
        assert float((outs[False][1][l].float() - outs[True][1][l].float()).abs().max()) / ks < 2e-2
        vs = float(outs[True][2][l].float().abs().max())
        assert float((outs[False][2][l].float() - outs[True][2][l].float()).abs().max()) / vs < 2e-2
    # the arrival counters are left at zero
    ws = m._ws["skinny_ws"]
    assert int(ws[:8192].view(torch.int32).abs().sum()) == 0


def test_prefill_fused_qkv_rope_matches_separate_kernels():
    """forward_inference prefill + two decode steps with the qkv/RoPE/KV-write fusion on and off: identical logits and caches
    (the fused epilogue reproduces the separate kernels' values exactly)."""
    args = plugin.ModelArgs(dim=256, n_layers=3, n_heads=4, n_kv_heads=2, vocab_size=640, multiple_of=128, max_seq_len=192)
    oargs = ref_cpu.OracleArgs(dim=256, n_layers=3, n_heads=4, n_kv_heads=2, vocab_size=640, multiple_of=128, max_seq_len=192)
    sd = ref_cpu.make_decoder_weights(oargs, seed=12, std=0.06)
    m = plugin.Transformer(args)
    m.load_state_dict(sd)
    m.to(BF).to(DEV)
    g = torch.Generator().manual_seed(3)
    B, T0 = 4, 61
    ex = torch.randint(3, 640, (B, T0 + 2), generator=g)
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
    for l in range(3):
        assert torch.equal(got[True][1][l][:, :, :T0 + 2], got[False][1][l][:, :, :T0 + 2])
        assert torch.equal(got[True][2][l][:, :, :, :T0 + 2], got[False][2][l][:, :, :, :T0 + 2])
    dec = ref_cpu.OracleDecoder(oargs, {k: v.to(BF) for k, v in sd.items()})
    want = dec.forward_inference(ex[:, :T0], 0).float()
    assert float((got[True][0][0].cpu() - want).abs().max()) / float(want.abs().max()) < 4e-2


def test_attach_hf_qformer_provider(gold):
    """N4: the BLIP-2 Q-Former stream through stock transformers.Blip2Model behind the plugin's hook (the package the reference
    itself instantiates, llama_ens5.py:285-293): module registered as ``qformer`` (checkpoint keys load by name), frozen, features
    computed on the plugin's own five views; logits equal the oracle fed with the same module's features on the CPU views."""
    transformers = pytest.importorskip("transformers")
    from transformers import Blip2Config, Blip2QFormerConfig, Blip2VisionConfig, OPTConfig
    from a3vlm_amd.model.encoders import attach_qformer, convnext_tokens as cnx_tokens, dinov2_input
    V = gold["j"]["vocab_size"]
    args = plugin.ModelArgs(vocab_size=V, **{**TINY, "max_seq_len": 1600}, vit_width=VIT["width"], vit_layers=VIT["layers"],
                            vit_heads=VIT["heads"], vit_patch=VIT["patch"], vit_crop=224, n_views=5, qformer_tokens=32)
    m = plugin.Transformer(args, with_visual=True)
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=VIT["width"], layers=VIT["layers"], patch=VIT["patch"], grid=VIT["grid"],
                                      with_qformer=True, seed=1, std=0.05)
    m.load_state_dict({**sd, **vsd}, strict=True)
    m.to(torch.float32).to(DEV)
    torch.manual_seed(3)
    cfg = Blip2Config(vision_config=Blip2VisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                                      image_size=224, patch_size=14).to_dict(),
                      qformer_config=Blip2QFormerConfig(hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=128,
                                                        encoder_hidden_size=32, vocab_size=100, max_position_embeddings=32).to_dict(),
                      text_config=OPTConfig(vocab_size=64, hidden_size=16, num_hidden_layers=1, ffn_dim=16, num_attention_heads=2,
                                            max_position_embeddings=32, word_embed_proj_dim=16).to_dict(), num_query_tokens=32)
    attach_qformer(m, config=cfg)
    assert any(k.startswith("qformer.qformer.") for k in m.state_dict()) and not any(p.requires_grad for p in m.qformer.parameters())
    assert not any(n.startswith("qformer.") for n in m.get_trainable_params())
    B = 2
    img = synth_image(B)
    g = torch.Generator().manual_seed(5)
    ex = torch.randint(3, V, (B, 9), generator=g)
    ex[:, 0] = 1
    out = m(ex.to(DEV), img.to(DEV)).float().cpu()
    assert m.image_words == (32 + 257 + 2) * 5
    # oracle: same HF module (CPU copy) on the oracle's own views
    import copy
    q_cpu = copy.deepcopy(m.qformer).to("cpu")
    with torch.no_grad():
        qf = q_cpu.get_qformer_features(pixel_values=ref_cpu.split_views(img, 224))
        qf = getattr(qf, "last_hidden_state", qf)
    views = ref_cpu.encode_image(img, vsd, vit_layers=VIT["layers"], vit_heads=VIT["heads"], n_views=5, patch=VIT["patch"], qformer_feats=qf)
    itok = ref_cpu.assemble_image_tokens(views, vsd["start_img"], vsd["end_img"])
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=V, **{**TINY, "max_seq_len": 1600}), sd)
    want = dec.forward(ex, itok).float()
    assert rel_err(out, want.numpy()) < 5e-3
    # the two pure post-processing helpers of the other providers (no third-party net needed to check them)
    fm = torch.arange(2 * 3072 * 64, dtype=torch.float32).view(2, 3072, 8, 8)
    t = cnx_tokens(fm)
    assert t.shape == (2, 257, 3072) and torch.allclose(t[:, 0], t[:, 1:].mean(1))
    assert float(t[0, 1, 5]) == float(fm[0, 5, 0, 0]) == float(t[0, 2, 5]) and float(t[0, 3, 5]) == float(fm[0, 5, 0, 1])
    x = torch.rand(1, 3, 4, 4)
    from a3vlm_amd.data.transform import CLIP_MEAN, CLIP_STD
    xn = (x - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)
    want_d = (x - torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    assert torch.allclose(dinov2_input(xn), want_d, atol=1e-5)


# ------------------------------------------------------------------ N4: all three frozen encoders of the ensemble run once
class _TinyConvNeXtTrunk(torch.nn.Module):
    """Architecture-shaped stand-in for the timm ConvNeXt trunk open_clip builds (stem 4x4/4, three 2x2/2 down-samplings -> stride
    32: 256 -> 8; depthwise 7x7 + pointwise MLP blocks; ``head`` with ``global_pool`` / ``flatten`` that the provider must neutralise).
    Nothing of timm / open_clip is vendored: this only has the SHAPE CONTRACT the reference relies on (llama_ens5.py:304-315, 402-405)."""

    class _Head(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.global_pool = torch.nn.AdaptiveAvgPool2d(1)
            self.flatten = torch.nn.Flatten(1)

        def forward(self, x):
            return self.flatten(self.global_pool(x))

    def __init__(self, out_ch):
        super().__init__()
        ch = [out_ch // 8, out_ch // 4, out_ch // 2, out_ch]
        self.stem = torch.nn.Conv2d(3, ch[0], 4, 4)
        self.down = torch.nn.ModuleList([torch.nn.Conv2d(ch[i], ch[i + 1], 2, 2) for i in range(3)])
        self.dw = torch.nn.ModuleList([torch.nn.Conv2d(c, c, 7, padding=3, groups=c) for c in ch])
        self.pw = torch.nn.ModuleList([torch.nn.Conv2d(c, c, 1) for c in ch])
        self.head = self._Head()

    def forward(self, x):
        x = self.stem(x)
        for i in range(4):
            x = x + self.pw[i](torch.nn.functional.gelu(self.dw[i](x)))
            if i < 3:
                x = self.down[i](x)
        return self.head(x)


class _TinyDinoV2(torch.nn.Module):
    """Stand-in with DINOv2's ``forward_features`` contract: patch 14 on 224 x 224 -> 256 patch tokens + cls, normalised."""

    def __init__(self, dim):
        super().__init__()
        self.patch_embed = torch.nn.Conv2d(3, dim, 14, 14)
        self.cls_token = torch.nn.Parameter(torch.randn(1, 1, dim) * 0.02)
        self.pos = torch.nn.Parameter(torch.randn(1, 257, dim) * 0.02)
        self.blocks = torch.nn.ModuleList([torch.nn.TransformerEncoderLayer(dim, 2, dim * 2, dropout=0.0, batch_first=True, norm_first=True)
                                           for _ in range(2)])
        self.norm = torch.nn.LayerNorm(dim)

    def forward_features(self, x):
        t = self.patch_embed(x).flatten(2).transpose(1, 2)
        t = torch.cat([self.cls_token.expand(t.shape[0], -1, -1), t], dim=1) + self.pos
        for b in self.blocks:
            t = b(t)
        t = self.norm(t)
        return {"x_norm_clstoken": t[:, 0], "x_norm_patchtokens": t[:, 1:]}


def test_attach_reference_encoders_end_to_end_with_stand_ins(gold):
    """N4: ``attach_reference_encoders`` runs all three frozen streams of the reference's ensemble
    (llama_ens5.py:283-322, 399-440) behind the plugin hooks -- the real BLIP-2 Q-Former class from ``transformers`` at reduced
    width, and architecture-shaped stand-ins for the ConvNeXt trunk / DINOv2 (open_clip and torch.hub are not in this image; the
    stand-ins carry exactly the contracts the reference relies on: trunk(256 x 256) -> [N, C, 8, 8] with a poolable head,
    forward_features -> cls + 256 patch tokens).  Checked: modules registered under the reference's attribute names (checkpoint keys
    load by name) and frozen; the pre/post-processing (fp16 round trip + nearest resize to 256, 2x repeat to 16 x 16, mean token,
    CLIP -> ImageNet re-normalisation); and the ORDER of the concatenated ``visual_proj`` input [CLIP | ConvNeXt | DINOv2]
    (:436-440, G8) -- logits equal the oracle fed the same modules' features, and differ when the two extra streams are swapped."""
    pytest.importorskip("transformers")
    import copy
    from transformers import Blip2Config, Blip2QFormerConfig, Blip2VisionConfig, OPTConfig
    from a3vlm_amd.model.encoders import attach_reference_encoders
    V = gold["j"]["vocab_size"]
    C_CNX, C_DINO = 48, 24
    args = plugin.ModelArgs(vocab_size=V, **{**TINY, "max_seq_len": 1600}, vit_width=VIT["width"], vit_layers=VIT["layers"],
                            vit_heads=VIT["heads"], vit_patch=VIT["patch"], vit_crop=224, n_views=5, qformer_tokens=32,
                            extra_feat_dim=C_CNX + C_DINO)
    m = plugin.Transformer(args, with_visual=True)
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=VIT["width"], layers=VIT["layers"], patch=VIT["patch"], grid=VIT["grid"],
                                      in_feat=VIT["width"] + C_CNX + C_DINO, with_qformer=True, seed=1, std=0.05)
    m.load_state_dict({**sd, **vsd}, strict=True)
    m.to(torch.float32).to(DEV)
    torch.manual_seed(3)
    cfg = Blip2Config(vision_config=Blip2VisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                                      image_size=224, patch_size=14).to_dict(),
                      qformer_config=Blip2QFormerConfig(hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=128,
                                                        encoder_hidden_size=32, vocab_size=100, max_position_embeddings=32).to_dict(),
                      text_config=OPTConfig(vocab_size=64, hidden_size=16, num_hidden_layers=1, ffn_dim=16, num_attention_heads=2,
                                            max_position_embeddings=32, word_embed_proj_dim=16).to_dict(), num_query_tokens=32)
    trunk, dino = _TinyConvNeXtTrunk(C_CNX), _TinyDinoV2(C_DINO)
    trunk_cpu, dino_cpu = copy.deepcopy(trunk), copy.deepcopy(dino)
    attach_reference_encoders(m, qformer_config=cfg, convnext_trunk=trunk, dinov2_net=dino)
    keys = set(m.state_dict())
    for prefix in ("qformer.", "openclip_convnext_xxl.", "dinov2_vitg14."):
        assert any(k.startswith(prefix) for k in keys), prefix
        assert not any(n.startswith(prefix) for n in m.get_trainable_params())
    assert isinstance(m.openclip_convnext_xxl.head.global_pool, torch.nn.Identity) and len(m.extra_feat_fns) == 2
    assert not any(p.requires_grad for mod in (m.qformer, m.openclip_convnext_xxl, m.dinov2_vitg14) for p in mod.parameters())
    B = 2
    img = synth_image(B)
    g = torch.Generator().manual_seed(5)
    ex = torch.randint(3, V, (B, 9), generator=g)
    ex[:, 0] = 1
    out = m(ex.to(DEV), img.to(DEV)).float().cpu()
    assert m.image_words == (32 + 257 + 2) * 5
    # oracle: the reference's arithmetic around the same three modules (CPU copies) on the oracle's own views
    views_px = ref_cpu.split_views(img, 224)
    q_cpu = copy.deepcopy(m.qformer).to("cpu")
    with torch.no_grad():
        qf = q_cpu.get_qformer_features(pixel_values=views_px)
        qf = getattr(qf, "last_hidden_state", qf)
    trunk_cpu.head.global_pool = torch.nn.Identity()
    trunk_cpu.head.flatten = torch.nn.Identity()
    extra = ref_cpu.ensemble_extra_feats(views_px, trunk_cpu.eval(), dino_cpu.eval())
    assert extra[0].shape == (5 * B, 257, C_CNX) and extra[1].shape == (5 * B, 257, C_DINO)
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=V, **{**TINY, "max_seq_len": 1600}), sd)

    def oracle(extra_feats):
        views = ref_cpu.encode_image(img, vsd, vit_layers=VIT["layers"], vit_heads=VIT["heads"], n_views=5, patch=VIT["patch"], qformer_feats=qf,
                                     extra_feats=extra_feats)
        return dec.forward(ex, ref_cpu.assemble_image_tokens(views, vsd["start_img"], vsd["end_img"])).float()
    want = oracle(extra)
    assert rel_err(out, want.numpy()) < 5e-3
    # the column order [CLIP | ConvNeXt | DINOv2] matters: the swapped concatenation (widths permuting with it) is a different model
    pad = torch.zeros(5 * B, 257, C_CNX - C_DINO)
    swapped = oracle([torch.cat([extra[1], pad], dim=2), extra[0][:, :, :C_DINO]])
    assert rel_err(out, swapped.numpy()) > 20 * rel_err(out, want.numpy())
