"""The CPU oracle (oracle/ref_cpu.py) against the golden vectors captured from the
reference itself by oracle/gen_golden.py.  CPU only; fp32 round-off tolerances."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.gen_golden import TINY, VIT, checksum, convnext_tokens, extra_feature_inputs, synth_image

ATOL = 2e-5


def close(a, b, atol=ATOL, rtol=1e-5):
    a = a.detach().float().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.fixture(scope="module")
def dec(golden_dir):
    return np.load(os.path.join(golden_dir, "decoder_tiny.npz"))


@pytest.fixture(scope="module")
def meta(golden_dir):
    with open(os.path.join(golden_dir, "meta_tiny.json")) as f:
        j = json.load(f)
    return np.load(os.path.join(golden_dir, "meta_tiny.npz")), j


@pytest.fixture(scope="module")
def tiny(meta):
    V = meta[1]["vocab_size"]
    args = ref_cpu.OracleArgs(vocab_size=V, **TINY)
    sd = ref_cpu.make_decoder_weights(args, seed=0, std=0.08)
    return args, sd


def test_weight_checksum(dec, tiny):
    assert abs(checksum(tiny[1]) - float(dec["dec_weight_checksum"])) < 1e-6 * float(dec["dec_weight_checksum"])


def test_g1_rmsnorm(dec):
    x, w = torch.from_numpy(dec["g1_x"]), torch.from_numpy(dec["g1_w"])
    close(ref_cpu.rmsnorm(x, w, 1e-5), dec["g1_y"])
    yb = ref_cpu.rmsnorm(x.bfloat16(), w.bfloat16(), 1e-5)
    assert yb.dtype == torch.bfloat16
    close(yb, dec["g1_y_bf16"], atol=0, rtol=0)  # bit-exact in bf16


def test_g2_rope(dec):
    fc = ref_cpu.precompute_freqs_cis(16, 128)
    close(fc.real, dec["g2_freqs_re"], atol=1e-6)
    close(fc.imag, dec["g2_freqs_im"], atol=1e-6)
    oq, ok = ref_cpu.apply_rotary_emb(torch.from_numpy(dec["g2_xq"]), torch.from_numpy(dec["g2_xk"]), fc[3:8])
    close(oq, dec["g2_oq"])
    close(ok, dec["g2_ok"])


def test_state_dict_keys(dec, tiny):
    assert sorted(tiny[1].keys()) == list(dec["dec_state_keys"])
    assert sorted(tiny[1].keys()) == list(dec["dec_trainable"])  # with_visual=False: everything trains


def test_g4_forward_logits(dec, tiny):
    d = ref_cpu.OracleDecoder(*tiny)
    ex = torch.from_numpy(dec["g4_examples"])
    close(d.forward(ex), dec["g4_logits"])
    # the independent pin of the restated RoPE: transformers' LlamaForCausalLM
    close(d.forward(ex), dec["hf_logits"], atol=1e-4)


def test_g4_forward_inference_and_cache(dec, tiny):
    d = ref_cpu.OracleDecoder(*tiny)
    ex = torch.from_numpy(dec["g4_examples"])
    lg = [d.forward_inference(ex[:, :7], 0)]
    for t in range(7, 11):
        lg.append(d.forward_inference(ex[:, t:t + 1], t))
    close(torch.stack(lg), dec["g4_inf_logits"])
    close(d.k_cache[0][:2, :11], dec["g4_kcache_l0"])
    close(d.v_cache[1][:2, :11], dec["g4_vcache_l1"])


def test_g3_attention(dec, tiny):
    d = ref_cpu.OracleDecoder(*tiny)
    xa = torch.from_numpy(dec["g3_x"])
    close(d.attention(0, xa, 0, d.freqs_cis[:6], "causal"), dec["g3_causal"])
    d.allocate_kv_cache(2)
    d.attention(0, xa[:, :5], 0, d.freqs_cis[:5], "causal")
    close(d.attention(0, xa[:, 5:6], 5, d.freqs_cis[5:6], None), dec["g3_decode"])
    d.destroy_kv_cache()
    d.allocate_kv_cache(2)
    d.attention(0, xa[:, :3], 0, d.freqs_cis[:3], "causal")
    close(d.attention(0, xa[:, 3:5], 3, d.freqs_cis[3:5], "causal"), dec["g3_chunk"])


def test_g4_bf16_scale(dec, tiny):
    """bf16 run of the oracle vs the bf16 run of the reference (CPU kernels differ in
    accumulation order -> bf16-ulp level agreement only)."""
    args, sd = tiny
    d = ref_cpu.OracleDecoder(args, {k: v.bfloat16() for k, v in sd.items()})
    out = d.forward(torch.from_numpy(dec["g4_examples"])).float().numpy()
    ref = dec["g4_logits_bf16"]
    assert np.abs(out - ref).max() < 0.06 * np.abs(ref).max()


def test_g5_loss(meta, tiny):
    m, _ = meta
    d = ref_cpu.OracleDecoder(*tiny)
    ex = torch.from_numpy(m["g5_examples"])
    close(ref_cpu.meta_forward_loss(d, ex, torch.from_numpy(m["g5_labels_a"])), m["g5_loss_a"])
    close(ref_cpu.meta_forward_loss(d, torch.from_numpy(m["g5_examples_b"]), torch.from_numpy(m["g5_labels_b"])),
          m["g5_loss_b"])
    lc = ref_cpu.meta_forward_loss(d, ex, torch.zeros_like(ex))
    assert float(lc) == 0.0 and float(m["g5_loss_c"]) == 0.0


@pytest.mark.parametrize("key,max_gen,stops", [("gen12", 12, ()), ("gen48", 48, ()), ("genstop", 12, ("li", "ab"))])
def test_g6_generate(meta, tiny, golden_dir, key, max_gen, stops):
    import sentencepiece as spm
    _, j = meta
    sp = spm.SentencePieceProcessor(model_file=os.path.join(golden_dir, "tokenizer.model"))
    prompts = [[sp.bos_id()] + sp.encode(p) for p in j["prompts"]]
    assert prompts == j["prompt_ids"]
    extra = []
    if stops:  # Tokenizer.encode_segment / encode_wo_prefix_space (model/tokenizer.py:64-88), spm style
        assert j["need_space_before_segment"] is False
        extra += [sp.encode(s.lstrip(" ")) for s in stops]
        for s in stops:
            for prefix in ["@", "\n", "\\", "=", ">", "`"]:
                pt, ct = sp.encode(prefix), sp.encode(prefix + s)
                if ct[:len(pt)] == pt:
                    extra.append(ct[len(pt):])
                    break
    d = ref_cpu.OracleDecoder(*tiny)
    _, outs = ref_cpu.generate_greedy(d, prompts, max_gen_len=max_gen, eos_id=sp.eos_id(), extra_stop=extra)
    assert outs == j[key + "_ids"]
    assert [sp.decode(o) for o in outs] == j[key + "_text"]


def test_g7_g8_vision(golden_dir, tiny):
    v = np.load(os.path.join(golden_dir, "vision_tiny.npz"))
    args, sd = tiny
    vsd = ref_cpu.make_vision_weights(64, width=VIT["width"], layers=VIT["layers"], patch=VIT["patch"],
                                      grid=VIT["grid"], in_feat=VIT["width"] + 3072 + 1536,
                                      with_qformer=True, seed=1, std=0.05)
    assert abs(checksum(vsd) - float(v["vis_weight_checksum"])) < 1e-6 * float(v["vis_weight_checksum"])
    assert sorted(vsd.keys()) == list(v["vis_state_keys"])
    B = 2
    img = synth_image(B)
    assert abs(img.double().abs().sum().item() - float(v["image_checksum"])) < 1e-3
    crops = ref_cpu.split_views(img, 224)
    close(ref_cpu.clip_encode_image(crops[:3], vsd, VIT["layers"], VIT["heads"], VIT["patch"]), v["g7_clip_feats"], atol=1e-4)
    qf, cnx, dino = extra_feature_inputs(5 * B)
    views = ref_cpu.encode_image(img, vsd, vit_layers=VIT["layers"], vit_heads=VIT["heads"], n_views=5,
                                 qformer_feats=qf, extra_feats=[convnext_tokens(cnx), dino])
    close(torch.stack(views), v["g8_views"], atol=1e-4)
    itok = ref_cpu.assemble_image_tokens(views, vsd["start_img"], vsd["end_img"])
    assert itok.shape[1] == 1455 == int(v["g8_cache_image_words"])
    args_v = ref_cpu.OracleArgs(vocab_size=args.vocab_size, **{**TINY, "max_seq_len": 1600})
    d = ref_cpu.OracleDecoder(args_v, sd)
    ex = torch.from_numpy(v["g8_examples"])
    close(d.forward(ex, itok), v["g8_logits"], atol=1e-4)
    l0 = d.forward_inference(ex[:, :6], 0, itok)
    l1 = d.forward_inference(ex[:, 6:7], 6)
    l2 = d.forward_inference(ex[:, 7:8], 7)
    close(torch.stack([l0, l1, l2]), v["g8_inf_logits"], atol=1e-4)


def test_g8b_two_image_plugin(golden_dir, tiny):
    """llama_ens5_2images: [BOS | RGB words | depth words (own tags) | text]; LM head from visual_image_words on;
    cached inference inserts image words only when both images are given."""
    from oracle.gen_golden import depth_tags
    v = np.load(os.path.join(golden_dir, "vision2_tiny.npz"))
    args, sd = tiny
    vsd = ref_cpu.make_vision_weights(64, width=VIT["width"], layers=VIT["layers"], patch=VIT["patch"],
                                      grid=VIT["grid"], in_feat=VIT["width"] + 3072 + 1536, with_qformer=True, seed=1, std=0.05)
    B = 2
    img, depth = synth_image(B, seed=5), synth_image(B, seed=6)
    assert abs(depth.double().abs().sum().item() - float(v["depth_checksum"])) < 1e-3
    qf, cnx, dino = extra_feature_inputs(5 * B)
    kw = dict(vit_layers=VIT["layers"], vit_heads=VIT["heads"], n_views=5, qformer_feats=qf, extra_feats=[convnext_tokens(cnx), dino])
    sdi, edi = depth_tags()
    t_rgb = ref_cpu.assemble_image_tokens(ref_cpu.encode_image(img, vsd, **kw), vsd["start_img"], vsd["end_img"])
    t_dep = ref_cpu.assemble_image_tokens(ref_cpu.encode_image(depth, vsd, **kw), sdi, edi)
    itok = torch.cat([t_rgb, t_dep], dim=1)
    assert itok.shape[1] == 2910 == int(v["cache_image_words"])
    d = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(vocab_size=args.vocab_size, **{**TINY, "max_seq_len": 3200}), sd)
    ex = torch.from_numpy(v["examples"])
    full = d.forward(ex, itok, out_from=1455)
    assert list(full.shape) == list(v["logits_2img_shape"])
    close(full[:, -12:], v["logits_2img_tail"], atol=1e-4)
    close(full.sum(-1), v["logits_2img_rowsum"], atol=2e-3)
    close(d.forward(ex, t_rgb), v["logits_rgb_only"], atol=1e-4)
    l0 = d.forward_inference(ex[:, :6], 0, itok)
    l1 = d.forward_inference(ex[:, 6:7], 6)
    l2 = d.forward_inference(ex[:, 7:8], 7)
    close(torch.stack([l0, l1, l2]), v["inf_logits"], atol=1e-4)
    close(d.forward_inference(ex[:, :6], 0), v["inf_logits_rgb_dropped"], atol=1e-4)


def test_g13_lora_linear(golden_dir):
    """model/peft.py: y = W x + lora_b(lora_a(x)) (no alpha / rank scaling), key names <linear>.lora_{a,b}.weight."""
    v = np.load(os.path.join(golden_dir, "lora_tiny.npz"))
    for name in ("plain", "col", "row"):
        sd = {"l.weight": torch.from_numpy(v[name + "_w"]), "l.lora_a.weight": torch.from_numpy(v[name + "_a"]),
              "l.lora_b.weight": torch.from_numpy(v[name + "_b"])}
        d = ref_cpu.OracleDecoder.__new__(ref_cpu.OracleDecoder)
        d.sd = sd
        close(d.lin(torch.from_numpy(v[name + "_x"]), "l"), v[name + "_y"], atol=1e-5)
        assert {"lora_a.weight", "lora_b.weight", "weight"} <= set(v[name + "_keys"].tolist())
