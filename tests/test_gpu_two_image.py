"""-m gpu: the two-image (RGB + depth) plugin ``llama_ens5_2images`` against the reference fixture (fp32 parity path),
greedy generation through MetaModel with depth images, and a training step (RGB + depth) against oracle autograd."""
import os

import numpy as np
import pytest
import torch

from a3vlm_amd.model.LLM import llama_ens5_2images as plugin2
from a3vlm_amd.model.meta import MetaModel
from a3vlm_amd.train import TrainEngine
from a3vlm_amd.util import promote_trainable_params_to_fp32
from oracle import ref_cpu
from oracle.gen_golden import TINY, VIT, convnext_tokens, depth_tags, extra_feature_inputs, synth_image

pytestmark = pytest.mark.gpu
DEV = "cuda"
REL = 1e-3


def rel_err(got, want):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else got
    want = want.detach().float().cpu().numpy() if isinstance(want, torch.Tensor) else want
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


def test_two_image_fixture_fp32(golden_dir):
    v = np.load(os.path.join(golden_dir, "vision2_tiny.npz"))
    V = 192
    args = plugin2.ModelArgs(vocab_size=V, **{**TINY, "max_seq_len": 3200}, vit_width=VIT["width"], vit_layers=VIT["layers"],
                             vit_heads=VIT["heads"], vit_patch=VIT["patch"], vit_crop=224, n_views=5,
                             extra_feat_dim=3072 + 1536, qformer_tokens=32)
    m = plugin2.Transformer(args, with_visual=True)
    assert m.image_words == 2910 and m.visual_image_words == 1455
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=VIT["width"], layers=VIT["layers"], patch=VIT["patch"], grid=VIT["grid"],
                                      in_feat=VIT["width"] + 3072 + 1536, with_qformer=True, seed=1, std=0.05)
    sdi, edi = depth_tags()
    m.load_state_dict({**sd, **vsd, "start_depth_img": sdi, "end_depth_img": edi}, strict=True)
    m.to(DEV)
    B = 2
    img, depth = synth_image(B, seed=5).to(DEV), synth_image(B, seed=6).to(DEV)
    qf, cnx, dino = extra_feature_inputs(5 * B)
    qf2 = torch.cat([qf, qf]).to(DEV)                              # same hook features for both images (fixture)
    ex2 = [torch.cat([convnext_tokens(cnx)] * 2).to(DEV), torch.cat([dino, dino]).to(DEV)]
    ex1 = [convnext_tokens(cnx).to(DEV), dino.to(DEV)]
    ex = torch.from_numpy(v["examples"]).to(DEV)
    full = m(ex, img, depth, qformer_feats=qf2, extra_feats=ex2)
    assert list(full.shape) == list(v["logits_2img_shape"])
    assert rel_err(full[:, -12:], v["logits_2img_tail"]) < REL
    assert rel_err(full.sum(-1), v["logits_2img_rowsum"]) < REL
    assert rel_err(m(ex, img, qformer_feats=qf.to(DEV), extra_feats=ex1), v["logits_rgb_only"]) < REL
    l0 = m.forward_inference(ex[:, :6], 0, img, depth, qformer_feats=qf2, extra_feats=ex2).clone()
    assert m.cache_image_words == int(v["cache_image_words"]) == 2910
    l1 = m.forward_inference(ex[:, 6:7], 6).clone()
    l2 = m.forward_inference(ex[:, 7:8], 7).clone()
    for i, l in enumerate((l0, l1, l2)):
        assert rel_err(l, v["inf_logits"][i]) < REL, i
        assert (l.argmax(-1).cpu().numpy() == v["inf_logits"][i].argmax(-1)).all()
    # image without depth is ignored by the cached path (llama_ens5_2images.py:517)
    l = m.forward_inference(ex[:, :6], 0, img, None)
    assert m.cache_image_words == 0 and rel_err(l, v["inf_logits_rgb_dropped"]) < REL


TK = dict(dim=64, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=192, multiple_of=64, max_seq_len=1024)


def _small(dtype):
    args = plugin2.ModelArgs(**TK, vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)
    m = plugin2.Transformer(args, with_visual=True)
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(**TK), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=64, layers=2, patch=14, grid=8, seed=1, std=0.05)
    sdi, edi = depth_tags()
    m.load_state_dict({**sd, **vsd, "start_depth_img": sdi, "end_depth_img": edi})
    for n, p in m.named_parameters():
        p.requires_grad = not n.startswith("clip.")
    m.to(dtype).to(DEV)
    return m, sd, vsd, sdi, edi


def test_two_image_train_step_fp32_matches_autograd():
    m, sd, vsd, sdi, edi = _small(torch.float32)
    promote_trainable_params_to_fp32(m)
    g = torch.Generator().manual_seed(9)
    B, T = 2, 10
    ex = torch.randint(3, 192, (B, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :3] = 0
    img, depth = synth_image(B, size=112, seed=3), synth_image(B, size=112, seed=4)
    # oracle autograd over [BOS | rgb words | depth words | text]
    sdg = {k: t.clone().requires_grad_(True) for k, t in sd.items()}
    vg = {k: t.clone().requires_grad_(k.startswith(("visual_proj", "start_img", "end_img"))) for k, t in vsd.items()}
    tg = [sdi.clone().requires_grad_(True), edi.clone().requires_grad_(True)]
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(**TK), sdg)
    kw = dict(vit_layers=2, vit_heads=4, n_views=1)
    itok = torch.cat([ref_cpu.assemble_image_tokens(ref_cpu.encode_image(img, vg, **kw), vg["start_img"], vg["end_img"]),
                      ref_cpu.assemble_image_tokens(ref_cpu.encode_image(depth, vg, **kw), tg[0], tg[1])], dim=1)
    want_loss = ref_cpu.meta_forward_loss(dec, ex, lab, itok)
    want_loss.backward()
    want = {k: t.grad for k, t in sdg.items()}
    want.update({k: t.grad for k, t in vg.items() if t.requires_grad})
    want.update({"start_depth_img": tg[0].grad, "end_depth_img": tg[1].grad})
    eng = TrainEngine(m, torch.float32, recompute=False)
    loss = eng.forward_loss(ex.to(DEV), lab.to(DEV), [img.to(DEV), depth.to(DEV)])
    assert abs(float(loss) - float(want_loss)) < 1e-3 * abs(float(want_loss))
    eng.backward(1.0)
    for name, p in m.get_trainable_params().items():
        assert p.grad is not None, name
        assert rel_err(p.grad, want[name]) < 1e-3, name


def test_two_image_generate_matches_oracle(golden_dir):
    mm = MetaModel("llama_ens5_2images", [os.path.join(golden_dir, "tiny_params.json")], os.path.join(golden_dir, "tokenizer.model"),
                   with_visual=False, max_seq_len=1024)
    m, sd, vsd, sdi, edi = _small(torch.float32)
    m.args.max_seq_len = 1024
    mm.llma = m
    prompts = ["Detect the lid", "Where is the drawer handle of the cabinet"]
    B = len(prompts)
    img, depth = synth_image(B, size=112, seed=3), synth_image(B, size=112, seed=4)
    texts, ids = mm.generate(prompts, img, depth_images=depth, max_gen_len=12, temperature=0.0, return_ids=True)
    kw = dict(vit_layers=2, vit_heads=4, n_views=1)
    itok = torch.cat([ref_cpu.assemble_image_tokens(ref_cpu.encode_image(img, vsd, **kw), vsd["start_img"], vsd["end_img"]),
                      ref_cpu.assemble_image_tokens(ref_cpu.encode_image(depth, vsd, **kw), sdi, edi)], dim=1)
    dec = ref_cpu.OracleDecoder(ref_cpu.OracleArgs(**TK), sd)
    pt = [mm.tokenizer.encode(p, bos=True, eos=False) for p in prompts]
    _, want = ref_cpu.generate_greedy(dec, pt, image_tokens=itok, image_words=itok.shape[1], max_gen_len=12, eos_id=mm.tokenizer.eos_id)
    assert ids == want


def test_two_image_fp8_weights_1024_context():
    """BASELINE configs[4]: RGB + depth images, 1024-token prompt, fp8 weight path -- the combination, on one model.
    ``llama_ens5_2images`` in bf16 with ``quantize_decode_weights("fp8", prefill=True)``: W8A8 prefill over
    [BOS | RGB words | depth words | 1023 text tokens], then weight-only fp8 decode steps on the cache it wrote.
    No reference oracle exists for fp8: checked against the W8A8 restatement of the oracle
    (oracle/quant_fp8.py) fed the SAME two-image token assembly, and against the bf16 run of the same plugin."""
    from oracle.quant_fp8 import W8A8OracleDecoder
    kw = dict(dim=512, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=640, multiple_of=256, max_seq_len=2048)
    args = plugin2.ModelArgs(**kw, vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)
    m = plugin2.Transformer(args, with_visual=True)
    oargs = ref_cpu.OracleArgs(**kw)
    sd = ref_cpu.make_decoder_weights(oargs, seed=31, std=0.05)
    vsd = ref_cpu.make_vision_weights(512, width=64, layers=2, patch=14, grid=8, seed=1, std=0.05)
    sdi, edi = depth_tags(dim=512)
    m.load_state_dict({**sd, **vsd, "start_depth_img": sdi, "end_depth_img": edi})
    m.to(torch.bfloat16).to(DEV)
    assert m.image_words == 2 * (64 + 1 + 2)
    B, T = 2, 1024
    g = torch.Generator().manual_seed(17)
    ND = 12                                                     # teacher-forced decode steps after the prefill
    ex = torch.randint(3, 640, (B, T + ND), generator=g)
    ex[:, 0] = 1
    img, depth = synth_image(B, size=112, seed=3), synth_image(B, size=112, seed=4)
    exd, imgd, depd = ex.to(DEV), img.to(DEV), depth.to(DEV)
    base = m.forward_inference(exd[:, :T], 0, imgd, depd).float().clone()
    base_next = m.forward_inference(exd[:, T:T + 1], T).float().clone()
    m.quantize_decode_weights("fp8", prefill=True)
    got = m.forward_inference(exd[:, :T], 0, imgd, depd).float().clone()
    assert m.cache_image_words == m.image_words
    nxt = [m.forward_inference(exd[:, t:t + 1], t).float().clone() for t in range(T, T + ND)]
    # oracle: bf16 weights, the same image words, W8A8 arithmetic in the decoder linears
    bf = torch.bfloat16
    vb = {k: v.to(bf) for k, v in vsd.items()}
    kwv = dict(vit_layers=2, vit_heads=4, n_views=1)
    itok = torch.cat([ref_cpu.assemble_image_tokens(ref_cpu.encode_image(img.to(bf), vb, **kwv), vb["start_img"], vb["end_img"]),
                      ref_cpu.assemble_image_tokens(ref_cpu.encode_image(depth.to(bf), vb, **kwv), sdi.to(bf), edi.to(bf))], dim=1)
    dec = W8A8OracleDecoder.make(ref_cpu.OracleDecoder)(oargs, {k: v.to(bf) for k, v in sd.items()})
    want = dec.forward_inference(ex[:, :T], 0, itok).float()

    def rms(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).norm() / b.norm())
    # the yardstick comes from the oracle alone: how far ITS W8A8 arithmetic moves the logits away from ITS bf16 run
    dec_bf = ref_cpu.OracleDecoder(oargs, {k: v.to(bf) for k, v in sd.items()})
    want_bf = dec_bf.forward_inference(ex[:, :T], 0, itok).float()
    e_q = rms(want, want_bf)
    e_bf, e_or = rms(got, base), rms(got, want)
    print(f"2 images + fp8 @ T=1024: oracle W8A8 vs oracle bf16 {e_q:.4f}; device fp8 vs device bf16 {e_bf:.4f}, vs W8A8 oracle {e_or:.4f}")
    assert 0 < e_bf <= 1.5 * e_q, (e_bf, e_q)                  # quantisation moves the device's logits no further than the oracle's
    assert e_or <= 0.75 * e_q and e_or < 0.7 * e_bf, (e_or, e_q, e_bf)      # and the device sits closer to the W8A8 oracle than to bf16
    assert rms(nxt[0], base_next) <= 1.5 * e_q + 0.02
    # greedy ids of the decode steps (weight-only fp8 on the W8A8-written cache) against the W8A8 oracle: equal wherever the
    # oracle's top-2 margin exceeds the measured deviation, and on most positions outright
    agree = clear = 0
    for i, t in enumerate(range(T, T + ND)):
        w2 = dec.forward_inference(ex[:, t:t + 1], t).float()
        e_t = rms(nxt[i], w2)
        assert e_t <= 1.5 * e_q + 0.02, (i, e_t, e_q)
        dev_ids, ora_ids = nxt[i].argmax(-1).cpu(), w2.argmax(-1)
        top2 = w2.topk(2, dim=-1).values
        noise = (nxt[i].cpu() - w2).abs().max(dim=-1).values
        for b in range(B):
            agree += int(dev_ids[b] == ora_ids[b])
            if float(top2[b, 0] - top2[b, 1]) > 2 * float(noise[b]):
                clear += 1
                assert int(dev_ids[b]) == int(ora_ids[b]), (i, b)
    print(f"fp8 greedy ids: {agree}/{B * ND} equal to the W8A8 oracle, {clear} positions with a clear margin")
    assert agree >= 0.75 * B * ND
    m.quantize_decode_weights(None)
    assert torch.equal(m.forward_inference(exd[:, :T], 0, imgd, depd).float(), base)
