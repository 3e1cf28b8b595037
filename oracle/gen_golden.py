"""Generate the golden fixtures under ``tests/golden/`` by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs
``/root/reference``); the fixtures it writes are plain arrays + JSON (inputs and
expected outputs), never reference source or pickled reference classes.

    python -m oracle.gen_golden            # rewrites tests/golden/*

What is captured:
  decoder_tiny.npz   RMSNorm, RoPE table/apply, Attention (prefill/decode, GQA,
                     KV-cache contents), Transformer.forward logits,
                     forward_inference prefill + decode logits, HF cross-check
  meta_tiny.npz/json MetaModel.forward loss (normal / trailing pad / all-zero
                     labels), MetaModel.generate(temperature=0) ids + strings
  vision_tiny.npz    clip_encode_image via an open_clip-shaped stand-in,
                     encode_image (5 views, all four feature streams),
                     image-token assembly, forward / forward_inference with image
Weights are NOT stored: they are regenerated from seeds by
``oracle.ref_cpu.make_*_weights`` and guarded by checksums in the fixtures.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu, refimport  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

TINY = dict(dim=64, n_layers=2, n_heads=4, n_kv_heads=2, multiple_of=64, norm_eps=1e-5,
            rope_theta=10000.0, max_seq_len=64)
VIT = dict(width=64, layers=2, heads=4, patch=14, grid=16)

CORPUS = [
    "A chat between a curious human and an artificial intelligence assistant.",
    "The assistant gives helpful, detailed, and polite answers to the human's questions.",
    "### Human: Detect all manipulable object parts and provide their 3D bounding boxes.",
    "### Assistant: There are two manipulable object parts with their 3d bounding boxes:",
    "<box>lid</box>[[0.12,0.34,0.56],[0.22,0.31,0.50],[0.41,0.77,0.62],[0.09,0.18,0.27]]",
    "<axis>revolute</axis>[0.10,0.20,0.30,0.40,0.50,0.60] <box>drawer</box> <axis>prismatic</axis>",
    "Please provide the joint's type and its 3D axis linked to the object part.",
    "handle door button knob slider wheel leg seat lever switch cap",
    "0 1 2 3 4 5 6 7 8 9 0.00 0.25 0.50 0.75 1.00 Hi my darling",
]


def checksum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def np32(t):
    return t.detach().float().cpu().numpy()


def train_tokenizer(path_prefix: str, vocab: int = 192):
    import sentencepiece as spm
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for _ in range(20):
            f.write("\n".join(CORPUS) + "\n")
        txt = f.name
    spm.SentencePieceTrainer.train(input=txt, model_prefix=path_prefix, vocab_size=vocab,
                                   model_type="bpe", unk_id=0, bos_id=1, eos_id=2, pad_id=-1,
                                   character_coverage=1.0, byte_fallback=False,
                                   hard_vocab_limit=False, normalization_rule_name="identity")
    os.unlink(txt)
    os.unlink(path_prefix + ".vocab")


# ---- open_clip-shaped stand-in built from stock torch modules -----------------
class _ResBlock(nn.Module):
    """Pre-LN block using nn.MultiheadAttention -- the module open_clip's
    ResidualAttentionBlock wraps; gives the in_proj packing ground truth."""

    def __init__(self, w, h):
        super().__init__()
        self.ln_1 = nn.LayerNorm(w)
        self.attn = nn.MultiheadAttention(w, h)
        self.ln_2 = nn.LayerNorm(w)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(w, 4 * w))
        self.mlp.add_module("gelu", nn.GELU())
        self.mlp.add_module("c_proj", nn.Linear(4 * w, w))

    def forward(self, x):  # LND
        y = self.ln_1(x)
        x = x + self.attn(y, y, y, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class _VT(nn.Module):
    def __init__(self, w, layers, h):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(w, h) for _ in range(layers)])

    def forward(self, x):
        for b in self.resblocks:
            x = b(x)
        return x


class _Visual(nn.Module):
    def __init__(self, w, layers, h, patch, grid):
        super().__init__()
        self.conv1 = nn.Conv2d(3, w, patch, patch, bias=False)
        self.class_embedding = nn.Parameter(torch.zeros(w))
        self.positional_embedding = nn.Parameter(torch.zeros(grid * grid + 1, w))
        self.ln_pre = nn.LayerNorm(w)
        self.transformer = _VT(w, layers, h)
        self.ln_post = nn.LayerNorm(w)


class _Clip(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.visual = _Visual(**kw)


class _FixedFeat(nn.Module):
    """Stand-in for an OUT-OF-SCOPE frozen encoder: returns a fixed tensor."""

    def __init__(self, feat):
        super().__init__()
        self.feat = feat

    def forward(self, x):
        return self.feat


class _QF(nn.Module):
    def __init__(self, feat):
        super().__init__()
        self.feat = feat

    def get_qformer_features(self, pixel_values=None):
        return type("O", (), {"last_hidden_state": self.feat})()


class _Dino(nn.Module):
    def __init__(self, feat):
        super().__init__()
        self.feat = feat

    def forward_features(self, x):
        return {"x_norm_clstoken": self.feat[:, 0], "x_norm_patchtokens": self.feat[:, 1:]}


def extra_feature_inputs(n_img: int, seed: int = 7):
    """Synthetic outputs of the three out-of-scope encoders (inputs of the path)."""
    g = torch.Generator().manual_seed(seed)
    qf = torch.randn(n_img, 32, 768, generator=g)
    cnx = torch.randn(n_img, 3072, 8, 8, generator=g)
    dino = torch.randn(n_img, 257, 1536, generator=g)
    return qf, cnx, dino


def synth_image(B: int, size: int = 448, seed: int = 5):
    """Synthetic normalised image batch (fp16-representable values), regenerated
    from the seed by the tests instead of being stored (2.4 MB)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, size, size, generator=g).half().float()


def convnext_tokens(cnx):
    """LLM/llama_ens5.py:410-419 reshaping, applied to the stand-in output so the
    fixture can also be consumed as a [N,257,3072] token stream."""
    t = cnx.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2).flatten(-2).permute(0, 2, 1)
    return torch.cat([t.mean(dim=1, keepdim=True), t], dim=1)


def main():
    os.makedirs(GOLD, exist_ok=True)
    refimport.install()
    refimport.init_dist_ws1()
    torch.manual_seed(0)
    import accessory.model.LLM.llama_ens5 as ens5
    import accessory.model.meta as meta
    from accessory.model.components import RMSNorm
    from accessory.util import misc
    misc.setup_for_distributed(True)  # print(..., force=True) used at meta.py:245

    tok_prefix = os.path.join(GOLD, "tokenizer")
    train_tokenizer(tok_prefix)
    tok_path = tok_prefix + ".model"
    import sentencepiece as spm
    V = spm.SentencePieceProcessor(model_file=tok_path).vocab_size()

    out = {}
    # ---------------- G1 RMSNorm ----------------
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 7, 64, generator=g) * 3
    w = 1 + 0.1 * torch.randn(64, generator=g)
    n = RMSNorm(64, eps=1e-5)
    n.weight.data.copy_(w)
    out["g1_x"], out["g1_w"], out["g1_y"] = np32(x), np32(w), np32(n(x))
    nb = RMSNorm(64, eps=1e-5).bfloat16()
    nb.weight.data.copy_(w.bfloat16())
    out["g1_y_bf16"] = np32(nb(x.bfloat16()))

    # ---------------- G2 RoPE (restated helpers; pinned by the HF cross-check below) ------
    fc = ref_cpu.precompute_freqs_cis(16, 128)
    out["g2_freqs_re"], out["g2_freqs_im"] = np32(fc.real), np32(fc.imag)
    xq = torch.randn(2, 5, 4, 16, generator=g)
    xk = torch.randn(2, 5, 2, 16, generator=g)
    oq, ok = ref_cpu.apply_rotary_emb(xq, xk, fc[3:8])
    out["g2_xq"], out["g2_xk"], out["g2_oq"], out["g2_ok"] = np32(xq), np32(xk), np32(oq), np32(ok)

    # ---------------- G3/G4 decoder ----------------
    args = ens5.ModelArgs(vocab_size=V, max_batch_size=32, **TINY)
    model = ens5.Transformer(args, with_visual=False)
    oargs = ref_cpu.OracleArgs(vocab_size=V, **TINY)
    sd = ref_cpu.make_decoder_weights(oargs, seed=0, std=0.08)
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    out["dec_weight_checksum"] = np.float64(checksum(sd))
    out["dec_state_keys"] = np.array(sorted(model.state_dict().keys()))
    out["dec_trainable"] = np.array(sorted(model.get_trainable_params().keys()))

    ex = torch.randint(3, V, (2, 12), generator=g)
    ex[:, 0] = 1
    with torch.no_grad():
        out["g4_examples"] = ex.numpy()
        out["g4_logits"] = np32(model(ex))
        # prefill 7 then 4 decode steps (teacher-forced with ex)
        lg = [np32(model.forward_inference(ex[:, :7], 0))]
        for t in range(7, 11):
            lg.append(np32(model.forward_inference(ex[:, t:t + 1], t)))
        out["g4_inf_logits"] = np.stack(lg)
        out["g4_kcache_l0"] = np32(model.layers[0].attention.k_cache[:2, :11])
        out["g4_vcache_l1"] = np32(model.layers[1].attention.v_cache[:2, :11])
        # G3: Attention alone
        att = model.layers[0].attention
        xa = torch.randn(2, 6, 64, generator=g)
        att.destroy_kv_cache()
        out["g3_x"] = np32(xa)
        out["g3_causal"] = np32(att(xa, 0, model.freqs_cis[:6], "causal"))
        att.allocate_kv_cache(2, 64)
        att(xa[:, :5], 0, model.freqs_cis[:5], "causal")
        out["g3_decode"] = np32(att(xa[:, 5:6], 5, model.freqs_cis[5:6], None))
        # unequal q/kv causal (right aligned): 2 new tokens on a 3-token cache
        att.allocate_kv_cache(2, 64)
        att(xa[:, :3], 0, model.freqs_cis[:3], "causal")
        out["g3_chunk"] = np32(att(xa[:, 3:5], 3, model.freqs_cis[3:5], "causal"))
        model._destroy_kv_cache()
        # bf16 run of the reference itself (information: bf16 deviation scale)
        mb = ens5.Transformer(args, with_visual=False).bfloat16()
        mb.load_state_dict({k: v.bfloat16() for k, v in sd.items()})
        out["g4_logits_bf16"] = np32(mb(ex))

    # ---------------- HF cross-check (pins the RoPE convention) ----------------
    from accessory.tools.convert_weights_to_hf import convert_merged_ckpt_to_hf
    from transformers import LlamaConfig, LlamaForCausalLM
    params = dict(n_heads=TINY["n_heads"], n_kv_heads=TINY["n_kv_heads"])
    shards = convert_merged_ckpt_to_hf({"llma." + k: v for k, v in sd.items()}, params)
    hf_sd = {}
    for s in shards:
        hf_sd.update(s)
    ffn = ref_cpu.ffn_hidden_dim(64, 64, None)
    cfg = LlamaConfig(vocab_size=V, hidden_size=64, intermediate_size=ffn, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5,
                      rope_theta=10000.0, max_position_embeddings=128, attention_bias=False,
                      tie_word_embeddings=False)
    hf = LlamaForCausalLM(cfg).eval()
    hf.load_state_dict(hf_sd, strict=True)
    with torch.no_grad():
        out["hf_logits"] = np32(hf(ex).logits)
    d = float(np.abs(out["hf_logits"] - out["g4_logits"]).max())
    print(f"HF cross-check max|dlogit| = {d:.3e}")
    assert d < 1e-4, "RoPE convention mismatch vs transformers"

    np.savez_compressed(os.path.join(GOLD, "decoder_tiny.npz"), **out)

    # ---------------- G5/G6 MetaModel ----------------
    mo = {}
    cfg_path = os.path.join(GOLD, "tiny_params.json")
    with open(cfg_path, "w") as f:
        json.dump({k: v for k, v in TINY.items() if k != "max_seq_len"}, f)
    mm = meta.MetaModel("llama_ens5", cfg_path, tok_path, with_visual=False, max_seq_len=64)
    mm.llma.load_state_dict(sd)
    mm.eval()
    ex5 = torch.randint(3, V, (3, 14), generator=g)
    ex5[:, 0] = 1
    lab = ex5.clone()
    lab[:, :5] = 0
    with torch.no_grad():
        mo["g5_examples"] = ex5.numpy()
        mo["g5_labels_a"] = lab.numpy()
        mo["g5_loss_a"] = np32(mm(ex5, lab)[0])
        lab_b = lab.clone()
        lab_b[:, 10:] = 0          # trailing pad -> trimmed to 10 columns
        ex_b = ex5.clone()
        ex_b[:, 10:] = 0
        mo["g5_examples_b"], mo["g5_labels_b"] = ex_b.numpy(), lab_b.numpy()
        mo["g5_loss_b"] = np32(mm(ex_b, lab_b)[0])
        lab_c = torch.zeros_like(lab)
        mo["g5_loss_c"] = np32(mm(ex5, lab_c)[0])
    prompts = ["### Human: Detect all manipulable object parts.\n### Assistant:",
               "Hi my darling",
               "### Human: Please provide the joint's type and its 3D axis linked to the object part handle.\n### Assistant: <box>"]
    decoded_ids = []
    orig_decode = mm.tokenizer.decode
    mm.tokenizer.decode = lambda t: (decoded_ids.append(list(t)), orig_decode(t))[1]
    texts = mm.generate(prompts, None, max_gen_len=12, temperature=0.0)
    texts_trunc = None
    ids_a = [list(t) for t in decoded_ids]
    decoded_ids.clear()
    # left truncation + tight max_seq_len: max_seq_len=64, max_gen_len=48 -> prompts cut to 16
    texts_trunc = mm.generate(prompts, None, max_gen_len=48, temperature=0.0)
    ids_b = [list(t) for t in decoded_ids]
    decoded_ids.clear()
    texts_stop = mm.generate(prompts, None, max_gen_len=12, temperature=0.0,
                             additional_stop_symbols=["li", "ab"])
    ids_c = [list(t) for t in decoded_ids]
    mm.tokenizer.decode = orig_decode
    meta_json = {"prompts": prompts,
                 "prompt_ids": [mm.tokenizer.encode(p, bos=True, eos=False) for p in prompts],
                 "gen12_text": texts, "gen12_ids": ids_a,
                 "gen48_text": texts_trunc, "gen48_ids": ids_b,
                 "genstop_text": texts_stop, "genstop_ids": ids_c, "genstop_symbols": ["li", "ab"],
                 "vocab_size": V, "tiny": TINY,
                 "need_space_before_segment": mm.tokenizer.need_space_before_segment,
                 "bos": mm.tokenizer.bos_id, "eos": mm.tokenizer.eos_id}
    np.savez_compressed(os.path.join(GOLD, "meta_tiny.npz"), **mo)
    with open(os.path.join(GOLD, "meta_tiny.json"), "w") as f:
        json.dump(meta_json, f, indent=1)

    # ---------------- G7/G8 vision ----------------
    vo = {}
    B = 2
    vsd = ref_cpu.make_vision_weights(64, width=VIT["width"], layers=VIT["layers"], patch=VIT["patch"],
                                      grid=VIT["grid"], in_feat=VIT["width"] + 3072 + 1536,
                                      with_qformer=True, seed=1, std=0.05)
    vo["vis_weight_checksum"] = np.float64(checksum(vsd))
    vm = ens5.Transformer(args, with_visual=False)
    vm.load_state_dict(sd)
    vm.clip = _Clip(w=VIT["width"], layers=VIT["layers"], h=VIT["heads"], patch=VIT["patch"], grid=VIT["grid"])
    vm.qformer_proj = nn.Sequential(nn.Linear(768, 64), nn.LayerNorm(64))
    vm.visual_proj = nn.Sequential(nn.Linear(3072 + VIT["width"] + 1536, 64), nn.LayerNorm(64))
    vm.start_img = nn.Parameter(torch.zeros(1, 1, 64))
    vm.end_img = nn.Parameter(torch.zeros(1, 1, 64))
    vm.image_words = (32 + 257 + 2) * 5
    vm.image_size = 448
    res = vm.load_state_dict(vsd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    qf, cnx, dino = extra_feature_inputs(5 * B)
    vm.qformer = _QF(qf)
    vm.openclip_convnext_xxl = _FixedFeat(cnx)
    vm.dinov2_vitg14 = _Dino(dino)
    vm.eval()
    vo["vis_state_keys"] = np.array(sorted(k for k in vm.state_dict().keys()
                                           if k.startswith(("clip.", "visual_proj", "qformer_proj", "start_", "end_"))))
    img = synth_image(B)
    vo["image_seed"] = np.int64(5)
    vo["image_checksum"] = np.float64(img.double().abs().sum().item())
    args_v = ens5.ModelArgs(vocab_size=V, max_batch_size=32, **{**TINY, "max_seq_len": 1600})
    vm.args = args_v
    vm.freqs_cis = ref_cpu.precompute_freqs_cis(16, 3200)
    for l in vm.layers:
        l.attention.args = args_v
    with torch.no_grad():
        crops = ref_cpu.split_views(img, 224)
        vo["g7_clip_feats"] = np32(vm.clip_encode_image(crops[:3]))
        views = vm.encode_image(img)
        vo["g8_views"] = np.stack([np32(v) for v in views])
        exv = ex[:, :9]
        vo["g8_examples"] = exv.numpy()
        vo["g8_logits"] = np32(vm(exv, img))
        l0 = np32(vm.forward_inference(exv[:, :6], 0, img))
        l1 = np32(vm.forward_inference(exv[:, 6:7], 6))
        l2 = np32(vm.forward_inference(exv[:, 7:8], 7))
        vo["g8_inf_logits"] = np.stack([l0, l1, l2])
        vo["g8_cache_image_words"] = np.int64(vm.cache_image_words)
    np.savez_compressed(os.path.join(GOLD, "vision_tiny.npz"), **vo)
    print("fixtures written to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print(f"  {fn:28s} {os.path.getsize(os.path.join(GOLD, fn)) / 1024:.1f} KB")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def gen_host():
    """G10: host-side contracts captured from the reference: FinetuneDistSampler index lists, the LR schedule
    table and add_weight_decay grouping (tests/golden/host_tiny.json)."""
    refimport.install(data_stubs=True)
    import types
    import accessory.util.lr_sched as lr_sched
    import pandas  # noqa: F401  (imported by data/alpaca.py)
    from accessory.data.alpaca import FinetuneDistSampler
    from accessory.util import misc

    class DS:
        def __init__(self, sizes):
            self.g, o = [], 0
            for s in sizes:
                self.g.append(list(range(o, o + s)))
                o += s

        def groups(self):
            return self.g

        def __len__(self):
            return sum(len(x) for x in self.g)

    out = {"sampler": []}
    for (sizes, ws, bs, acc, seed) in [((200, 130), 8, 2, 2, 0), ((64, 48, 100), 2, 4, 1, 3), ((1000,), 8, 8, 1, 1)]:
        ds = DS(sizes)
        for rank in sorted({0, ws - 1, ws // 2}):
            for epoch, start_iter, shuffle in [(0, 0, True), (1, 3, True), (0, 0, False)]:
                s = FinetuneDistSampler(ds, num_replicas=ws, rank=rank, shuffle=shuffle, seed=seed, batch_size=bs, acc_grad=acc)
                s.set_epoch(epoch, start_iter)
                out["sampler"].append(dict(sizes=list(sizes), ws=ws, bs=bs, acc=acc, seed=seed, rank=rank, epoch=epoch,
                                           start_iter=start_iter, shuffle=shuffle, indices=list(iter(s)), length=len(s)))
    args = types.SimpleNamespace(lr=2e-5, min_lr=0.0, warmup_epochs=0.03, epochs=3)
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}])
    table = []
    for e in [0.0, 0.01, 0.03, 0.5, 1.0, 1.7, 2.999]:
        lr = lr_sched.adjust_learning_rate_epoch(opt, e, args)
        table.append([e, lr, opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]])
    out["lr_table"] = table
    m = nn.Module()
    m.a = nn.Linear(4, 4)
    m.attention_norm = nn.LayerNorm(4)
    m.frozen = nn.Linear(4, 4)
    for p in m.frozen.parameters():
        p.requires_grad = False
    groups = misc.add_weight_decay(m, 0.1)
    names = {id(p): n for n, p in m.named_parameters()}
    out["wd_groups"] = [dict(weight_decay=g["weight_decay"], names=sorted(names[id(p)] for p in g["params"])) for g in groups]
    with open(os.path.join(GOLD, "host_tiny.json"), "w") as f:
        json.dump(out, f)
    print("host fixture written:", os.path.getsize(os.path.join(GOLD, "host_tiny.json")), "bytes")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "host":
    gen_host()


def gen_ckpt():
    """G11: a 2-shard ``consolidated.*-of-02.model.pth`` checkpoint and a ``meta_ori`` one, merged to MP=1 by the
    REFERENCE loader (util/tensor_parallel.py) -> checksums of every merged tensor (tests/golden/ckpt_tiny.json)."""
    refimport.install()
    refimport.init_dist_ws1()
    import accessory.model.meta as meta
    from accessory.util import tensor_parallel as tp
    tok_path = os.path.join(GOLD, "tokenizer.model")
    mm = meta.MetaModel("llama_ens5", os.path.join(GOLD, "tiny_params.json"), tok_path, with_visual=False, max_seq_len=64)
    V = mm.tokenizer.n_words
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=21, std=0.05)
    full = {"llma." + k: v for k, v in sd.items()}
    from a3vlm_amd.checkpoint import merge_dim
    out = {}
    with tempfile.TemporaryDirectory() as d:
        # split with the reference's own dims (Column 0 / Row 1 / Embedding 1) through torch.chunk
        for fmt, names in (("consolidated", ["consolidated.00-of-02.model.pth", "consolidated.01-of-02.model.pth"]),
                           ("meta_ori", ["consolidated.00.pth", "consolidated.01.pth"])):
            sub = os.path.join(d, fmt)
            os.makedirs(sub)
            for r, fn in enumerate(names):
                shard = {}
                for k, v in full.items():
                    dd = merge_dim(k)
                    t = torch.chunk(v, 2, dd)[r].clone() if dd >= 0 else v.clone()
                    shard[k if fmt == "consolidated" else k[len("llma."):]] = t
                torch.save({"model": shard} if fmt == "consolidated" else shard, os.path.join(sub, fn))
            for p in mm.parameters():
                p.data.zero_()
            res = tp.load_tensor_parallel_model_list(mm, [sub])
            got = mm.state_dict()
            out[fmt] = {"load_result": res, "sums": {k: float(v.double().sum()) for k, v in got.items()},
                        "abs_sums": {k: float(v.double().abs().sum()) for k, v in got.items()}}
            assert all(torch.equal(got[k], full[k]) for k in full), "reference merge must reproduce the unsharded weights"
    out["seed"] = 21
    with open(os.path.join(GOLD, "ckpt_tiny.json"), "w") as f:
        json.dump(out, f)
    print("ckpt fixture written")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ckpt":
    gen_ckpt()


def gen_text():
    """G9/G12-lite: prompt strings of the reference's Conversation for demo.json, a multi-turn example with label spans,
    and format_bounding_box / normalize_number outputs (the two functions are exec'd from the reference's eval script,
    whose module-level imports -- gradio, cv2, torchvision -- are not installable here)."""
    import ast
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_conv_lib", os.path.join(refimport.REF_ROOT, "accessory/data/conversation/lib.py"))
    lib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lib)
    demo = json.load(open(os.path.join(refimport.REF_ROOT, "accessory/demo_data/demo.json")))
    out = {"demo_prompts": []}
    for item in demo:
        conv = lib.conv_v1_2()
        conv.load_qas([[item["conversations"][0]["value"], None]])
        out["demo_prompts"].append(conv.get_prompt())
    conv = lib.conv_v1_2()
    conv.load_qas([["Detect all manipulable object parts.", "<box>lid</box>[[0.12,0.34,0.56]]"], ["And the joint?", "<axis>revolute</axis>[0.10,0.20]"], ["Again?", None]])
    out["multi_turn"] = conv.process()
    out["response_end_signal"] = conv.response_end_signal
    src = open(os.path.join(refimport.REF_ROOT, "accessory/eval_affordance_v2.py")).read()
    tree = ast.parse(src)
    ns = {"re": __import__("re")}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("normalize_number", "format_bounding_box"):
            exec(compile(ast.Module([node], []), "ref_eval", "exec"), ns)
    cases = ["[0.12, 0.34, 0.56, 0.78]", "The box is [123, 456, 789, 999]", "[[0.15,0.20],[0.62,0.90]]", "0.5,12,345,0.0071", "no numbers",
             "<box>lid</box>[[12.5, 33.1, 0.9], [1234,5678,1,2]]"]
    out["bbox_cases"] = [{"in": c, "out": ns["format_bounding_box"](c)} for c in cases]
    with open(os.path.join(GOLD, "text_tiny.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("text fixture written")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "text":
    gen_text()


def dialog_yaml(tmpdir):
    """The dataset config used by the G9 fixture and its test (paths are machine-specific, so it is written on the fly)."""
    d = os.path.join(GOLD, "dialog")
    p = os.path.join(tmpdir, "dialog.yaml")
    with open(p, "w") as f:
        f.write("META:\n"
                f"  - path: '{d}/mm.json'\n    type: 'image_text'\n    root: '{os.path.join(GOLD, 'demo')}'\n"
                f"  - path: '{d}/text.jsonl'\n    type: 'text'\n    ratio: 0.9\n")
    return p


def dialog_transform(img):
    """Stand-in transform for the fixture: a tiny deterministic function of the decoded PIL image."""
    a = np.asarray(img.resize((4, 4)), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).contiguous()


def gen_dialog():
    """G9: items of the reference's FinetuneDialogDataset (tokens, assistant-span labels, masks, group layout) over
    tests/golden/dialog/* with the fixture tokenizer -> tests/golden/dialog/items.json."""
    import builtins
    import tempfile
    refimport.install(data_stubs=True)
    import pandas  # noqa: F401
    from accessory.data.conversation.dataset import FinetuneDialogDataset
    real_print = builtins.print
    builtins.print = lambda *a, force=False, **k: real_print(*a, **k)      # the reference prints with force= (misc.py patch)
    with tempfile.TemporaryDirectory() as td:
        ds = FinetuneDialogDataset(dialog_yaml(td), dialog_transform, max_words=150, image_words=30,
                                   tokenizer=os.path.join(GOLD, "tokenizer.model"))
        out = {"len": len(ds), "groups": ds.groups(), "items": []}
        for i in range(len(ds)):
            it = ds[i]
            rec = {"tokens": it[0].tolist(), "labels": it[1].tolist(), "mask_sum": float(it[2].sum())}
            if len(it) == 4:
                rec["image_sum"] = float(it[3].double().sum())
            out["items"].append(rec)
    builtins.print = real_print
    with open(os.path.join(GOLD, "dialog", "items.json"), "w") as f:
        json.dump(out, f)
    print("dialog fixture written:", out["len"], "items, groups", [len(g) for g in out["groups"]])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "dialog":
    gen_dialog()


def depth_tags(dim: int = 64, seed: int = 21):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(1, 1, dim, generator=g), torch.rand(1, 1, dim, generator=g)


def gen_2img():
    """G8b: the reference's two-image plugin (llama_ens5_2images): RGB + depth image words with their own tags,
    teacher-forced logits and cached prefill + 2 decode steps -> tests/golden/vision2_tiny.npz."""
    refimport.install()
    refimport.init_dist_ws1()
    torch.manual_seed(0)
    import sentencepiece as spm
    import accessory.model.LLM.llama_ens5_2images as ens2
    V = spm.SentencePieceProcessor(model_file=os.path.join(GOLD, "tokenizer.model")).vocab_size()
    args = ens2.ModelArgs(vocab_size=V, max_batch_size=32, **{**TINY, "max_seq_len": 3200})
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=V, **TINY), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=VIT["width"], layers=VIT["layers"], patch=VIT["patch"], grid=VIT["grid"],
                                      in_feat=VIT["width"] + 3072 + 1536, with_qformer=True, seed=1, std=0.05)
    vm = ens2.Transformer(args, with_visual=False)
    vm.load_state_dict(sd)
    vm.clip = _Clip(w=VIT["width"], layers=VIT["layers"], h=VIT["heads"], patch=VIT["patch"], grid=VIT["grid"])
    vm.qformer_proj = nn.Sequential(nn.Linear(768, 64), nn.LayerNorm(64))
    vm.visual_proj = nn.Sequential(nn.Linear(3072 + VIT["width"] + 1536, 64), nn.LayerNorm(64))
    vm.start_img = nn.Parameter(torch.zeros(1, 1, 64))
    vm.end_img = nn.Parameter(torch.zeros(1, 1, 64))
    sdi, edi = depth_tags()
    vm.start_depth_img, vm.end_depth_img = nn.Parameter(sdi), nn.Parameter(edi)
    vm.visual_image_words = (32 + 257 + 2) * 5
    vm.image_words = vm.visual_image_words * 2
    vm.image_size = 448
    res = vm.load_state_dict(vsd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    B = 2
    qf, cnx, dino = extra_feature_inputs(5 * B)       # the three out-of-scope streams: same features for both images
    vm.qformer, vm.openclip_convnext_xxl, vm.dinov2_vitg14 = _QF(qf), _FixedFeat(cnx), _Dino(dino)
    vm.eval()
    img, depth = synth_image(B, seed=5), synth_image(B, seed=6)
    g = torch.Generator().manual_seed(31)
    ex = torch.randint(3, V, (B, 9), generator=g)
    ex[:, 0] = 1
    vo = {"examples": ex.numpy(), "depth_checksum": np.float64(depth.double().abs().sum().item())}
    with torch.no_grad():
        full = vm(ex, img, depth)                  # reference quirk (:505): rows from visual_image_words on = depth words + text
        vo["logits_2img_shape"] = np.array(full.shape)
        vo["logits_2img_tail"] = np32(full[:, -12:])
        vo["logits_2img_rowsum"] = np32(full.sum(-1))
        vo["logits_rgb_only"] = np32(vm(ex, img))
        l0 = np32(vm.forward_inference(ex[:, :6], 0, img, depth))
        vo["cache_image_words"] = np.int64(vm.cache_image_words)
        l1 = np32(vm.forward_inference(ex[:, 6:7], 6))
        l2 = np32(vm.forward_inference(ex[:, 7:8], 7))
        vo["inf_logits"] = np.stack([l0, l1, l2])
        vo["inf_logits_rgb_dropped"] = np32(vm.forward_inference(ex[:, :6], 0, img, None))   # image ignored without depth (:517)
    np.savez_compressed(os.path.join(GOLD, "vision2_tiny.npz"), **vo)
    print("two-image fixture written", {k: getattr(v, "shape", None) for k, v in vo.items()})


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "2img":
    gen_2img()


def gen_lora():
    """G13: the reference's LoRA linears (model/peft.py LoraLinear / LoraColumnParallelLinear / LoraRowParallelLinear at
    MP=1) on seeded inputs -> tests/golden/lora_tiny.npz; pins y = W x (+ b) + B (A x) with no alpha/r scaling."""
    refimport.install(data_stubs=True)
    refimport.init_dist_ws1()
    import sys as _sys
    fs = _sys.modules["fairscale.nn.model_parallel.layers"]
    import accessory.model.peft as peft
    g = torch.Generator().manual_seed(41)
    out = {}
    for name, cls, kw in [("plain", peft.LoraLinear, dict(bias=True)), ("col", peft.LoraColumnParallelLinear, dict(bias=False, gather_output=False)),
                          ("row", peft.LoraRowParallelLinear, dict(bias=False, input_is_parallel=True))]:
        m = cls(48, 80, lora_rank=8, **kw)
        w = torch.randn(80, 48, generator=g) * 0.1
        a = torch.randn(8, 48, generator=g) * 0.1
        b = torch.randn(80, 8, generator=g) * 0.1
        x = torch.randn(5, 48, generator=g)
        with torch.no_grad():
            m.weight.copy_(w)
            if getattr(m, "bias", None) is not None:
                m.bias.zero_()
            assert float(m.lora_b.weight.abs().sum()) == 0.0          # reference init: B = 0
            m.lora_a.weight.copy_(a)
            m.lora_b.weight.copy_(b)
            y = m(x)
        out[name + "_w"], out[name + "_a"], out[name + "_b"], out[name + "_x"], out[name + "_y"] = np32(w), np32(a), np32(b), np32(x), np32(y)
        out[name + "_keys"] = np.array(sorted(m.state_dict().keys()))
    np.savez_compressed(os.path.join(GOLD, "lora_tiny.npz"), **out)
    print("lora fixture written", list(out["col_keys"]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "lora":
    gen_lora()
