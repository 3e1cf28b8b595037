"""CPU oracle for the A3VLM multimodal hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU) restatement of the reference algorithm for
the path BASELINE.json names (ViT patch-embed + encoder -> projector -> Llama
decoder over [BOS | image tokens | text] -> LM head / CE loss / greedy decode).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product package ``a3vlm_amd`` never does.

Every function cites the reference file:line it follows (paths relative to
``/root/reference/model/accessory``).  Weights are passed as a flat dict that
uses the reference's own state-dict key names (without the ``llma.`` prefix),
so the same dict drives the oracle and the HIP plugin.

Pinning status
--------------
* decoder / RMSNorm / attention / FFN / MetaModel.forward / generate(T=0):
  pinned -- ``oracle/gen_golden.py`` imports the reference itself (MP=1 stubs)
  and ``tests/test_oracle_golden.py`` checks this file against the committed
  outputs in ``tests/golden/``.
* RoPE helpers (``model/LLM/llama.py``) are ABSENT from the reference snapshot.
  They are restated here from the call sites (``LLM/llama_ens5.py:118,271-274``)
  and pinned independently through the reference's own HF exporter
  (``tools/convert_weights_to_hf.py:209-219``) + installed ``transformers``
  LlamaForCausalLM (fixture ``hf_crosscheck.npz``).  ``rope_scaling`` (position
  multiplier) has no such pin: parity unpinned for rope_scaling != None.
* CLIP ViT-L/14 internals live in third-party ``open_clip`` (un-pinned, not
  installed).  ``clip_encode_image`` (``LLM/llama_ens5.py:351-375``) is pinned by
  running the reference method against a stand-in exposing the open_clip
  attribute surface; the resblock internals (pre-LN, nn.MultiheadAttention
  packing ``in_proj_weight`` = [q;k;v], erf-GELU MLP) restate open_clip's
  published ``VisionTransformer``: parity unpinned at that boundary.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# ModelArgs  (LLM/llama_ens5.py:33-50)
# --------------------------------------------------------------------------
@dataclass
class OracleArgs:
    dim: int = 5120
    n_layers: int = 40
    n_heads: int = 40
    n_kv_heads: Optional[int] = None
    vocab_size: int = -1
    multiple_of: int = 256
    ffn_dim_multiplier: Optional[float] = None
    norm_eps: float = 1e-5
    rope_theta: float = 10000
    max_batch_size: int = 32
    max_seq_len: int = 2048
    rope_scaling: Optional[float] = None


def ffn_hidden_dim(dim: int, multiple_of: int, ffn_dim_multiplier: Optional[float]) -> int:
    """LLM/llama_ens5.py:196-200 (called with hidden_dim = 4*dim at :229)."""
    hidden = int(2 * (4 * dim) / 3)
    if ffn_dim_multiplier is not None:
        hidden = int(ffn_dim_multiplier * hidden)
    return multiple_of * ((hidden + multiple_of - 1) // multiple_of)


# --------------------------------------------------------------------------
# RoPE helpers -- restated (missing LLM/llama.py); call sites
# LLM/llama_ens5.py:118,152-153,271-274.  Interleaved-pair convention, pinned
# by tools/convert_weights_to_hf.py:209-219 (see module docstring).
# --------------------------------------------------------------------------
def precompute_freqs_cis(dim: int, end: int, theta: float = 10000.0,
                         scaling: Optional[float] = None) -> Tensor:
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
    t = torch.arange(end, dtype=torch.float32)
    if scaling is not None:
        t = t * scaling
    freqs = torch.outer(t, freqs).float()
    return torch.polar(torch.ones_like(freqs), freqs)  # complex64 [end, dim/2]


def reshape_for_broadcast(freqs_cis: Tensor, x: Tensor) -> Tensor:
    ndim = x.ndim
    assert freqs_cis.shape == (x.shape[1], x.shape[-1])
    shape = [d if i == 1 or i == ndim - 1 else 1 for i, d in enumerate(x.shape)]
    return freqs_cis.view(*shape)


def apply_rotary_emb(xq: Tensor, xk: Tensor, freqs_cis: Tensor):
    xq_ = torch.view_as_complex(xq.float().reshape(*xq.shape[:-1], -1, 2))
    xk_ = torch.view_as_complex(xk.float().reshape(*xk.shape[:-1], -1, 2))
    freqs_cis = reshape_for_broadcast(freqs_cis, xq_)
    xq_out = torch.view_as_real(xq_ * freqs_cis).flatten(3)
    xk_out = torch.view_as_real(xk_ * freqs_cis).flatten(3)
    return xq_out.type_as(xq), xk_out.type_as(xk)


def repeat_kv(x: Tensor, n_rep: int) -> Tensor:
    bs, slen, n_kv_heads, head_dim = x.shape
    if n_rep == 1:
        return x
    return (x[:, :, :, None, :]
            .expand(bs, slen, n_kv_heads, n_rep, head_dim)
            .reshape(bs, slen, n_kv_heads * n_rep, head_dim))


# --------------------------------------------------------------------------
# RMSNorm  (model/components.py:39,52-53)
# --------------------------------------------------------------------------
def rmsnorm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    xf = x.float()
    out = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x)
    return out * weight


def make_causal_mask(q_len: int, kv_len: int) -> Tensor:
    """Right-aligned boolean causal mask, LLM/llama_ens5.py:181-185."""
    q_idx = torch.arange(q_len) - q_len
    kv_idx = torch.arange(kv_len) - kv_len
    return q_idx.view(-1, 1) >= kv_idx.view(1, -1)


def sdpa(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor]) -> Tensor:
    """softmax(q k^T / sqrt(d) + mask) v -- what F.scaled_dot_product_attention
    computes at LLM/llama_ens5.py:164; written out so that the accumulation
    dtype is explicit (fp32) independent of the torch CPU kernel chosen."""
    scale = 1.0 / math.sqrt(q.shape[-1])
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v.float()).to(q.dtype)


# --------------------------------------------------------------------------
# Decoder
# --------------------------------------------------------------------------
class OracleDecoder:
    """Llama decoder with the reference's KV-cache semantics.

    ``sd`` keys: tok_embeddings.weight, layers.{i}.attention.{wq,wk,wv,wo}.weight,
    layers.{i}.feed_forward.{w1,w2,w3}.weight, layers.{i}.{attention_norm,ffn_norm}.weight,
    norm.weight, output.weight  (LLM/llama_ens5.py:258-269; tools/convert_weights_to_hf.py:101-115).
    """

    def __init__(self, args: OracleArgs, sd: Dict[str, Tensor]):
        self.args = args
        self.sd = sd
        self.n_kv_heads = args.n_heads if args.n_kv_heads is None else args.n_kv_heads
        self.n_rep = args.n_heads // self.n_kv_heads
        self.head_dim = args.dim // args.n_heads
        # LLM/llama_ens5.py:271-274
        self.freqs_cis = precompute_freqs_cis(self.head_dim, args.max_seq_len * 2,
                                              theta=args.rope_theta, scaling=args.rope_scaling)
        self.k_cache: List[Optional[Tensor]] = [None] * args.n_layers
        self.v_cache: List[Optional[Tensor]] = [None] * args.n_layers
        self.cache_image_words = 0

    # LLM/llama_ens5.py:97-169
    def lin(self, x: Tensor, name: str) -> Tensor:
        """F.linear with the optional LoRA branch of model/peft.py:84-99: y = W x + lora_b(lora_a(x)), no alpha/r scale."""
        y = F.linear(x, self.sd[name + ".weight"])
        if name + ".lora_a.weight" in self.sd:
            y = y + F.linear(F.linear(x, self.sd[name + ".lora_a.weight"]), self.sd[name + ".lora_b.weight"])
        return y

    def attention(self, i: int, x: Tensor, start_pos: int, freqs_cis: Tensor, mask) -> Tensor:
        sd, p = self.sd, f"layers.{i}.attention."
        bsz, seqlen, _ = x.shape
        xq = self.lin(x, p + "wq").view(bsz, seqlen, self.args.n_heads, self.head_dim)
        xk = self.lin(x, p + "wk").view(bsz, seqlen, self.n_kv_heads, self.head_dim)
        xv = self.lin(x, p + "wv").view(bsz, seqlen, self.n_kv_heads, self.head_dim)
        xq, xk = apply_rotary_emb(xq, xk, freqs_cis)
        if self.k_cache[i] is None:
            keys, values = xk, xv
        else:
            self.k_cache[i] = self.k_cache[i].to(xk)
            self.v_cache[i] = self.v_cache[i].to(xv)
            self.k_cache[i][:bsz, start_pos:start_pos + seqlen] = xk
            self.v_cache[i][:bsz, start_pos:start_pos + seqlen] = xv
            keys = self.k_cache[i][:bsz, :start_pos + seqlen]
            values = self.v_cache[i][:bsz, :start_pos + seqlen]
        keys = repeat_kv(keys, self.n_rep).transpose(1, 2)
        values = repeat_kv(values, self.n_rep).transpose(1, 2)
        xq = xq.transpose(1, 2)
        m = make_causal_mask(xq.size(2), keys.size(2)) if mask == "causal" else None
        out = sdpa(xq, keys, values, m).transpose(1, 2).contiguous().view(bsz, seqlen, -1)
        return self.lin(out, p + "wo")

    # LLM/llama_ens5.py:213-217
    def feed_forward(self, i: int, x: Tensor) -> Tensor:
        p = f"layers.{i}.feed_forward."
        return self.lin(F.silu(self.lin(x, p + "w1")) * self.lin(x, p + "w3"), p + "w2")

    # LLM/llama_ens5.py:237-249
    def block(self, i: int, x: Tensor, start_pos: int, freqs_cis: Tensor, mask) -> Tensor:
        eps = self.args.norm_eps
        h = x + self.attention(i, rmsnorm(x, self.sd[f"layers.{i}.attention_norm.weight"], eps),
                               start_pos, freqs_cis, mask)
        return h + self.feed_forward(i, rmsnorm(h, self.sd[f"layers.{i}.ffn_norm.weight"], eps))

    def embed(self, tokens: Tensor) -> Tensor:
        return F.embedding(tokens, self.sd["tok_embeddings.weight"])

    def allocate_kv_cache(self, bsz: int) -> None:
        """LLM/llama_ens5.py:171-176,533-535 -- contents are uninitialised in the
        reference; zeros here (never read before written)."""
        shape = (bsz, self.args.max_seq_len, self.n_kv_heads, self.head_dim)
        for i in range(self.args.n_layers):
            if self.k_cache[i] is None or tuple(self.k_cache[i].shape) != shape:
                self.k_cache[i] = torch.zeros(shape)
                self.v_cache[i] = torch.zeros(shape)

    def destroy_kv_cache(self) -> None:
        self.k_cache = [None] * self.args.n_layers
        self.v_cache = [None] * self.args.n_layers

    # LLM/llama_ens5.py:461-487
    def forward(self, examples: Tensor, image_tokens: Optional[Tensor] = None, out_from: Optional[int] = None) -> Tensor:
        """``out_from``: first sequence row fed to the LM head when it is not the number of image words (the
        two-image plugin slices from ``visual_image_words``, LLM/llama_ens5_2images.py:505)."""
        self.destroy_kv_cache()
        h = self.embed(examples)
        image_words = 0
        if image_tokens is not None:
            image_words = image_tokens.shape[1]
            h = torch.cat((h[:, :1], image_tokens.to(h.dtype), h[:, 1:]), dim=1)
        seqlen = h.shape[1]
        freqs_cis = self.freqs_cis[:seqlen]
        for i in range(self.args.n_layers):
            h = self.block(i, h, 0, freqs_cis, "causal")
        h = rmsnorm(h, self.sd["norm.weight"], self.args.norm_eps)
        return F.linear(h[:, image_words if out_from is None else out_from:, :], self.sd["output.weight"])

    # LLM/llama_ens5.py:490-531
    def forward_inference(self, tokens: Tensor, start_pos: int,
                          image_tokens: Optional[Tensor] = None) -> Tensor:
        bsz, seqlen = tokens.shape
        if start_pos == 0:
            self.allocate_kv_cache(bsz)
        h = self.embed(tokens)
        if image_tokens is not None:
            assert start_pos == 0
            self.cache_image_words = image_tokens.shape[1]
            h = torch.cat((h[:, :1], image_tokens.to(h.dtype), h[:, 1:]), dim=1).to(h)
            seqlen = h.shape[1]
            freqs_cis = self.freqs_cis[0:seqlen]
        else:
            if start_pos == 0:
                self.cache_image_words = 0
                freqs_cis = self.freqs_cis[0:seqlen]
            else:
                start_pos = start_pos + self.cache_image_words
                freqs_cis = self.freqs_cis[start_pos:start_pos + seqlen]
        mask = None if seqlen == 1 else "causal"
        for i in range(self.args.n_layers):
            h = self.block(i, h, start_pos, freqs_cis, mask)
        h = rmsnorm(h, self.sd["norm.weight"], self.args.norm_eps)
        return F.linear(h[:, -1, :], self.sd["output.weight"]).float()


# --------------------------------------------------------------------------
# CLIP ViT  (wrapper: LLM/llama_ens5.py:351-375; internals: open_clip, see docstring)
# key names: util/param_group.py:72-90
# --------------------------------------------------------------------------
def vit_resblock(x: Tensor, sd: Dict[str, Tensor], p: str, n_heads: int,
                 quick_gelu: bool = False) -> Tensor:
    """One open_clip ResidualAttentionBlock on NLD input (batch_first math; the
    reference permutes to LND at :365 only because nn.MultiheadAttention's
    default is sequence-first -- the arithmetic is identical)."""
    N, L, D = x.shape
    hd = D // n_heads
    y = F.layer_norm(x, (D,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
    qkv = F.linear(y, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(N, L, n_heads, hd).transpose(1, 2)
    k = k.view(N, L, n_heads, hd).transpose(1, 2)
    v = v.view(N, L, n_heads, hd).transpose(1, 2)
    a = sdpa(q, k, v, None).transpose(1, 2).reshape(N, L, D)
    x = x + F.linear(a, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
    y = F.layer_norm(x, (D,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
    y = F.linear(y, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
    y = y * torch.sigmoid(1.702 * y) if quick_gelu else F.gelu(y)
    return x + F.linear(y, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])


def clip_encode_image(x: Tensor, sd: Dict[str, Tensor], n_layers: int, n_heads: int,
                      patch: int = 14, quick_gelu: bool = False,
                      prefix: str = "clip.visual.") -> Tensor:
    """LLM/llama_ens5.py:351-375: conv1(k=s=patch, no bias) -> [cls | patches] + pos
    -> ln_pre -> resblocks -> ln_post on ALL tokens, no final proj."""
    p = prefix
    x = F.conv2d(x, sd[p + "conv1.weight"], None, stride=patch)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd[p + "class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([cls, x], dim=1)
    x = x + sd[p + "positional_embedding"].to(x.dtype)
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"], 1e-5)
    for j in range(n_layers):
        x = vit_resblock(x, sd, f"{p}transformer.resblocks.{j}.", n_heads, quick_gelu)
    return F.layer_norm(x, (D,), sd[p + "ln_post.weight"], sd[p + "ln_post.bias"], 1e-5)


def split_views(image: Tensor, crop: int) -> Tensor:
    """LLM/llama_ens5.py:383-385: fp16 bicubic global view + 4 quadrant crops,
    concatenated along batch ([global; TL; TR; BL; BR])."""
    g = F.interpolate(image.half(), size=(crop, crop), mode="bicubic").to(image)
    parts = [image[..., :crop, :crop], image[..., :crop, crop:],
             image[..., crop:, :crop], image[..., crop:, crop:]]
    return torch.cat([g] + parts, dim=0)


def linear_ln(x: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    """nn.Sequential(Linear, LayerNorm), LLM/llama_ens5.py:325-333."""
    y = F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])
    return F.layer_norm(y, (y.shape[-1],), sd[p + "1.weight"], sd[p + "1.bias"], 1e-5)


def encode_image(image: Tensor, sd: Dict[str, Tensor], *, vit_layers: int, vit_heads: int,
                 n_views: int = 5, patch: int = 14, quick_gelu: bool = False,
                 qformer_feats: Optional[Tensor] = None,
                 extra_feats: Sequence[Tensor] = ()) -> List[Tensor]:
    """LLM/llama_ens5.py:377-458.

    ``n_views=5``: reference geometry (image is 2*crop square -> 5 crops).
    ``n_views=1``: single-crop geometry (image used as is).
    ``qformer_feats`` [n_views*B, 32, 768] and ``extra_feats`` (each
    [n_views*B, 257, C], concatenated after the CLIP features in the order
    given: convnext, dinov2 at :436-440) stand for the OUT-OF-SCOPE frozen
    encoders; they are inputs, not computed here.
    Returns a list of n_views tensors [B, (32+)T, dim].
    """
    if n_views == 5:
        views = split_views(image, image.shape[-1] // 2)
    else:
        assert n_views == 1
        views = image
    feats = clip_encode_image(views, sd, vit_layers, vit_heads, patch, quick_gelu)
    if len(extra_feats):
        feats = torch.cat([feats] + [e.to(feats) for e in extra_feats], dim=2)
    feats = linear_ln(feats, sd, "visual_proj.")
    if qformer_feats is not None:
        qf = linear_ln(qformer_feats.to(feats), sd, "qformer_proj.")
        feats = torch.cat([qf, feats], dim=1)
    return list(torch.chunk(feats, n_views))


def ensemble_extra_feats(views: Tensor, convnext_trunk, dinov2_net) -> List[Tensor]:
    """LLM/llama_ens5.py:402-434: the two extra streams of the reference's ensemble from the per-view pixels, given the frozen nets
    (any modules with the trunk / ``forward_features`` contracts -- the real ones are third-party and out of scope).
    ConvNeXt: fp16 round trip of the input, F.interpolate to 256x256 (nearest, the default), trunk -> [N, C, 8, 8], 2x repeat to
    16x16, flatten to 256 tokens, their mean prepended as "cls".  DINOv2: CLIP-normalised pixels re-normalised with the ImageNet
    mean / std, tokens = [x_norm_clstoken | x_norm_patchtokens]."""
    with torch.no_grad():
        cf = convnext_trunk(F.interpolate(views.half(), size=(256, 256)).to(views))
        assert cf.shape[2:] == (8, 8)
        cf = cf.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2)
        cf = cf.flatten(-2).permute(0, 2, 1)
        cf = torch.cat([cf.mean(dim=1, keepdim=True), cf], dim=1)
        clip_mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).to(views).view(3, 1, 1)
        clip_std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).to(views).view(3, 1, 1)
        d_mean = torch.tensor([0.485, 0.456, 0.406]).to(views).view(3, 1, 1)
        d_std = torch.tensor([0.229, 0.224, 0.225]).to(views).view(3, 1, 1)
        df = dinov2_net.forward_features((views * clip_std + clip_mean - d_mean) / d_std)
        df = torch.cat([df["x_norm_clstoken"].unsqueeze(1), df["x_norm_patchtokens"]], dim=1)
    return [cf, df]


def assemble_image_tokens(views: List[Tensor], start_img: Tensor, end_img: Tensor) -> Tensor:
    """LLM/llama_ens5.py:471-476: per view cat(start_img, tokens, end_img); cat views."""
    bsz = views[0].shape[0]
    out = []
    for v in views:
        out.append(torch.cat((start_img.expand(bsz, -1, -1).to(v), v,
                              end_img.expand(bsz, -1, -1).to(v)), dim=1))
    return torch.cat(out, dim=1)


# --------------------------------------------------------------------------
# MetaModel.forward (loss) and generate (greedy)  -- model/meta.py
# --------------------------------------------------------------------------
def trim_to_last_label(examples: Tensor, labels: Tensor):
    """model/meta.py:235-249."""
    non_zero = torch.count_nonzero(labels, dim=0)
    pos = non_zero.shape[0] - 1
    while pos >= 0 and non_zero[pos] == 0:
        pos -= 1
    if pos == -1:
        pos = 2
    return examples[:, :pos + 1], labels[:, :pos + 1]


def meta_forward_loss(dec: OracleDecoder, examples: Tensor, labels: Tensor,
                      image_tokens: Optional[Tensor] = None) -> Tensor:
    """model/meta.py:234-263 with CrossEntropyLoss(ignore_index=0) (:67)."""
    examples, labels = trim_to_last_label(examples, labels)
    output = dec.forward(examples, image_tokens)
    output = output[:, :-1, :]
    labels = labels[:, 1:]
    if labels.sum() == 0:
        return output.mean() * 0
    V = output.shape[-1]
    return F.cross_entropy(output.reshape(-1, V).float(), labels.flatten(), ignore_index=0)


def top_p_nucleus(probs: torch.Tensor, p: float):
    """model/meta.py:569-573 (sample_top_p up to the draw): probabilities sorted descending, everything whose PRECEDING mass
    exceeds p zeroed, the rest renormalised.  Returns (renormalised sorted probabilities, their token ids).  Pinned against the
    nucleus sets the reference produced (tests/golden/sampling.json)."""
    probs_sort, probs_idx = torch.sort(probs, dim=-1, descending=True, stable=True)
    probs_sum = torch.cumsum(probs_sort, dim=-1)
    mask = probs_sum - probs_sort > p
    probs_sort = probs_sort.masked_fill(mask, 0.0)
    probs_sort = probs_sort / probs_sort.sum(dim=-1, keepdim=True)
    return probs_sort, probs_idx


def sample_top_p_at(probs: torch.Tensor, p: float, u: torch.Tensor) -> torch.Tensor:
    """model/meta.py:568-583 with the one random ingredient made explicit: ``torch.multinomial(probs_sort, 1)`` (:576) draws from
    the renormalised nucleus; here the draw is the inverse CDF of that same distribution at the given uniforms u [B] (first sorted
    position whose cumulative mass exceeds u), so a device sampler fed the same u must return the same ids.  The reference's own
    ids depend on torch's generator stream (different on CPU and GPU), which is not part of the contract; its nucleus is."""
    ps, idx = top_p_nucleus(probs, p)
    n_kept = (ps > 0).sum(-1)
    cdf = torch.cumsum(ps.double(), dim=-1)
    pos = (cdf <= u.double().reshape(-1, 1)).sum(-1)
    pos = torch.minimum(pos, n_kept - 1)
    return idx.gather(-1, pos.unsqueeze(-1)).squeeze(-1)


def generate_greedy(dec: OracleDecoder, prompt_tokens: List[List[int]], *, image_tokens=None,
                    image_words: int = 0, max_gen_len: int = 512, eos_id: int = 2,
                    extra_stop: Sequence[Sequence[int]] = (), sampler=None):
    """model/meta.py:413-485 with temperature == 0 (argmax branch :459-460); ``sampler(step, logits) -> ids [B]`` replaces the
    argmax by the sampled branch (:456-458) with whatever draw the caller defines (``sample_top_p_at`` with recorded uniforms, or
    ids recorded from another run = teacher forcing), ``step`` counting from 0 at the first generated position.

    Works on token ids (tokenizer encode/decode stay in the caller).  Returns
    (tokens [B,total_len] int64, per-row generated id lists) -- the id lists are
    exactly what the reference passes to tokenizer.decode at :482-484.
    """
    bsz = len(prompt_tokens)
    args = dec.args
    assert bsz <= args.max_batch_size
    min_prompt = min(len(t) for t in prompt_tokens)
    max_prompt = max(len(t) for t in prompt_tokens)
    max_seq_len = args.max_seq_len - (image_words if image_tokens is not None else 0)
    total_len = min(max_seq_len, max_gen_len + max_prompt)
    prompt_tokens = [t[-(max_seq_len - max_gen_len):] for t in prompt_tokens]
    tokens = torch.zeros((bsz, total_len), dtype=torch.long)
    text_mask = torch.zeros((bsz, total_len), dtype=torch.bool)
    for k, t in enumerate(prompt_tokens):
        tokens[k, :len(t)] = torch.tensor(t).long()
        text_mask[k, :len(t)] = True
    start_pos, prev_pos = min_prompt, 0
    l_stop = [torch.tensor([eos_id])] + [torch.tensor(list(s)) for s in extra_stop]
    stopped = torch.zeros(bsz, dtype=torch.bool)
    stop_pos = torch.full((bsz,), start_pos + 1)
    for cur_pos in range(start_pos, total_len):
        logits = dec.forward_inference(tokens[:, prev_pos:cur_pos], prev_pos,
                                       image_tokens if prev_pos == 0 else None).float()
        if sampler is None:
            next_token = torch.argmax(logits, dim=-1).reshape(-1)
        else:
            next_token = sampler(cur_pos - start_pos, logits).reshape(-1)
        next_token = torch.where(text_mask[:, cur_pos], tokens[:, cur_pos], next_token)
        tokens[:, cur_pos] = next_token
        stop_pos = torch.where(stopped, stop_pos, cur_pos + 1)
        for st in l_stop:
            if cur_pos + 1 - len(st) >= 0:
                c1 = (tokens[:, cur_pos + 1 - len(st):cur_pos + 1] == st.unsqueeze(0)).all(dim=-1)
                c2 = ~text_mask[:, cur_pos]
                new = c1 * c2 * (~stopped)
                stop_pos = torch.where(new, cur_pos + 1 - len(st), stop_pos)
                stopped = torch.logical_or(new, stopped)
        if stopped.all():
            break
        prev_pos = cur_pos
    outs = [t[len(prompt_tokens[i]):stop_pos[i].item()] for i, t in enumerate(tokens.tolist())]
    return tokens, outs


# --------------------------------------------------------------------------
# Synthetic weights shared by tests / bench
# --------------------------------------------------------------------------
def make_decoder_weights(args: OracleArgs, seed: int = 0, std: float = 0.02,
                         dtype=torch.float32) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    hd = args.dim // args.n_heads
    n_kv = args.n_heads if args.n_kv_heads is None else args.n_kv_heads
    ffn = ffn_hidden_dim(args.dim, args.multiple_of, args.ffn_dim_multiplier)

    def rn(*s):
        return (torch.randn(*s, generator=g) * std).to(dtype)

    sd = {"tok_embeddings.weight": rn(args.vocab_size, args.dim)}
    for i in range(args.n_layers):
        p = f"layers.{i}."
        sd[p + "attention.wq.weight"] = rn(args.n_heads * hd, args.dim)
        sd[p + "attention.wk.weight"] = rn(n_kv * hd, args.dim)
        sd[p + "attention.wv.weight"] = rn(n_kv * hd, args.dim)
        sd[p + "attention.wo.weight"] = rn(args.dim, args.n_heads * hd)
        sd[p + "feed_forward.w1.weight"] = rn(ffn, args.dim)
        sd[p + "feed_forward.w2.weight"] = rn(args.dim, ffn)
        sd[p + "feed_forward.w3.weight"] = rn(ffn, args.dim)
        sd[p + "attention_norm.weight"] = (1.0 + 0.1 * torch.randn(args.dim, generator=g)).to(dtype)
        sd[p + "ffn_norm.weight"] = (1.0 + 0.1 * torch.randn(args.dim, generator=g)).to(dtype)
    sd["norm.weight"] = (1.0 + 0.1 * torch.randn(args.dim, generator=g)).to(dtype)
    sd["output.weight"] = rn(args.vocab_size, args.dim)
    return sd


def make_vision_weights(dim: int, *, width: int, layers: int, patch: int, grid: int,
                        in_feat: Optional[int] = None, with_qformer: bool = False,
                        seed: int = 1, std: float = 0.02, dtype=torch.float32) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)

    def rn(*s, sc=std):
        return (torch.randn(*s, generator=g) * sc).to(dtype)

    def ln(n):
        return (1.0 + 0.1 * torch.randn(n, generator=g)).to(dtype), rn(n, sc=0.1)

    p = "clip.visual."
    sd = {p + "conv1.weight": rn(width, 3, patch, patch),
          p + "class_embedding": rn(width, sc=width ** -0.5),
          p + "positional_embedding": rn(grid * grid + 1, width, sc=width ** -0.5)}
    sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"] = ln(width)
    sd[p + "ln_post.weight"], sd[p + "ln_post.bias"] = ln(width)
    for j in range(layers):
        q = f"{p}transformer.resblocks.{j}."
        sd[q + "ln_1.weight"], sd[q + "ln_1.bias"] = ln(width)
        sd[q + "ln_2.weight"], sd[q + "ln_2.bias"] = ln(width)
        sd[q + "attn.in_proj_weight"] = rn(3 * width, width)
        sd[q + "attn.in_proj_bias"] = rn(3 * width)
        sd[q + "attn.out_proj.weight"] = rn(width, width)
        sd[q + "attn.out_proj.bias"] = rn(width)
        sd[q + "mlp.c_fc.weight"] = rn(4 * width, width)
        sd[q + "mlp.c_fc.bias"] = rn(4 * width)
        sd[q + "mlp.c_proj.weight"] = rn(width, 4 * width)
        sd[q + "mlp.c_proj.bias"] = rn(width)
    in_feat = width if in_feat is None else in_feat
    sd["visual_proj.0.weight"] = rn(dim, in_feat)
    sd["visual_proj.0.bias"] = rn(dim)
    sd["visual_proj.1.weight"], sd["visual_proj.1.bias"] = ln(dim)
    if with_qformer:
        sd["qformer_proj.0.weight"] = rn(dim, 768)
        sd["qformer_proj.0.bias"] = rn(dim)
        sd["qformer_proj.1.weight"], sd["qformer_proj.1.bias"] = ln(dim)
    sd["start_img"] = torch.rand(1, 1, dim, generator=g).to(dtype)
    sd["end_img"] = torch.rand(1, 1, dim, generator=g).to(dtype)
    return sd


def make_lora_weights(args, rank: int, seed: int = 0, std_a: float = 0.02, std_b: float = 0.02) -> Dict[str, Tensor]:
    """LoRA adapters for the seven decoder linears of every layer (model/peft.py key names ``<linear>.lora_a.weight`` [r, in],
    ``<linear>.lora_b.weight`` [out, r]).  The reference initialises lora_b to zero (:76); tests use a non-zero B so the
    branch is exercised."""
    g = torch.Generator().manual_seed(seed)
    hd = args.dim // args.n_heads
    n_kv = args.n_heads if args.n_kv_heads is None else args.n_kv_heads
    ffn = ffn_hidden_dim(args.dim, args.multiple_of, args.ffn_dim_multiplier)
    shapes = {"attention.wq": (args.n_heads * hd, args.dim), "attention.wk": (n_kv * hd, args.dim), "attention.wv": (n_kv * hd, args.dim),
              "attention.wo": (args.dim, args.n_heads * hd), "feed_forward.w1": (ffn, args.dim), "feed_forward.w2": (args.dim, ffn),
              "feed_forward.w3": (ffn, args.dim)}
    sd = {}
    for i in range(args.n_layers):
        for name, (o, k) in shapes.items():
            sd[f"layers.{i}.{name}.lora_a.weight"] = torch.randn(rank, k, generator=g) * std_a
            sd[f"layers.{i}.{name}.lora_b.weight"] = torch.randn(o, rank, generator=g) * std_b
    return sd

