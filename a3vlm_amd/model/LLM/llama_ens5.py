"""MI355X-native model plugin with the interface of the reference's
``accessory.model.LLM.llama_ens5`` (reference: model/accessory/model/LLM/llama_ens5.py).

Drop-in contract (what ``MetaModel`` reads): module-level ``ModelArgs``
and ``Transformer``; ``Transformer(args, with_visual)`` is an ``nn.Module`` exposing
``.args .image_words .image_size .layers``, ``forward(examples, image) -> logits[B,T,V]``,
``forward_inference(tokens, start_pos, image) -> fp32 [B,V]`` (stateful KV cache keyed on
``start_pos == 0``) and ``get_trainable_params()``; parameter names are the reference's
checkpoint keys (``tok_embeddings.weight``, ``layers.{i}.attention.wq.weight`` ...,
``clip.visual.*``, ``visual_proj.{0,1}.*``, ``start_img``, ``end_img``).

Nothing here computes with torch ops: every arithmetic step is a kernel of
``liba3vlm_hip.so`` (``a3vlm_amd.ops``).  torch owns memory, streams and the module tree.

Differences from the reference that are deliberate and documented in DESIGN.md:
  * only the CLIP ViT branch of the 4-encoder ensemble is native; the Q-Former / ConvNeXt /
    DINOv2 streams (OUT OF SCOPE frozen third-party nets) enter through ``extra_feats`` /
    ``qformer_feats`` hooks and default to absent (``ModelArgs.extra_feat_dim = 0``,
    ``qformer_tokens = 0``);
  * the KV cache is stored as K [B,Hkv,S,hd] and V^T [B,Hkv,hd,S] (kernel-friendly), not
    [B,S,Hkv,hd]; it is private to the plugin in the reference too (llama_ens5.py:171-179).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch
from torch import nn

from ... import lib as _lib
from ... import ops


@dataclass
class ModelArgs:
    # --- identical to the reference (llama_ens5.py:33-50) ---
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
    load_pretrained_visual_encoder: bool = False
    # --- vision geometry (reference values are hard-coded at llama_ens5.py:297,335-336,383) ---
    vit_width: int = 1024
    vit_layers: int = 24
    vit_heads: int = 16
    vit_patch: int = 14
    vit_crop: int = 224          # side of one ViT input view
    n_views: int = 5             # 5 = global + 4 quadrants of a (2*crop)^2 image; 1 = single crop
    vit_quick_gelu: bool = False
    extra_feat_dim: int = 0      # reference: 3072 (ConvNeXt-XXL) + 1536 (DINOv2-g), via hooks
    qformer_tokens: int = 0      # reference: 32 (BLIP-2 Q-Former), via hook


def _ffn_hidden(dim: int, multiple_of: int, mult: Optional[float]) -> int:
    hidden = int(2 * (4 * dim) / 3)   # llama_ens5.py:196-200,229
    if mult is not None:
        hidden = int(mult * hidden)
    return multiple_of * ((hidden + multiple_of - 1) // multiple_of)


def precompute_cos_sin(head_dim: int, end: int, theta: float, scaling: Optional[float]) -> torch.Tensor:
    """fp32 [end, head_dim/2, 2] = (cos, sin) of the reference's complex table
    (``precompute_freqs_cis`` of the missing llama.py; call site llama_ens5.py:271-274).
    Built on the host with the same torch CPU ops the reference uses so the table is
    bit-identical to ``torch.polar``'s real/imag parts."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2, device="cpu")[: head_dim // 2].float() / head_dim))
    t = torch.arange(end, dtype=torch.float32, device="cpu")
    if scaling is not None:
        t = t * scaling
    ang = torch.outer(t, freqs).float()
    fc = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([fc.real, fc.imag], dim=-1).contiguous()


class _W(nn.Module):
    """Holder of a ``weight`` (and optional ``bias``) so state-dict keys match the reference."""

    def __init__(self, *shape, bias: bool = False, init: str = "linear"):
        super().__init__()
        w = torch.empty(*shape)
        if init == "linear":      # default_linear_init, llama_ens5.py:28
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        elif init == "ones":
            nn.init.ones_(w)
        else:
            nn.init.normal_(w, std=0.02)
        self.weight = nn.Parameter(w)
        if bias:
            self.bias = nn.Parameter(torch.zeros(shape[0]))


class Attention(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        n_kv = args.n_heads if args.n_kv_heads is None else args.n_kv_heads
        hd = args.dim // args.n_heads
        self.wq = _W(args.n_heads * hd, args.dim)
        self.wk = _W(n_kv * hd, args.dim)
        self.wv = _W(n_kv * hd, args.dim)
        self.wo = _W(args.dim, args.n_heads * hd)


class FeedForward(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.w1 = _W(hidden, dim)
        self.w2 = _W(dim, hidden)
        self.w3 = _W(hidden, dim)


class TransformerBlock(nn.Module):
    def __init__(self, layer_id: int, args: ModelArgs):
        super().__init__()
        self.layer_id = layer_id
        self.attention = Attention(args)
        self.feed_forward = FeedForward(args.dim, _ffn_hidden(args.dim, args.multiple_of, args.ffn_dim_multiplier))
        self.attention_norm = _W(args.dim, init="ones")
        self.ffn_norm = _W(args.dim, init="ones")


class _ResBlock(nn.Module):
    """Parameter tree of one open_clip ResidualAttentionBlock (key names: util/param_group.py:80-88)."""

    def __init__(self, w: int):
        super().__init__()
        self.ln_1 = _LN(w)
        self.attn = _MHA(w)
        self.ln_2 = _LN(w)
        self.mlp = nn.Module()
        self.mlp.c_fc = _W(4 * w, w, bias=True, init="normal")
        self.mlp.c_proj = _W(w, 4 * w, bias=True, init="normal")


class _LN(nn.Module):
    def __init__(self, w: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(w))
        self.bias = nn.Parameter(torch.zeros(w))


class _MHA(nn.Module):
    def __init__(self, w: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * w, w) * 0.02)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * w))
        self.out_proj = _W(w, w, bias=True, init="normal")


class _Visual(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        w, p = args.vit_width, args.vit_patch
        g = args.vit_crop // p
        self.conv1 = _W(w, 3, p, p, init="normal")
        self.class_embedding = nn.Parameter(torch.randn(w) * w ** -0.5)
        self.positional_embedding = nn.Parameter(torch.randn(g * g + 1, w) * w ** -0.5)
        self.ln_pre = _LN(w)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList([_ResBlock(w) for _ in range(args.vit_layers)])
        self.ln_post = _LN(w)


class _Clip(nn.Module):
    def __init__(self, args: ModelArgs):
        super().__init__()
        self.visual = _Visual(args)


class _Seq(nn.Module):
    """nn.Sequential(Linear, LayerNorm) parameter tree: keys ``0.weight 0.bias 1.weight 1.bias``."""

    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.add_module("0", _W(fout, fin, bias=True))
        self.add_module("1", _LN(fout))


class Transformer(nn.Module):
    def __init__(self, args: ModelArgs, with_visual: bool = False):
        super().__init__()
        self.args = args
        self._fuse_qkv_rope = True      # prefill: RoPE + KV-cache write in the qkv GEMM epilogue (False: separate kernel; test hook)
        self.vocab_size = args.vocab_size
        self.n_layers = args.n_layers
        self.n_heads = args.n_heads
        self.n_kv_heads = args.n_heads if args.n_kv_heads is None else args.n_kv_heads
        self.head_dim = args.dim // args.n_heads
        self.ffn = _ffn_hidden(args.dim, args.multiple_of, args.ffn_dim_multiplier)
        self.tok_embeddings = _W(args.vocab_size, args.dim)
        self.layers = nn.ModuleList([TransformerBlock(i, args) for i in range(args.n_layers)])
        self.norm = _W(args.dim, init="ones")
        self.output = _W(args.vocab_size, args.dim)
        # llama_ens5.py:271-274
        self._cos_sin_cpu = precompute_cos_sin(self.head_dim, args.max_seq_len * 2, args.rope_theta, args.rope_scaling)
        self._cos_sin = None

        self.with_visual = with_visual
        self.image_words = 0
        self.cache_image_words = 0
        self.image_size = 224
        if with_visual:
            g = args.vit_crop // args.vit_patch
            self.clip = _Clip(args)
            self.visual_proj = _Seq(args.vit_width + args.extra_feat_dim, args.dim)
            if args.qformer_tokens:
                self.qformer_proj = _Seq(768, args.dim)
            self.image_words = (args.qformer_tokens + g * g + 1 + 2) * args.n_views
            self.image_size = args.vit_crop * (2 if args.n_views == 5 else 1)
            self.start_img = nn.Parameter(torch.rand(1, 1, args.dim))   # llama_ens5.py:338-339
            self.end_img = nn.Parameter(torch.rand(1, 1, args.dim))
        # hooks for the OUT-OF-SCOPE frozen encoders: callables views[N,3,c,c] -> features
        self.qformer_fn: Optional[Callable] = None          # -> [N, 32, 768]
        self.extra_feat_fns: List[Callable] = []            # each -> [N, L, C_i]

        self._packed: Dict[str, torch.Tensor] = {}
        self._packed_version = None
        self._k_cache: List[torch.Tensor] = []
        self._vt_cache: List[torch.Tensor] = []
        self._cache_shape = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._row_maps: Dict[tuple, tuple] = {}

    # ------------------------------------------------------------------ plugin API
    def get_trainable_params(self, pretrain_stage: bool = False):
        """llama_ens5.py:342-349: everything except the frozen visual encoders."""
        no_train = ["qformer.", "openclip_convnext_xxl.", "clip.", "dinov2_vitg14."]
        return {n: p for n, p in self.named_parameters() if not any(n.startswith(x) for x in no_train)}

    def get_quant_blocklist(self) -> List[str]:
        pre = ["clip.", "openclip_convnext_xxl.", "dinov2_vitg14.", "qformer.", "visual_proj.", "qformer_proj."]
        return [n for n, _ in self.named_modules() if any(n.startswith(x) for x in pre)]

    # ------------------------------------------------------------------ helpers
    @property
    def _dtype(self) -> torch.dtype:
        return self.norm.weight.dtype

    @property
    def _device(self) -> torch.device:
        return self.norm.weight.device

    def _buf(self, name: str, shape, dtype=None) -> torch.Tensor:
        """Workspace cache: one allocation per (name, shape, dtype); re-used across calls so the
        steady state allocates nothing (graph-capturable, no allocator traffic in the hot loop)."""
        dtype = dtype or self._dtype
        key = (name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = torch.empty(*shape, dtype=dtype, device=self._device)
            self._ws[key] = t
        return t

    def invalidate_packed_weights(self) -> None:
        """Call after parameters change (optimizer step / load_state_dict)."""
        self._packed_version = None

    def _weights_version(self):
        from ...util import param_state_key
        return tuple(param_state_key(p) for p in self.parameters()) + (self._dtype, str(self._device))

    def _pack(self, check: bool = False) -> Dict[str, torch.Tensor]:
        """Kernel-side weight images (built once per weight version):
        wqkv.{i} = [wq; wk; wv] rows (one fused QKV GEMM); w13.{i} = w1/w3 interleaved in
        16-row blocks (SwiGLU fused in the GEMM epilogue); conv1 as [width, Kpad] zero padded.
        ``check`` re-validates against the parameters' version counters (done once per
        forward / prefill, not per decode step)."""
        if self._packed_version is not None and not check:
            return self._packed
        ver = self._weights_version()
        if self._packed_version == ver:
            return self._packed
        dtp = self._dtype
        pk: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for i, lyr in enumerate(self.layers):
                a, f = lyr.attention, lyr.feed_forward
                pk[f"wqkv.{i}"] = torch.cat([a.wq.weight, a.wk.weight, a.wv.weight], dim=0).to(dtp).contiguous()
                w1, w3 = f.w1.weight, f.w3.weight
                nb = w1.shape[0] // 16
                pk[f"w13.{i}"] = torch.stack([w1.view(nb, 16, -1), w3.view(nb, 16, -1)], dim=1).reshape(2 * w1.shape[0], -1).to(dtp).contiguous()
            if self.with_visual:
                pk.update(self._pack_vision())
        self._packed = pk
        self._packed_version = ver
        return pk

    def _pack_vision(self) -> Dict[str, torch.Tensor]:
        """conv1 as a [width, Kpad] GEMM weight (zero padded to the GEMM's K granule) in the ViT's own dtype;
        visual_proj.0 padded likewise in the model dtype."""
        pk: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            cw = self.clip.visual.conv1.weight
            vdt = cw.dtype
            K = cw[0].numel()
            Kpad = (K + 63) // 64 * 64
            w2d = torch.zeros(cw.shape[0], Kpad, dtype=vdt, device=cw.device)
            w2d[:, :K] = cw.reshape(cw.shape[0], -1)
            pk["conv1"] = w2d
            pw = getattr(self.visual_proj, "0").weight
            dtp = self._dtype
            Kp = pw.shape[1]
            Kp_pad = (Kp + 63) // 64 * 64
            if Kp_pad != Kp:
                w2 = torch.zeros(pw.shape[0], Kp_pad, dtype=dtp, device=pw.device)
                w2[:, :Kp] = pw.to(dtp)
                pk["visual_proj"] = w2
            else:
                pk["visual_proj"] = pw.to(dtp).contiguous()
        return pk

    def _vision_images(self) -> Dict[str, torch.Tensor]:
        """Vision-only weight images (used by the training engine, which keeps its own decoder images)."""
        key = (self.clip.visual.conv1.weight._version, self.clip.visual.conv1.weight.dtype, str(self._device))
        if getattr(self, "_vis_pack_key", None) != key:
            self._vis_pack = self._pack_vision()
            self._vis_pack_key = key
        return self._vis_pack

    def _cos_sin_dev(self) -> torch.Tensor:
        if self._cos_sin is None or self._cos_sin.device != self._device:
            self._cos_sin = self._cos_sin_cpu.to(self._device)
        return self._cos_sin

    # ------------------------------------------------------------------ KV cache
    def _allocate_kv_cache(self, bsz: int) -> None:
        """llama_ens5.py:171-176,533-535 (re-allocated only when the shape changes)."""
        smax = (self.args.max_seq_len + 63) // 64 * 64
        shape = (bsz, self.n_kv_heads, smax, self.head_dim, self._dtype, str(self._device))
        if self._cache_shape == shape:
            return
        dev, dtp = self._device, self._dtype
        self._k_cache = [torch.zeros(bsz, self.n_kv_heads, smax, self.head_dim, dtype=dtp, device=dev)
                         for _ in range(self.n_layers)]
        self._vt_cache = [torch.zeros(bsz, self.n_kv_heads, self.head_dim, smax, dtype=dtp, device=dev)
                          for _ in range(self.n_layers)]
        self._cache_shape = shape

    def _destroy_kv_cache(self) -> None:
        self._k_cache, self._vt_cache, self._cache_shape = [], [], None

    # ------------------------------------------------------------------ linear dispatch
    def _skinny_ws(self, M: int, N: int, K: int) -> torch.Tensor:
        """One zero-initialised GEMV workspace (grown on demand; the kernels leave its counters zero)."""
        need = ops.gemm_skinny_ws_bytes(M, N, K)
        ws = self._ws.get("skinny_ws")
        if ws is None or ws.numel() * 4 < need:
            ws = torch.zeros((need + 3) // 4, dtype=torch.float32, device=self._device)
            self._ws["skinny_ws"] = ws
        return ws

    def _linear(self, x, w, out, **kw):
        if x.shape[0] <= 16 and x.dtype == torch.bfloat16 and "bias" not in kw and w.shape[1] % 32 == 0:
            ep = kw.get("epilogue", 0)
            return ops.gemm_skinny(x, w, out, self._skinny_ws(x.shape[0], w.shape[0], w.shape[1]), residual=kw.get("residual"), epilogue=ep)
        return ops.gemm_nt(x, w, out, **kw)

    # ------------------------------------------------------------------ decoder stack
    def _decoder_layers(self, h: torch.Tensor, B: int, S: int, start_pos: int, rope_pos0: int,
                        k_caches, vt_caches, causal: bool) -> None:
        """h [B*S, dim] updated in place through all blocks (llama_ens5.py:237-249)."""
        a = self.args
        H, Hkv, hd, dim = self.n_heads, self.n_kv_heads, self.head_dim, a.dim
        rows = B * S
        pk = self._pack()
        cs = self._cos_sin_dev()
        xn = self._buf("xn", (rows, dim))
        qkv = self._buf("qkv", (rows, (H + 2 * Hkv) * hd))
        att = self._buf("att", (rows, H * hd))
        act = self._buf("act", (rows, self.ffn))
        Sk = start_pos + S
        scratch = None
        if S == 1 and h.dtype == torch.bfloat16:
            scratch = self._buf("attn_scratch", (2 * ops.attention_scratch_floats(B, H, hd, a.max_seq_len + 64),), torch.float32)
        ldq = qkv.stride(0)
        q8 = getattr(self, "_q8", None)
        w8a8 = (q8 is not None and getattr(self, "_fp8_prefill", False) and rows > 16 and h.dtype == torch.bfloat16
                and hd in (64, 128))
        if w8a8:
            if q8[0] != self._packed_version:
                raise RuntimeError("parameters changed after quantize_decode_weights(): call it again (or with mode=None)")
            xq = self._buf("fp8_x", (rows, max(dim, H * hd)), torch.uint8)
            aq = self._buf("fp8_act", (rows, self.ffn), torch.uint8)
            sx = self._buf("fp8_sx", (rows,), torch.float32)
        for i, lyr in enumerate(self.layers):
            kc, vc = k_caches[i], vt_caches[i]
            smax = kc.shape[2]
            strides = (S * ldq, ldq, hd,                       # q: view into the qkv buffer
                       Hkv * smax * hd, smax * hd, hd,         # k cache
                       Hkv * hd * smax, hd * smax, smax,       # v^T cache
                       S * H * hd, H * hd, hd)                 # out
            if w8a8:
                # W8A8 prefill (opt-in, BASELINE config 5): every decoder GEMM on fp8 operands -- activations quantised per
                # token on the fly (the RMSNorm output never exists in bf16), weights per output row, MX-scaled MFMA
                (wqkv_q, wqkv_s), (wo_q, wo_s), (w13_q, w13_s), (w2_q, w2_s) = q8[1][i]
                ops.quantize_rows_fp8(h, xq[:, :dim], sx, lyr.attention_norm.weight, a.norm_eps)
                ops.gemm_qkv_rope_fp8(xq[:, :dim], sx, wqkv_q, wqkv_s, qkv, kc, vc, cs, B, S, H, Hkv, hd, start_pos, rope_pos0)
                ops.attention(qkv, kc, vc, att, B, S, Sk, H, Hkv, hd, strides, causal and S > 1, scratch)
                ops.quantize_rows_fp8(att, xq[:, :H * hd], sx)
                ops.gemm_nt_fp8(xq[:, :H * hd], sx, wo_q, wo_s, h, residual=h)
                ops.quantize_rows_fp8(h, xq[:, :dim], sx, lyr.ffn_norm.weight, a.norm_eps)
                ops.gemm_nt_fp8(xq[:, :dim], sx, w13_q, w13_s, act, epilogue=ops.EPI_SWIGLU)
                ops.quantize_rows_fp8(act, aq, sx)
                ops.gemm_nt_fp8(aq, sx, w2_q, w2_s, h, residual=h)
                continue
            ops.rmsnorm(h, lyr.attention_norm.weight, xn, a.norm_eps)
            if rows > 16 and h.dtype == torch.bfloat16 and hd in (64, 128) and self._fuse_qkv_rope:
                # rotary embedding + cache write in the GEMM epilogue: qkv never makes a second trip through HBM
                ops.gemm_qkv_rope(xn, pk[f"wqkv.{i}"], qkv, kc, vc, cs, B, S, H, Hkv, hd, start_pos, rope_pos0)
            else:
                self._linear(xn, pk[f"wqkv.{i}"], qkv)
                ops.rope_kvcache(qkv, qkv, kc, vc, cs, B, S, H, Hkv, hd, start_pos, rope_pos0)
            ops.attention(qkv, kc, vc, att, B, S, Sk, H, Hkv, hd, strides, causal and S > 1, scratch)
            self._linear(att, lyr.attention.wo.weight, h, residual=h)
            ops.rmsnorm(h, lyr.ffn_norm.weight, xn, a.norm_eps)
            self._linear(xn, pk[f"w13.{i}"], act, epilogue=ops.EPI_SWIGLU)
            self._linear(act, lyr.feed_forward.w2.weight, h, residual=h)

    def quantize_decode_weights(self, mode: str = "fp8", prefill: bool = False) -> None:
        """Opt-in fp8 images of the four decoder matrices of every layer (BASELINE config 5; quantiser = a3v_quantize_rows_fp8).
        The single-call decode step streams them weight-only (half the bytes, bf16 activations).  ``prefill=True`` also runs
        the multi-token forward W8A8 (a3v_gemm_nt_fp8: activations quantised per token on the fly); otherwise prefill keeps
        the bf16 weights.  ``mode=None`` drops the images.  Re-run after the weights change."""
        self._q8 = None
        self._fp8_prefill = bool(prefill) and mode is not None
        self._layer_tab_key = None
        if mode is None:
            return
        if mode != "fp8":
            raise ValueError("only weight-only 'fp8' (OCP e4m3fn) is implemented")
        a = self.args
        if a.dim % 256 or self.ffn % 256 or (self.n_heads * self.head_dim) % 256 or self._dtype != torch.bfloat16:
            raise ValueError("fp8 decode weights need bf16 parameters and dim, ffn, n_heads*head_dim multiples of 256")
        pk = self._pack(check=True)
        q8 = []

        def quant(w):                      # per-row e4m3 image + fp32 scales by the HIP quantiser (a3v_quantize_rows_fp8)
            q = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
            sc = torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
            ops.quantize_rows_fp8(w, q, sc)
            return q, sc
        with torch.no_grad():
            for i, lyr in enumerate(self.layers):
                q8.append(tuple(quant(w) for w in (pk[f"wqkv.{i}"], lyr.attention.wo.weight, pk[f"w13.{i}"],
                                                   lyr.feed_forward.w2.weight)))
        self._q8 = (self._packed_version, q8)

    def _decode_step(self, h: torch.Tensor, B: int, pos: int) -> None:
        """seqlen == 1, bf16: the whole layer stack from one C call (a3v_llama_decode_step)."""
        import ctypes
        from ... import lib as _l
        a = self.args
        pk = self._pack()
        q8 = getattr(self, "_q8", None)
        if q8 is not None and q8[0] != self._packed_version:
            raise RuntimeError("parameters changed after quantize_decode_weights(): call it again (or with mode=None)")
        key = (self._packed_version, self._cache_shape, q8 is not None)
        if getattr(self, "_layer_tab_key", None) != key:
            tab = (_l.LlamaLayer * self.n_layers)()
            for i, lyr in enumerate(self.layers):
                if q8 is not None:
                    (tab[i].wqkv_q, tab[i].wqkv_s), (tab[i].wo_q, tab[i].wo_s), (tab[i].w13_q, tab[i].w13_s), (tab[i].w2_q, tab[i].w2_s) = \
                        [(q.data_ptr(), sc.data_ptr()) for q, sc in q8[1][i]]
                tab[i].attn_norm_w = lyr.attention_norm.weight.data_ptr()
                tab[i].wqkv = pk[f"wqkv.{i}"].data_ptr()
                tab[i].wo = lyr.attention.wo.weight.data_ptr()
                tab[i].ffn_norm_w = lyr.ffn_norm.weight.data_ptr()
                tab[i].w13 = pk[f"w13.{i}"].data_ptr()
                tab[i].w2 = lyr.feed_forward.w2.weight.data_ptr()
                tab[i].k_cache = self._k_cache[i].data_ptr()
                tab[i].vt_cache = self._vt_cache[i].data_ptr()
            self._layer_tab, self._layer_tab_key = tab, key
        H, Hkv, hd = self.n_heads, self.n_kv_heads, self.head_dim
        smax = self._k_cache[0].shape[2]
        xn = self._buf("xn", (B, a.dim))
        qkv = self._buf("qkv", (B, (H + 2 * Hkv) * hd))
        att = self._buf("att", (B, H * hd))
        act = self._buf("act", (B, self.ffn))
        scratch = self._buf("attn_scratch", (2 * ops.attention_scratch_floats(B, H, hd, a.max_seq_len + 64),), torch.float32)
        Bc = B if B <= 16 else (B + 1) // 2          # 17..32 rows run as two row chunks inside the C call
        sws = self._skinny_ws(Bc, max((H + 2 * Hkv) * hd, 2 * self.ffn, a.dim), max(a.dim, self.ffn))
        for (n_, k_) in (((H + 2 * Hkv) * hd, a.dim), (a.dim, H * hd), (2 * self.ffn, a.dim), (a.dim, self.ffn)):
            sws = self._skinny_ws(Bc, n_, k_)
        rc = _l.load().a3v_llama_decode_step(self._layer_tab, self.n_layers, h.data_ptr(), xn.data_ptr(), qkv.data_ptr(),
                                             att.data_ptr(), act.data_ptr(), scratch.data_ptr(), sws.data_ptr(), self._cos_sin_dev().data_ptr(),
                                             B, a.dim, H, Hkv, hd, self.ffn, smax, pos, a.norm_eps,
                                             torch.cuda.current_stream().cuda_stream)
        _l.check(rc, "a3v_llama_decode_step")

    # ------------------------------------------------------------------ vision
    def _vit_geometry(self):
        a = self.args
        g = a.vit_crop // a.vit_patch
        return g, g * g, g * g + 1     # grid, patch tokens, tokens incl. cls

    def clip_encode_image(self, views: torch.Tensor) -> torch.Tensor:
        """llama_ens5.py:351-375 on [N,3,c,c] views -> [N*L, width] (row = n*L + token)."""
        a = self.args
        pk = self._vision_images()
        vis = self.clip.visual
        vdt = vis.conv1.weight.dtype
        _mbuf = self._buf

        def _vbuf(name, shape):
            return _mbuf(name, shape, vdt)
        N = views.shape[0]
        g, T, L = self._vit_geometry()
        W, Hh = a.vit_width, a.vit_heads
        hd = W // Hh
        Kpad = pk["conv1"].shape[1]
        cols = _vbuf("vit_cols", (N * T, Kpad))
        ops.patch_im2col(views.contiguous(), cols, a.vit_patch)
        patch = _vbuf("vit_patch", (N * T, W))
        ops.gemm_nt(cols, pk["conv1"], patch)
        x = _vbuf("vit_x", (N * L, W))
        ops.vit_embed(patch, vis.class_embedding, vis.positional_embedding, x, N, T, W)
        ops.layernorm(x, vis.ln_pre.weight, vis.ln_pre.bias, x)
        y = _vbuf("vit_y", (N * L, W))
        qkv = _vbuf("vit_qkv", (N * L, 3 * W))
        Lpad = (L + 63) // 64 * 64
        vt = _vbuf("vit_vt", (N, Hh, hd, Lpad))
        att = _vbuf("vit_att", (N * L, W))
        mlp = _vbuf("vit_mlp", (N * L, 4 * W))
        act = ops.EPI_QUICKGELU if a.vit_quick_gelu else ops.EPI_GELU
        ld = 3 * W
        strides = (L * ld, ld, hd, L * ld, hd, ld, Hh * hd * Lpad, hd * Lpad, Lpad, L * W, W, hd)
        for blk in vis.transformer.resblocks:
            ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias, y)
            ops.gemm_nt(y, blk.attn.in_proj_weight, qkv, bias=blk.attn.in_proj_bias)
            ops.vt_pack(qkv[:, 2 * W:], ld, vt, N, L, Hh, hd, Lpad)
            ops.attention(qkv, qkv[:, W:], vt, att, N, L, L, Hh, Hh, hd, strides, False, None)
            ops.gemm_nt(att, blk.attn.out_proj.weight, x, bias=blk.attn.out_proj.bias, residual=x)
            ops.layernorm(x, blk.ln_2.weight, blk.ln_2.bias, y)
            ops.gemm_nt(y, blk.mlp.c_fc.weight, mlp, bias=blk.mlp.c_fc.bias, epilogue=act)
            ops.gemm_nt(mlp, blk.mlp.c_proj.weight, x, bias=blk.mlp.c_proj.bias, residual=x)
        feats = _vbuf("vit_feats", (N * L, W))
        ops.layernorm(x, vis.ln_post.weight, vis.ln_post.bias, feats)
        return feats

    def _image_slots(self):
        """(start tag, end tag) parameters per image slot; slot 0 = RGB (llama_ens5.py:338-339)."""
        return [(self.start_img, self.end_img)]

    @property
    def words_per_image(self) -> int:
        _, _, L = self._vit_geometry()
        return (self.args.qformer_tokens + L + 2) * self.args.n_views

    def _image_row_maps(self, B: int, S: int, slots=(0,)):
        """int32 device maps from projector rows to rows of the [B*S, dim] sequence buffer
        for the layout of llama_ens5.py:471-479: h = [BOS | per image: per view (start, [qformer], clip, end) | text].
        Image j of the call occupies words [j*Wv, (j+1)*Wv); projector rows are ordered (image, view, batch)."""
        key = (B, S, tuple(slots), str(self._device))
        m = self._row_maps.get(key)
        if m is not None:
            return m
        a = self.args
        _, _, L = self._vit_geometry()
        Q, V, J = a.qformer_tokens, a.n_views, len(slots)
        per_view = Q + L + 2
        clip_map = torch.empty(J * V * B * L, dtype=torch.int32)
        qf_map = torch.empty(J * V * B * max(Q, 1), dtype=torch.int32)
        start_rows, end_rows = [[] for _ in slots], [[] for _ in slots]
        for j in range(J):
            for v in range(V):
                for b in range(B):
                    base = b * S + 1 + (j * V + v) * per_view
                    start_rows[j].append(base)
                    end_rows[j].append(base + 1 + Q + L)
                    n = (j * V + v) * B + b
                    clip_map[n * L:(n + 1) * L] = torch.arange(base + 1 + Q, base + 1 + Q + L, dtype=torch.int32)
                    if Q:
                        qf_map[n * Q:(n + 1) * Q] = torch.arange(base + 1, base + 1 + Q, dtype=torch.int32)
        dev = self._device
        m = (clip_map.to(dev), qf_map.to(dev) if Q else None,
             [torch.tensor(r, dtype=torch.int32, device=dev) for r in start_rows],
             [torch.tensor(r, dtype=torch.int32, device=dev) for r in end_rows])
        self._row_maps[key] = m
        return m

    def _gather_views(self, images, B: int, dtype=None) -> torch.Tensor:
        """[J*V*B, 3, c, c] crops of the J images of a call (llama_ens5.py:381-392: fp16 bicubic 2x down + quadrants)."""
        a = self.args
        c, J = a.vit_crop, len(images)
        if a.n_views == 5:
            views = self._buf("views", (J * 5 * B, 3, c, c), dtype)
            for j, img in enumerate(images):
                assert img.shape[-1] == 2 * c and img.shape[-2] == 2 * c, img.shape
                ops.split_views(img.contiguous(), views[j * 5 * B:(j + 1) * 5 * B])
            return views
        assert a.n_views == 1
        if J == 1:
            return images[0]
        views = self._buf("views", (J * B, 3, c, c), images[0].dtype)
        for j, img in enumerate(images):
            views[j * B:(j + 1) * B].copy_(img)        # data movement only
        return views

    def encode_image_into(self, h: torch.Tensor, image, B: int, S: int,
                          qformer_feats: Optional[torch.Tensor] = None,
                          extra_feats: Optional[List[torch.Tensor]] = None, slots=(0,)) -> None:
        """llama_ens5.py:377-458 + 471-478, writing the image words straight into rows 1..W of
        every sequence of ``h`` [B*S, dim].  ``image`` is one tensor or a list of len(slots) tensors (the
        two-image plugin encodes RGB and depth in one ViT/projector pass)."""
        a = self.args
        dtp = self._dtype
        _, _, L = self._vit_geometry()
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        assert len(images) == len(slots)
        views = self._gather_views(images, B)
        N = len(images) * a.n_views * B
        feats = self.clip_encode_image(views)
        if extra_feats is None and self.extra_feat_fns:
            extra_feats = [fn(views) for fn in self.extra_feat_fns]
        if qformer_feats is None and self.qformer_fn is not None:
            qformer_feats = self.qformer_fn(views)
        pk = self._pack()
        pw = pk["visual_proj"]
        if a.extra_feat_dim or pw.shape[1] != feats.shape[1]:
            assert (extra_feats is not None) == bool(a.extra_feat_dim), "extra_feat_dim set but no extra features given"
            cat = self._buf("proj_in", (N * L, pw.shape[1]))
            cat.zero_()
            cat[:, :a.vit_width] = feats              # data movement only (concat, llama_ens5.py:436-440)
            o = a.vit_width
            for e in (extra_feats or []):
                e2 = e.reshape(N * L, -1).to(dtp)
                cat[:, o:o + e2.shape[1]] = e2
                o += e2.shape[1]
            feats = cat
        proj = self._buf("proj_out", (N * L, a.dim))
        vp0, vp1 = getattr(self.visual_proj, "0"), getattr(self.visual_proj, "1")
        ops.gemm_nt(feats, pw, proj, bias=vp0.bias)
        clip_map, qf_map, start_rows, end_rows = self._image_row_maps(B, S, slots)
        ops.layernorm(proj, vp1.weight, vp1.bias, h, row_map=clip_map)
        if a.qformer_tokens:
            assert qformer_feats is not None, "qformer_tokens set but no Q-Former features given"
            Q = a.qformer_tokens
            qin = qformer_feats.reshape(N * Q, 768).to(dtp).contiguous()
            qp0, qp1 = getattr(self.qformer_proj, "0"), getattr(self.qformer_proj, "1")
            qout = self._buf("qf_out", (N * Q, a.dim))
            ops.gemm_nt(qin, qp0.weight, qout, bias=qp0.bias)
            ops.layernorm(qout, qp1.weight, qp1.bias, h, row_map=qf_map)
        tags = self._image_slots()
        for j, sl in enumerate(slots):
            ops.fill_rows(tags[sl][0].view(-1), h, start_rows[j])
            ops.fill_rows(tags[sl][1].view(-1), h, end_rows[j])

    # ------------------------------------------------------------------ forward (teacher forced)
    def forward(self, examples: torch.Tensor, image: Optional[torch.Tensor] = None, *,
                qformer_feats=None, extra_feats=None) -> torch.Tensor:
        """llama_ens5.py:461-487: logits [B, T, V] in the model dtype for all text positions."""
        return self._forward_images(examples, [] if image is None else [image], (0,) if image is not None else (),
                                    qformer_feats, extra_feats)

    def _forward_images(self, examples, images, slots, qformer_feats=None, extra_feats=None, out_from=None) -> torch.Tensor:
        self._destroy_kv_cache()
        self._pack(check=True)
        a = self.args
        B, T = examples.shape
        W = self.words_per_image * len(images) if images else 0
        S = T + W
        h = self._buf("h", (B * S, a.dim))
        ops.embed_assemble(examples.contiguous(), self.tok_embeddings.weight, h, B, T, W, a.dim)
        if images:
            self.encode_image_into(h, images, B, S, qformer_feats, extra_feats, slots)
        spad = (S + 63) // 64 * 64
        kc = self._buf("fw_k", (B, self.n_kv_heads, spad, self.head_dim))
        vc = self._buf("fw_vt", (B, self.n_kv_heads, self.head_dim, spad))
        L = self.n_layers
        self._decoder_layers(h, B, S, 0, 0, [kc] * L, [vc] * L, True)
        xn = self._buf("xn_final", (B * S, a.dim))
        ops.rmsnorm(h, self.norm.weight, xn, a.norm_eps)
        o0 = W if out_from is None else out_from       # h[:, image_words:] (llama_ens5.py:486)
        out = torch.empty(B, S - o0, a.vocab_size, dtype=self._dtype, device=self._device)
        xv = xn.view(B, S, a.dim)
        for b in range(B):
            ops.gemm_nt(xv[b, o0:], self.output.weight, out[b])
        return out

    # ------------------------------------------------------------------ forward_inference (KV cached)
    @torch.no_grad()
    def forward_inference(self, tokens: torch.Tensor, start_pos: int, image: Optional[torch.Tensor] = None, *,
                          qformer_feats=None, extra_feats=None) -> torch.Tensor:
        """llama_ens5.py:490-531: fp32 logits [B, V] of the last position."""
        return self._forward_inference_images(tokens, start_pos, [] if image is None else [image],
                                              (0,) if image is not None else (), qformer_feats, extra_feats)

    def _forward_inference_images(self, tokens, start_pos, images, slots, qformer_feats=None, extra_feats=None) -> torch.Tensor:
        a = self.args
        B, T = tokens.shape
        if start_pos == 0:
            self._allocate_kv_cache(B)
            self._pack(check=True)
        W = 0
        image = images if images else None
        if image is not None:
            assert start_pos == 0
            W = self.words_per_image * len(images)
            self.cache_image_words = W
            assert self.cache_image_words == self.image_words
            rope0 = 0
        else:
            if start_pos == 0:
                self.cache_image_words = 0
                rope0 = 0
            else:
                start_pos = start_pos + self.cache_image_words
                rope0 = start_pos
        S = T + W
        h = self._buf("h", (B * S, a.dim))
        ops.embed_assemble(tokens.contiguous(), self.tok_embeddings.weight, h, B, T, W, a.dim)
        if image is not None:
            self.encode_image_into(h, image, B, S, qformer_feats, extra_feats, slots)
        if (S == 1 and (B <= 16 or (B <= 32 and a.dim % 128 == 0 and self.ffn % 128 == 0)) and self._dtype == torch.bfloat16
                and self.head_dim in (64, 128) and a.dim % 32 == 0 and self.ffn % 32 == 0
                and not getattr(self, "_per_kernel_decode", False)):     # test hook: run the step kernel by kernel
            # the C entry says up front which form it takes (a3v_llama_decode_step_form): 17..32 rows need the fused GEMV forms (two row
            # chunks); a geometry they do not take goes through the general kernels -- decided BEFORE h or the KV cache are touched, an
            # error from inside the step is never papered over
            w8 = 1 if getattr(self, "_q8", None) is not None else 0
            if _lib.load().a3v_llama_decode_step_form(B, a.dim, a.n_heads, self.n_kv_heads, self.head_dim, self.ffn, w8) or B <= 16:
                self._decode_step(h, B, start_pos)
            else:
                self._decoder_layers(h, B, S, start_pos, rope0, self._k_cache, self._vt_cache, True)
        else:
            self._decoder_layers(h, B, S, start_pos, rope0, self._k_cache, self._vt_cache, True)
        last = h.view(B, S, a.dim)[:, -1, :]            # strided rows, no copy
        xn = self._buf("xn_last", (B, a.dim))
        ops.rmsnorm(last, self.norm.weight, xn, a.norm_eps)
        logits = self._buf("logits", (B, a.vocab_size), torch.float32)
        if self._dtype == torch.float32:
            ops.gemm_nt(xn, self.output.weight, logits)
        else:
            self._linear(xn, self.output.weight, logits, epilogue=ops.EPI_OUT_F32)
        # the reference returns a fresh tensor (output(h[:, -1, :]).float(), llama_ens5.py:530-531); `logits` is a cached workspace
        # that the next call overwrites, so hand out a copy (B x V fp32 = 1 MB at bs 8: microseconds next to a 4 ms step)
        return logits.clone()
