"""LoRA variant of the ``llama_ens5`` plugin (BASELINE config 3: LoRA fine-tune of the same backbone).

The reference snapshot ships the adapter layers (``model/peft.py``: ``y = W x (+ b) + lora_b(lora_a(x))``, no alpha / rank
scaling, ``lora_a ~ trunc_normal(0.02)``, ``lora_b = 0``, state-dict keys ``<linear>.lora_a.weight`` / ``.lora_b.weight``)
and the façade hooks (``MetaModel.is_peft`` meta.py:72, whole-model FSDP wrap when PEFT main_finetune.py:246) but no plugin
that instantiates them (``llama_peft`` is absent, ``LLM/__init__.py:3``).  This plugin puts adapters on the seven decoder
linears of every block -- the upstream LLaMA2-Accessory ``llama_peft`` arrangement -- on top of the ``llama_ens5``
multimodal stack, with the same key names, so a checkpoint of base + adapters loads by name.

Trainable: adapters, RMSNorm weights, the vision->language projectors and the image tags (everything the base plugin
trains that is not a frozen backbone matrix).  Base matrices, embeddings and the LM head are frozen.

Kernels: the three (two) adapters that share an input are fused -- ``t = x . [A_q; A_k; A_v]^T`` (one GEMM, rank padded to
a 64 multiple) and ``y += t . blockdiag(B_q, B_k, B_v)^T`` (one GEMM with the residual epilogue: ``bf16(acc) + y``, the
reference's rounding order) -- so an adapter group costs two skinny-N / skinny-K MFMA GEMMs instead of six.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn as nn

from ... import ops
from . import llama_ens5 as base
from .llama_ens5 import _W


@dataclass
class ModelArgs(base.ModelArgs):
    lora_rank: int = 16
    bias_tuning: bool = False          # the Llama linears carry no bias (llama_ens5.py:63-90); kept for config compatibility


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


class Transformer(base.Transformer):
    is_peft = True

    def __init__(self, args: ModelArgs, with_visual: bool = False):
        super().__init__(args, with_visual=with_visual)
        if args.bias_tuning:
            raise NotImplementedError("bias_tuning: the decoder linears of llama_ens5 have no bias terms")
        r = args.lora_rank
        assert r > 0 and r % 8 == 0, "lora_rank must be a positive multiple of 8"
        self.lora_rank = r
        for lyr in self.layers:
            for mod in (lyr.attention.wq, lyr.attention.wk, lyr.attention.wv, lyr.attention.wo,
                        lyr.feed_forward.w1, lyr.feed_forward.w2, lyr.feed_forward.w3):
                out_f, in_f = mod.weight.shape
                mod.lora_a = _W(r, in_f, init="normal")
                nn.init.trunc_normal_(mod.lora_a.weight, std=0.02)      # peft.py:72-74
                mod.lora_b = _W(out_f, r, init="normal")
                nn.init.zeros_(mod.lora_b.weight)                       # peft.py:76
        self._per_kernel_decode = True       # the single-call decode step has no adapter hooks
        self._lora_img: Dict[str, torch.Tensor] = {}
        self._lora_ver = None

    def get_trainable_params(self, pretrain_stage: bool = False):
        frozen_pre = ("qformer.", "openclip_convnext_xxl.", "clip.", "dinov2_vitg14.", "tok_embeddings.", "output.")
        out = {}
        for n, p in self.named_parameters():
            if n.startswith(frozen_pre):
                continue
            if n.startswith("layers.") and not ("lora_" in n or "norm" in n):
                continue
            out[n] = p
        return out

    # ------------------------------------------------------------------ fused adapter images
    def lora_groups(self, i: int):
        """(key, [modules sharing the input], interleave16) for layer i, in the row order of the fused base GEMMs."""
        a, f = self.layers[i].attention, self.layers[i].feed_forward
        return [(f"qkv.{i}", [a.wq, a.wk, a.wv], False), (f"wo.{i}", [a.wo], False),
                (f"w13.{i}", [f.w1, f.w3], True), (f"w2.{i}", [f.w2], False)]

    def lora_images(self, dtype: Optional[torch.dtype] = None, interleave_w13: bool = True) -> Dict[str, torch.Tensor]:
        """Per group: ``A`` [Rp, in] (stacked lora_a, zero rows up to Rp = pad64(n*r)) and ``B`` [N, Rp] (block-diagonal
        lora_b in the row order of the fused base weight; w1/w3 rows interleaved in 16-row blocks when the base image is)."""
        dtype = dtype or self._dtype
        from ...util import param_state_key
        ver = (tuple(param_state_key(p) for n, p in self.named_parameters() if "lora_" in n), dtype, interleave_w13, str(self._device))
        if self._lora_ver == ver:
            return self._lora_img
        r = self.lora_rank
        im: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for i in range(self.n_layers):
                for key, mods, inter in self.lora_groups(i):
                    Rp = _pad64(len(mods) * r)
                    in_f = mods[0].weight.shape[1]
                    A = torch.zeros(Rp, in_f, dtype=dtype, device=self._device)
                    blocks = []
                    for j, mod in enumerate(mods):
                        A[j * r:(j + 1) * r] = mod.lora_a.weight.to(dtype)
                        Bj = torch.zeros(mod.weight.shape[0], Rp, dtype=dtype, device=self._device)
                        Bj[:, j * r:(j + 1) * r] = mod.lora_b.weight.to(dtype)
                        blocks.append(Bj)
                    if inter and interleave_w13:
                        nb = blocks[0].shape[0] // 16
                        Bm = torch.stack([blocks[0].view(nb, 16, Rp), blocks[1].view(nb, 16, Rp)], dim=1).reshape(-1, Rp)
                    else:
                        Bm = torch.cat(blocks, dim=0)
                    im[key + ".A"], im[key + ".B"] = A, Bm.contiguous()
        self._lora_img, self._lora_ver = im, ver
        return im

    def _lora_add(self, key: str, x: torch.Tensor, y: torch.Tensor) -> None:
        """y += lora_b(lora_a(x)) for a fused group (peft.py:89-95)."""
        im = self.lora_images()
        A, Bm = im[key + ".A"], im[key + ".B"]
        t = self._buf("lora_t", (x.shape[0], A.shape[0]))
        self._linear(x, A, t)
        self._linear(t, Bm, y, residual=y)

    # ------------------------------------------------------------------ decoder stack with adapters
    def _decoder_layers(self, h: torch.Tensor, B: int, S: int, start_pos: int, rope_pos0: int,
                        k_caches, vt_caches, causal: bool) -> None:
        a = self.args
        H, Hkv, hd, dim = self.n_heads, self.n_kv_heads, self.head_dim, a.dim
        rows = B * S
        pk = self._pack()
        cs = self._cos_sin_dev()
        xn = self._buf("xn", (rows, dim))
        qkv = self._buf("qkv", (rows, (H + 2 * Hkv) * hd))
        att = self._buf("att", (rows, H * hd))
        gu = self._buf("gu", (rows, 2 * self.ffn))
        act = self._buf("act", (rows, self.ffn))
        o = self._buf("lora_o", (rows, dim))
        Sk = start_pos + S
        scratch = None
        if S == 1 and h.dtype == torch.bfloat16:
            scratch = self._buf("attn_scratch", (2 * ops.attention_scratch_floats(B, H, hd, a.max_seq_len + 64),), torch.float32)
        ldq = qkv.stride(0)
        for i, lyr in enumerate(self.layers):
            kc, vc = k_caches[i], vt_caches[i]
            smax = kc.shape[2]
            ops.rmsnorm(h, lyr.attention_norm.weight, xn, a.norm_eps)
            if rows > 16 and h.dtype == torch.bfloat16 and hd in (64, 128) and self._fuse_qkv_rope:
                # prefill: the adapter term is written into the qkv buffer and enters the fused qkv / RoPE / cache GEMM as an
                # additive term before the rotation (in place: a lane reads its delta before it stores the rotated q)
                im = self.lora_images()
                A, Bm = im[f"qkv.{i}.A"], im[f"qkv.{i}.B"]
                t = self._buf("lora_t", (rows, A.shape[0]))
                self._linear(xn, A, t)
                self._linear(t, Bm, qkv)
                ops.gemm_qkv_rope(xn, pk[f"wqkv.{i}"], qkv, kc, vc, cs, B, S, H, Hkv, hd, start_pos, rope_pos0, delta=qkv)
            else:
                self._linear(xn, pk[f"wqkv.{i}"], qkv)
                self._lora_add(f"qkv.{i}", xn, qkv)
                ops.rope_kvcache(qkv, qkv, kc, vc, cs, B, S, H, Hkv, hd, start_pos, rope_pos0)
            strides = (S * ldq, ldq, hd, Hkv * smax * hd, smax * hd, hd, Hkv * hd * smax, hd * smax, smax, S * H * hd, H * hd, hd)
            ops.attention(qkv, kc, vc, att, B, S, Sk, H, Hkv, hd, strides, causal and S > 1, scratch)
            # out = wo(att) + lora (rounded), then the residual add -- the reference's order (peft.py:95, llama_ens5.py:238)
            self._linear(att, lyr.attention.wo.weight, o)
            self._lora_add(f"wo.{i}", att, o)
            ops.add2d(h, o)
            ops.rmsnorm(h, lyr.ffn_norm.weight, xn, a.norm_eps)
            self._linear(xn, pk[f"w13.{i}"], gu)                 # un-fused SwiGLU: the adapters add before the activation
            self._lora_add(f"w13.{i}", xn, gu)
            ops.swiglu_fwd(gu, act, self.ffn, interleaved=True)
            self._linear(act, lyr.feed_forward.w2.weight, o)
            self._lora_add(f"w2.{i}", act, o)
            ops.add2d(h, o)
