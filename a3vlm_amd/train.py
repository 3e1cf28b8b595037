"""Training step of the plugin: forward + backward of the multimodal path through HIP kernels.

Reference semantics (model/accessory/engine_finetune.py:44-68, main_finetune.py:212-217,268-276):
autocast(bf16) over fp32 master weights of the trainables, bf16 frozen visual encoder, fp32
residual stream, activation checkpointing per TransformerBlock, loss = CE(ignore_index=0).
Here that is: one custom autograd node per step whose forward runs the kernels and keeps only each
block's input (fp32) and whose backward recomputes a block, then back-propagates through it with
GEMMs on transposed operand images (dgrad / wgrad), RMSNorm / SwiGLU / RoPE / attention backward
kernels, and accumulates fp32 gradients straight into one flat buffer that ``param.grad`` views
alias (wq|wk|wv and w1|w3 share fused buffers).  ``on_layer_grads_ready`` lets the DP reducer start
an all-reduce of a layer's gradient range while earlier layers are still back-propagating.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import ops
from .util import install_param_epoch_hook, optimizer_steps, param_state_key


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def skinny_slices(M: int, N: int, K: int, cus: int) -> int:
    """Split-K slices for an adapter-sized product out[M, N] = a[M, K] @ w[N, K]^T (``TrainEngine._skinny``).  ``cus`` = compute units of
    the device (0: not a GPU tensor).  N <= 64 and M >= 2048: 256-row blocks, ONE resident block per CU -- a3v_gemm_nt_splitk takes that form
    when S x ceil(M / 256) fills between half and all of the CUs: 8728 rows -> 35 row tiles x 7 slices = 245 blocks; one slice more and the
    blocks that wait for a CU double the time (profiles/r04k_skinny_stages_sweep.txt).  A3V_SKINNY_LEGACY=1 (with A3V_SKINNY_NARROW=3
    A3V_SKINNY_STAGES=2): the slice count of rounds 2-3."""
    blocks = ((M + 127) // 128) * ((N + 127) // 128)
    nk = K // 64
    S = 1
    while blocks * S < 512 and S * 2 <= nk and S < 32:
        S *= 2
    if N <= 64 and M >= 2048 and cus > 0 and os.environ.get("A3V_SKINNY_LEGACY", "0") != "1":
        S = max(1, min(cus // ((M + 255) // 256), nk // 2, 32))
    elif N <= 64 and M >= 512 and S > 4:
        S = 4            # 64-row tiles (a3v_gemm_nt_splitk): M / 64 blocks per slice already fill the chip; more slices only add reduce work
    return S


class TrainEngine:
    # A/B and test hooks (environment A3V_FUSE_QKV_ROPE=0 / A3V_TN_WGRAD=0 / A3V_NN_DGRAD=0 flips the default for a whole process)
    fuse_qkv_rope = os.environ.get("A3V_FUSE_QKV_ROPE", "1") != "0"   # qkv GEMM with the RoPE / cache-write epilogue
    tn_wgrad = os.environ.get("A3V_TN_WGRAD", "1") != "0"              # weight gradients by a3v_gemm_tn (else transposes + NT)
    nn_dgrad = os.environ.get("A3V_NN_DGRAD", "1") != "0"              # input gradients by a3v_gemm_nn (else NT on W^T images); unset: see __init__
    packed_attn_bwd = os.environ.get("A3V_PACKED_ATTN_BWD", "1") != "0"  # attention backward writes the rotated-back qkv gradient itself
    lora_kext = os.environ.get("A3V_LORA_KEXT", "1") != "0"            # adapters inside the main GEMMs: [x | t] . [W | B]^T (K extended by Rp)
    # LoRA input gradients dx = [dy | dt] . [W ; A]: the base matrices are FROZEN, so a transposed image [W^T | A^T] can be kept for
    # free and the product runs on the NT ring kernel (1.38-1.42 PF) instead of the NN kernel (1.28-1.30) for the groups named here
    # ("0": none, "all", or a comma list of qkv / wo / w13 / w2).  Same-box LoRA step: 253.0 ms without, 248.9 with wo,w13,w2, 247.9 all
    lora_nt_dgrad = os.environ.get("A3V_LORA_NT_DGRAD", "all")
    strip_wgrad = os.environ.get("A3V_STRIP_WGRAD", "1") != "0"        # adapter weight gradients by a3v_gemm_tn_strip (0: the 256 x 256 TN split-K kernel)
    fuse_swiglu_bwd = os.environ.get("A3V_FUSE_SWIGLU_BWD", "1") != "0"   # LoRA: SwiGLU backward in the epilogue of w2's input-gradient GEMM (0: separate pass, A/B)

    def __init__(self, model, compute_dtype: torch.dtype = torch.bfloat16, recompute: Optional[bool] = None,
                 stream_dtype: Optional[torch.dtype] = None, zero1_world: int = 0):
        """``stream_dtype``: dtype of the residual stream h, of its per-layer checkpoints and of its gradient dh.  Default = the compute
        dtype: under ``--precision bf16`` the reference wraps the model in FSDP with MixedPrecision(param_dtype=bf16)
        (main_finetune.py:241-263) and runs it under autocast (engine_finetune.py:44-50), so the embeddings come out in bf16 and every
        ``h = h + block(h)`` -- and autograd's gradient of it -- is a bf16 tensor; the fp32 masters only exist for the optimizer.  fp32
        (``A3V_STREAM_FP32=1`` or ``stream_dtype=torch.float32``) keeps the more precise stream of rounds 1-2 (+5 ms per step at 7B:
        twice the bytes in every norm / norm-backward / residual epilogue pass).
        ``recompute``: True = keep only each block's input and re-run the block in backward (the reference's
        activation checkpointing, main_finetune.py:268-276); False = keep every block's intermediates (about
        1.15 GB per 7B layer at 8 x 1091 tokens -- affordable in 288 GB of HBM and ~1/4 fewer GEMM FLOPs per step).
        None = decide from free HBM at the first step.
        ``zero1_world`` = N > 0 (``a3vlm_amd.zero1``, the reference's FSDP(SHARD_GRAD_OP) sizing for configs[3]): trainable parameters left in
        the compute dtype (the big matrices: not promoted to fp32) have NO fp32 replica here -- they become views of one flat buffer in the
        compute dtype laid out like the flat gradient buffer (every bucket's sharded span padded to a multiple of 64 N elements, so that
        reduce-scatter / all-gather work on it in place), the GEMM images are views of the same storage, and ``Zero1Optimizer`` owns the
        fp32 master / AdamW state of 1/N of it.  fp32 parameters (norms, projector) stay replicated as before."""
        install_param_epoch_hook()     # weight-image caches must notice optimizers that do not bump Tensor._version
        self.m = model
        self.lora = int(getattr(model, "lora_rank", 0) or 0) > 0
        self.act = compute_dtype
        if stream_dtype is None:
            stream_dtype = torch.float32 if os.environ.get("A3V_STREAM_FP32", "0") == "1" else compute_dtype
        self.stream = stream_dtype
        self.recompute = recompute
        self._img = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._flat: Optional[torch.Tensor] = None
        self._views: Dict[str, torch.Tensor] = {}
        self._ranges: List[Tuple[str, int, int]] = []       # (bucket name, start, end) in the flat grad buffer
        self._fresh: set = set()                            # grads attached this step whose storage is still undefined
        self._gemm_written: set = set()                     # names whose gradient comes from exactly one wgrad GEMM per micro-step
        self.on_layer_grads_ready: Optional[Callable[[str, int, int], None]] = None
        self.sumsq_sink = None               # see _wgrad
        self.static_grad_scale: Optional[float] = None      # set by a trainer that knows d(total)/d(loss) (1 / accum_iter)
        self._saved = None
        self._weights_ready: Dict[str, torch.cuda.Event] = {}   # bucket -> event of FusedAdamW.step(overlap=True), see await_weights
        self.zero1_world = int(zero1_world or 0)
        if "A3V_NN_DGRAD" not in os.environ:
            # full fine-tune (round 5): input gradients on the NT ring kernel over transposed images W^T that a3v_adamw_scaled_t keeps
            # current in the optimizer pass (+2 B of writes per parameter; same-box 336.0 -> 328 ms per 7B step).  ZeRO-1 updates flat
            # slices (no per-matrix pass to write W^T from): the NN kernel on the forward images stays there.
            self.nn_dgrad = bool(self.zero1_world)
        self._flat_params: Optional[torch.Tensor] = None    # zero1: the sharded parameters' storage (compute dtype, flat-gradient layout)
        self._z1: List[tuple] = []                          # zero1: (bucket, start, end, shard_start, shard_end, [(seg_start, seg_end, name)])
        self._z1_fresh = True                               # zero1: the sharded gradients restart (store / zero) at the next backward

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, shape, dtype=None, zero=False):
        dtype = dtype or self.act
        key = (name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = torch.empty(*shape, dtype=dtype, device=self.m._device)
            self._ws[key] = t
        if zero:
            t.zero_()
        return t

    # ------------------------------------------------------------------ optimizer hand-off (a3vlm_amd.optim.FusedAdamW(engine=...))
    def image_sink(self, p: torch.Tensor) -> Optional[torch.Tensor]:
        """Where the optimizer may store the bf16 value of the updated parameter p (its rows of the fused forward image)."""
        return self._images().sink(p)

    def image_sink_t(self, p: torch.Tensor):
        """(bf16 view of p's columns inside the TRANSPOSED image of its fused matrix, row stride) when that image is in use (full
        fine-tune input gradients on the NT kernel) -- ``a3v_adamw_scaled_t`` then keeps it current; else None."""
        if self.nn_dgrad or self.lora:
            return None
        return self._images().sink_t(p)

    def images_adopted(self, written_ids, written_t_ids=()) -> None:
        self._opt_stepped = True
        self._images().adopted(written_ids, written_t_ids)

    def forward_order(self):
        """Gradient buckets in the order the forward first touches their parameters: ``FusedAdamW.step(overlap=True)`` updates
        them in this order on its own stream, and the next forward waits per bucket (``await_weights``) -- the HBM-bound update
        of layer i+1.. runs under the MFMA-bound GEMMs of layers ..i of the next step."""
        lay = self._layout()
        rank = {"embed": 0, "vision_proj": 1, "head": 3}
        return sorted(lay, key=lambda it: rank.get(it[0], 2))          # stable: the layers keep their order

    def await_weights(self, bucket: str) -> None:
        """The current stream waits until the optimizer has written this bucket's parameters (no-op without an overlapped step)."""
        ev = self._weights_ready.pop(bucket, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def sync_optimizer(self) -> None:
        """The current stream waits for every update still in flight: before anything but ``forward_loss`` reads parameters or
        writes gradients (backward, checkpoints, evaluation through the model)."""
        if self._weights_ready:
            for ev in self._weights_ready.values():
                torch.cuda.current_stream().wait_event(ev)
            self._weights_ready.clear()

    def _kext_cols(self, key: str) -> int:
        """Width of the adapter block appended to the image / input of decoder GEMM ``key`` (0: adapters not folded in)."""
        if not self._kext() or not key.startswith(("qkv.", "wo.", "w13.", "w2.")):
            return 0
        n = {"qkv": 3, "wo": 1, "w13": 2, "w2": 1}[key.split(".")[0]]
        return _pad64(n * int(self.m.lora_rank))

    def _nt_dgrad(self, key: str) -> bool:
        sel = self.lora_nt_dgrad
        if not self._kext() or sel in ("", "0"):
            return False
        g = key.split(".")[0]
        frozen = not any(q.requires_grad for q in self._group_base_params(key))
        return frozen and (sel == "all" or g in sel.split(","))

    def _group_base_params(self, key: str):
        g, i = key.split(".")[0], int(key.split(".")[1])
        a, f = self.m.layers[i].attention, self.m.layers[i].feed_forward
        return {"qkv": (a.wq.weight, a.wk.weight, a.wv.weight), "wo": (a.wo.weight,), "w13": (f.w1.weight, f.w3.weight),
                "w2": (f.w2.weight,)}[g]

    def _kext(self) -> int:
        """Columns appended to the inputs / weight images of the four decoder GEMMs when the adapters ride inside them
        (LoRA, bf16): y = [x | t] . [W | B]^T with t = x . A^T -- no separate read-modify-write pass over y."""
        if not (self.lora and self.lora_kext and self.act == torch.bfloat16):
            return 0
        return _pad64(3 * int(self.m.lora_rank))          # the widest group (wq|wk|wv) sets one width for all four

    def _check_dtypes(self):
        for n, p in self.m.get_trainable_params().items():
            if p.requires_grad and p.dtype != torch.float32 and not (self.zero1_world and p.dtype == self.act):
                raise TypeError(f"trainable parameter {n} must be fp32 (promote_trainable_params_to_fp32, "
                                f"util/tensor_type.py:60-66); got {p.dtype}")

    # ------------------------------------------------------------------ flat gradient buffer
    def _layout(self):
        """Order = backward completion order reversed (layer-major), so one layer's grads are contiguous."""
        m = self.m
        items: List[Tuple[str, List[Tuple[str, torch.nn.Parameter]]]] = []
        items.append(("embed", [("tok_embeddings.weight", m.tok_embeddings.weight)]))
        for i, l in enumerate(m.layers):
            p = f"layers.{i}."
            items.append((f"layer{i}", [
                (p + "attention.wq.weight", l.attention.wq.weight), (p + "attention.wk.weight", l.attention.wk.weight),
                (p + "attention.wv.weight", l.attention.wv.weight), (p + "attention.wo.weight", l.attention.wo.weight),
                (p + "feed_forward.w1.weight", l.feed_forward.w1.weight), (p + "feed_forward.w3.weight", l.feed_forward.w3.weight),
                (p + "feed_forward.w2.weight", l.feed_forward.w2.weight),
                (p + "attention_norm.weight", l.attention_norm.weight), (p + "ffn_norm.weight", l.ffn_norm.weight)]))
            if self.lora:
                # group-major: the lora_a of the modules that share an input are adjacent ([n r, in] as one matrix: the group's dA
                # GEMM writes the gradients in place), then their lora_b
                for group in ((("attention.wq", l.attention.wq), ("attention.wk", l.attention.wk), ("attention.wv", l.attention.wv)),
                              (("attention.wo", l.attention.wo),),
                              (("feed_forward.w1", l.feed_forward.w1), ("feed_forward.w3", l.feed_forward.w3)),
                              (("feed_forward.w2", l.feed_forward.w2),)):
                    for sub, mod in group:
                        items[-1][1].append((p + sub + ".lora_a.weight", mod.lora_a.weight))
                    for sub, mod in group:
                        items[-1][1].append((p + sub + ".lora_b.weight", mod.lora_b.weight))
        items.append(("head", [("norm.weight", m.norm.weight), ("output.weight", m.output.weight)]))
        if m.with_visual:
            vp0, vp1 = getattr(m.visual_proj, "0"), getattr(m.visual_proj, "1")
            vis = [("visual_proj.0.weight", vp0.weight), ("visual_proj.0.bias", vp0.bias),
                   ("visual_proj.1.weight", vp1.weight), ("visual_proj.1.bias", vp1.bias),
                   ("start_img", m.start_img), ("end_img", m.end_img)]
            if hasattr(m, "start_depth_img"):                  # two-image plugin (llama_ens5_2images.py:343-344)
                vis += [("start_depth_img", m.start_depth_img), ("end_depth_img", m.end_depth_img)]
            if m.args.qformer_tokens:
                q0, q1 = getattr(m.qformer_proj, "0"), getattr(m.qformer_proj, "1")
                vis += [("qformer_proj.0.weight", q0.weight), ("qformer_proj.0.bias", q0.bias),
                        ("qformer_proj.1.weight", q1.weight), ("qformer_proj.1.bias", q1.bias)]
            items.append(("vision_proj", vis))
        # frozen parameters get no gradient storage (LoRA / partial fine-tuning)
        lay = [(b, [(n, q) for n, q in plist if q.requires_grad]) for b, plist in items]
        if self.zero1_world:           # sharded (compute-dtype) parameters first in every bucket: one contiguous span per bucket (stable order)
            lay = [(b, [it for it in pl if it[1].dtype != torch.float32] + [it for it in pl if it[1].dtype == torch.float32]) for b, pl in lay]
        return lay

    def ensure_grads(self):
        """Allocate the flat fp32 gradient buffer once; (re)attach zeroed views where .grad is None."""
        m = self.m
        if self._flat is None:
            total = 0
            offs = {}
            self._ranges = []
            zw = self.zero1_world
            for bucket, plist in self._layout():
                if zw:
                    total = (total + 64 * zw - 1) // (64 * zw) * (64 * zw)              # a bucket's sharded span starts on a 64 N granule
                start = total
                segs = []
                for name, p in plist:
                    if zw and p.dtype == torch.float32 and segs and segs[-1] is not None:
                        total = (total + 64 * zw - 1) // (64 * zw) * (64 * zw)          # end of the bucket's sharded span: whole 64 N granules
                        self._z1.append((bucket, segs))
                        segs = [None]                                                   # (closed)
                    offs[name] = (total, p)
                    if zw and p.dtype != torch.float32:
                        segs.append((total, total + p.numel(), name))
                    total += (p.numel() + 63) // 64 * 64
                if zw and segs and segs[-1] is not None:
                    total = (total + 64 * zw - 1) // (64 * zw) * (64 * zw)
                    self._z1.append((bucket, segs))
                self._ranges.append((bucket, start, total))
            self._flat = torch.zeros(total, dtype=torch.float32, device=m._device)
            for name, (o, p) in offs.items():
                self._views[name] = self._flat[o:o + p.numel()].view(p.shape)
            self._params = {name: p for name, (o, p) in offs.items()}
            self._offs = {name: (o, (p.numel() + 63) // 64 * 64) for name, (o, p) in offs.items()}
            self._gemm_written = {n for n in self._params if n == "output.weight" or (n.startswith("layers.") and n.endswith(
                ("wq.weight", "wk.weight", "wv.weight", "wo.weight", "w1.weight", "w2.weight", "w3.weight")) and "lora_" not in n)}
            if zw:
                # the sharded parameters move into ONE flat buffer in the compute dtype with the gradient buffer's layout (their old
                # storage is released): reduce-scatter / all-gather address it by the same flat ranges as the gradients
                rng = {b: (st, en) for b, st, en in self._ranges}
                z1 = []
                for bucket, segs in self._z1:
                    segs = [x for x in segs if x is not None]
                    ss = segs[0][0]
                    se = (segs[-1][1] + 64 * zw - 1) // (64 * zw) * (64 * zw)
                    assert (se - ss) % (64 * zw) == 0 and se <= rng[bucket][1], (bucket, ss, se, rng[bucket])
                    z1.append((bucket, rng[bucket][0], rng[bucket][1], ss, se, segs))
                self._z1 = z1
                self._flat_params = torch.zeros(total, dtype=self.act, device=m._device)
                with torch.no_grad():
                    for _, _, _, _, _, segs in z1:
                        for a, z, name in segs:
                            q = self._params[name]
                            dst = self._flat_params[a:z].view(q.shape)
                            dst.copy_(q.data)
                            q.data = dst
                if hasattr(m, "invalidate_packed_weights"):
                    m.invalidate_packed_weights()
        runs = []                     # [start, end) float ranges of the flat buffer to zero, adjacent ones merged (one fill per run)
        for name, p in self._params.items():
            if not p.requires_grad:
                if p.grad is not None and p.grad.data_ptr() == self._views[name].data_ptr():
                    p.grad = None
                continue
            v = self._views[name]
            if self.zero1_world and p.dtype != torch.float32:
                # a sharded parameter: its gradient lives in the flat buffer only (torch refuses an fp32 .grad on a bf16 tensor); it
                # restarts after every Zero1Optimizer.step (zero1_mark_fresh)
                if self._z1_fresh:
                    if name in self._gemm_written:
                        self._fresh.add(name)
                    else:
                        o, n = self._offs[name]
                        if runs and runs[-1][1] == o:
                            runs[-1][1] = o + n
                        else:
                            runs.append([o, o + n])
                continue
            if p.grad is None:
                # big decoder / head matrices are written (not accumulated) by their first weight-gradient GEMM of the
                # step: no 27 GB zero-fill and no read of the old value.  Everything else is zeroed here.
                if name in self._gemm_written:
                    self._fresh.add(name)
                else:
                    o, n = self._offs[name]
                    if runs and runs[-1][1] == o:
                        runs[-1][1] = o + n
                    else:
                        runs.append([o, o + n])
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError(f"{name}.grad was replaced by a foreign tensor; use zero_grad(set_to_none=True) or keep the views")
        for o, e in runs:
            self._flat[o:e].zero_()
        self._z1_fresh = False

    # ------------------------------------------------------------------ ZeRO-1 (a3vlm_amd.zero1.Zero1Optimizer)
    def flat_params(self) -> torch.Tensor:
        if self._flat is None:
            self.ensure_grads()
        if self._flat_params is None:
            raise RuntimeError("flat_params(): the engine was not built with zero1_world")
        return self._flat_params

    def zero1_buckets(self, weight_decay: float = 0.0):
        """[(bucket, start, end, shard_start, shard_end, [(seg_start, seg_end, weight_decay)])]: per-parameter pieces of each bucket's sharded
        span with the reference's decay rule (no decay for *.bias / *norm.weight, util/misc.py:586-599: none of the sharded matrices)."""
        self.flat_params()
        return [(b, s, e, ss, se, [(a, z, 0.0 if (n.endswith(".bias") or n.endswith("norm.weight")) else weight_decay) for a, z, n in segs])
                for b, s, e, ss, se, segs in self._z1]

    def zero1_mark_fresh(self) -> None:
        """The sharded parameters were updated in place (all-gather into flat_params): their gradients restart at the next backward and
        every cache keyed on parameter state (packed inference weights) is stale; the training images are views and need nothing."""
        self._z1_fresh = True
        for _, _, _, _, _, segs in self._z1:
            for _, _, name in segs:
                torch.autograd.graph.increment_version(self._params[name])

    def _pview(self, ps) -> Optional[torch.Tensor]:
        """[rows, cols] view of consecutive sharded parameters (wq|wk|wv, w1|w3, or one matrix) in the flat parameter buffer, or None."""
        fp = self._flat_params
        if fp is None or any(q.dtype != fp.dtype or q.untyped_storage().data_ptr() != fp.untyped_storage().data_ptr() for q in ps):
            return None
        es = fp.element_size()
        o = (ps[0].data_ptr() - fp.data_ptr()) // es
        n, cols = sum(q.numel() for q in ps), ps[0].shape[1]
        for a, b2 in zip(ps, ps[1:]):
            if b2.data_ptr() != a.data_ptr() + a.numel() * es or b2.shape[1] != cols:
                return None
        return fp[o:o + n].view(n // cols, cols)

    def grad_ranges(self):
        return list(self._ranges)

    def flat_grads(self) -> torch.Tensor:
        return self._flat

    def _has(self, *names: str) -> bool:
        return all(n in self._views for n in names)

    def _gview(self, first: str, last: str) -> torch.Tensor:
        """Fused [rows, cols] view over consecutive parameters (wq|wk|wv, w1|w3)."""
        a, b = self._views[first], self._views[last]
        cols = a.shape[1]
        n = (b.data_ptr() - a.data_ptr()) // 4 + b.numel()
        assert n % cols == 0
        base = self._flat[(a.data_ptr() - self._flat.data_ptr()) // 4:]
        return base[:n].view(n // cols, cols)

    # ------------------------------------------------------------------ weight images in the compute dtype
    def _images(self):
        """Lazy bf16 weight images: group ``L{i}`` (qkv / wo / w13 / w2 and their transposes of layer i), ``out``, ``vp``; a group
        is (re)built on first use after its parameters changed.  (Issuing the AdamW update per layer on a side stream under the
        next forward was tried on top of this and measured no gain: the 256x256 GEMM leaves no VGPRs for a co-resident kernel.)"""
        if self._img is None or not isinstance(self._img, _Images):
            self._img = _Images(self)
        return self._img

    # ------------------------------------------------------------------ GEMM helpers
    def _wgrad(self, dy: torch.Tensor, x: torch.Tensor, grad: torch.Tensor, tag: str, names=(), store: bool = False):
        """grad[N,K] (fp32) += dy[M,N]^T @ x[M,K]; plain store when every parameter in ``names`` is fresh this step (or ``store``)."""
        M, N = dy.shape
        K = x.shape[1]
        fresh = store or (bool(names) and all(n in self._fresh for n in names))
        assert fresh or not any(n in self._fresh for n in names), "partially fresh fused gradient view"
        tn_ok = (self.tn_wgrad and self.act == torch.bfloat16 and N % 8 == 0 and K % 4 == 0 and dy.stride(0) % 8 == 0
                 and x.stride(0) % 8 == 0 and 2 * M * max(dy.stride(0), x.stride(0)) < 2 ** 31)
        if tn_ok and min(N, K) >= 256:
            # both operands as they are (token-major): the TN kernel transposes fragments on the LDS read
            self._fresh.difference_update(names if fresh else ())
            sink = self.sumsq_sink            # dp.GradSquareSums (one rank): the clip's sum of squares comes out of this GEMM's epilogue
            sq = sink.wgrad_slots(grad, N, K) if sink is not None and grad.is_contiguous() else None
            ops.gemm_tn(dy, x, grad, residual=None if fresh else grad, epilogue=ops.EPI_OUT_F32 if fresh else ops.EPI_RES_F32, sumsq=sq)
            return
        if (tn_ok and self.strip_wgrad and N <= 64 and N % 4 == 0 and K >= 256 and dy.stride(0) >= 64
                and (dy.data_ptr() | x.data_ptr()) % 16 == 0):
            # adapter gradients [N <= 64, K]: the small-block streaming kernel (three blocks per CU; token slices chosen so that the
            # grid holds ~2 blocks per CU: fewer slices = less reduce work, tools/lora_skinny_bench.py)
            tiles = (K + 127) // 128
            S = max(1, min(64, (M + 63) // 64, -(-512 // tiles)))
            self._fresh.difference_update(names if fresh else ())
            ops.gemm_tn_strip(dy, x, grad, self._buf("splitk", (S * N * K,), torch.float32), S, accumulate=not fresh)
            return
        if tn_ok and min(N, K) <= 64 and max(N, K) >= 256:
            # adapter gradients: a strip of 256 x 256 tiles, split over the tokens so the whole chip streams dy / x once
            blocks = ((N + 255) // 256) * ((K + 255) // 256)
            S = 1
            while blocks * S < 256 and S < 16 and 2 * S <= (M + 63) // 64:
                S *= 2
            self._fresh.difference_update(names if fresh else ())
            ops.gemm_tn_splitk(dy, x, grad, self._buf("splitk", (S * N * K,), torch.float32), S, accumulate=not fresh)
            return
        Mp = _pad64(M)
        dyt = self._buf("wg_dyt" + tag, (N, Mp))
        xt = self._buf("wg_xt" + tag, (K, Mp))
        ops.transpose(dy, dyt, M, N, Mp)
        ops.transpose(x, xt, M, K, Mp)
        if fresh:
            self._fresh.difference_update(names)
            if min(N, K) <= 64:
                self._skinny(dyt, xt, grad)
            else:
                ops.gemm_nt(dyt, xt, grad, epilogue=ops.EPI_OUT_F32 if self.act == torch.bfloat16 else 0)
        else:
            if min(N, K) <= 64:
                self._skinny(dyt, xt, grad, accumulate=True)
            else:
                ops.gemm_nt(dyt, xt, grad, residual=grad, epilogue=ops.EPI_RES_F32 if self.act == torch.bfloat16 else 0)

    def _skinny(self, a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, accumulate: bool = False):
        """out (+)= a @ w.T for adapter-sized problems (M or N <= 64, long K): split-K planes of the 128x128 kernel + one
        reduce pass; a plain launch would be a handful of blocks with a serial K loop (122 us instead of ~25 at 8728 x 64 x 4096)."""
        M, K = a.shape
        N = w.shape[0]
        S = skinny_slices(M, N, K, self._cus(a.device) if a.is_cuda else 0)
        if S == 1 or self.act != torch.bfloat16 or K % 64:
            f32 = out.dtype == torch.float32 and self.act == torch.bfloat16
            ops.gemm_nt(a, w, out, residual=out if accumulate else None,
                        epilogue=(ops.EPI_RES_F32 if accumulate else ops.EPI_OUT_F32) if f32 else 0)
            return
        ops.gemm_nt_splitk(a, w, out, self._buf("splitk", (S * M * N,), torch.float32), S, accumulate)

    def _cus(self, dev) -> int:
        n = getattr(self, "_cu_count", None)
        if n is None:
            n = self._cu_count = int(torch.cuda.get_device_properties(dev).multi_processor_count)
        return n

    def _dgrad_w(self, dy: torch.Tensor, key: str, out: torch.Tensor):
        """out[M,K] = dy[M,N] @ W[N,K] for the weight image ``key``: the NN kernel on the forward image itself (the transposed
        image ``key.t`` is then never built: no per-step transpose of the updated weights, no second copy in HBM); falls back
        to the NT kernel on the transposed image for shapes / dtypes the NN kernel does not take."""
        im = self._images()
        M, N = dy.shape
        if self.nn_dgrad and self.act == torch.bfloat16 and N % 64 == 0 and dy.stride(0) % 8 == 0:
            w = im[key]
            if w.shape[0] == N and w.shape[1] % 8 == 0 and w.shape[1] >= 256 and 2 * M * dy.stride(0) < 2 ** 31 and 2 * w.numel() < 2 ** 31:
                ops.gemm_nn(dy, w, out)
                return
        self._dgrad(dy, im[key + ".t"], out)

    def _dgrad(self, dy: torch.Tensor, wt: torch.Tensor, out: torch.Tensor):
        """out[M,K] = dy[M,N] @ W[N,K] with wt = W^T [K, Np]; dy may have N < Np columns -> padded copy."""
        M, N = dy.shape
        Np = wt.shape[1]
        if Np != N:
            dyp = self._buf("dg_pad", (M, Np), zero=True)
            dyp[:, :N].copy_(dy)
            dy = dyp
        if wt.shape[0] <= 64:
            self._skinny(dy, wt, out)
        else:
            ops.gemm_nt(dy, wt, out)

    # ------------------------------------------------------------------ LoRA adapters (model/peft.py semantics)
    def _lora_step_images(self):
        """A [Rp,in], B [N,Rp] (+ transposes for the backward GEMMs) of every adapter group in the compute dtype.  Built once;
        after an optimizer step only the r rows / columns each adapter owns are re-written in place (one a3v_lora_refresh
        launch per adapter) -- rebuilding the padded / block-diagonal images from scratch was ~2500 small launches per step."""
        m = self.m
        if getattr(self, "_li_checked", False) and self._li_steps == optimizer_steps():
            return self._li                                # validated for this forward / backward (flag reset at their entry), no step since
        self._li_steps = optimizer_steps()
        if getattr(self, "_lora_plist", None) is None:
            self._lora_plist = [q for n, q in m.named_parameters() if "lora_" in n]
        ver = tuple(param_state_key(q) for q in self._lora_plist)
        if getattr(self, "_li_ver", None) == ver:
            self._li_checked = True
            return self._li
        li = getattr(self, "_li", None)
        if li is not None and getattr(self, "_li_act", None) == self.act:
            r = m.lora_rank
            with torch.no_grad():
                for i in range(m.n_layers):
                    for key, mods, _ in m.lora_groups(i):
                        A, Bm, At, Bt = li[key + ".A"], li[key + ".B"], li[key + ".At"], li[key + ".Bt"]
                        row = 0
                        for j, mod in enumerate(mods):
                            wa, wb = mod.lora_a.weight, mod.lora_b.weight          # [r, in], [N_j, r]
                            nj = wb.shape[0]
                            if self.act == torch.bfloat16 and wa.dtype == torch.float32 and wa.is_contiguous() and wb.is_contiguous():
                                ops.lora_refresh(wa, wb, A, At, Bm, Bt, j * r, row)
                            else:
                                A[j * r:(j + 1) * r].copy_(wa)
                                At[:wa.shape[1], j * r:(j + 1) * r].copy_(wa.t())
                                Bm[row:row + nj, j * r:(j + 1) * r].copy_(wb)
                                Bt[j * r:(j + 1) * r, row:row + nj].copy_(wb.t())
                            row += nj
            self._li_ver = ver
            self._li_checked = True
            return li
        src = m.lora_images(dtype=self.act, interleave_w13=False)
        li = {}
        for k, v in src.items():
            li[k] = v.clone()              # the engine's own copies (updated in place from now on)
            R, C = v.shape
            vt = torch.empty(C, _pad64(R), dtype=v.dtype, device=v.device)
            if _pad64(R) != R:
                vp = torch.zeros(_pad64(R), C, dtype=v.dtype, device=v.device)
                vp[:R] = v
            else:
                vp = v
            ops.transpose(vp, vt, _pad64(R), C, _pad64(R))
            li[k + "t"] = vt
        if self._kext():
            im = self._images()
            for i in range(m.n_layers):
                for key, _, _ in m.lora_groups(i):
                    fx, fy = im[key + ".x"], im[key + ".y"]   # [N, K + Rp] = [W | B],  [N + Rp, K] = [W ; A]
                    Rp = li[key + ".A"].shape[0]
                    blk = fx[:, fx.shape[1] - Rp:]
                    blk.copy_(li[key + ".B"])
                    li[key + ".B"] = blk                        # refreshed in place from now on (row stride K + Rp)
                    rows_a = fy[fy.shape[0] - Rp:]
                    rows_a.copy_(li[key + ".A"])
                    li[key + ".A"] = rows_a
                    if self._nt_dgrad(key):
                        ft = im[key + ".yT"]                      # [K, N + Rp]: the A^T block lives in its tail columns
                        cols_at = ft[:, ft.shape[1] - Rp:]
                        cols_at.copy_(li[key + ".At"][:cols_at.shape[0]])
                        li[key + ".At"] = cols_at
        self._li, self._li_ver, self._li_act = li, ver, self.act
        self._li_checked = True
        self._adapter_sinks = None
        return li

    def adapter_sink(self, p: torch.Tensor):
        """For ``FusedAdamW``'s multi-tensor launch: where the bf16 value of element (i, j) of adapter parameter ``p`` lives inside
        the fused group images, as (d1, s1r, s1c, d2, s2r, s2c) in element strides -- lora_a [r, in]: its rows of A and its columns
        of A^T; lora_b [n_j, r]: its columns of B and its rows of B^T.  None when ``p`` is not an adapter or the images are not
        built yet (first step: a3v_lora_refresh does it)."""
        li = getattr(self, "_li", None)
        if li is None or self.act != torch.bfloat16 or getattr(self, "_li_act", None) != self.act:
            return None
        sinks = getattr(self, "_adapter_sinks", None)
        if sinks is None:
            sinks = {}
            r = self.m.lora_rank
            for i in range(self.m.n_layers):
                for key, mods, _ in self.m.lora_groups(i):
                    A, Bm, At, Bt = li[key + ".A"], li[key + ".B"], li[key + ".At"], li[key + ".Bt"]
                    row = 0
                    for j, mod in enumerate(mods):
                        wa, wb = mod.lora_a.weight, mod.lora_b.weight
                        sinks[id(wa)] = (A.data_ptr() + 2 * (j * r) * A.stride(0), A.stride(0), 1,
                                         At.data_ptr() + 2 * (j * r), 1, At.stride(0))
                        sinks[id(wb)] = (Bm.data_ptr() + 2 * (row * Bm.stride(0) + j * r), Bm.stride(0), 1,
                                         Bt.data_ptr() + 2 * ((j * r) * Bt.stride(0) + row), 1, Bt.stride(0))
                        row += wb.shape[0]
            self._adapter_sinks = sinks
        return sinks.get(id(p))

    def adapters_adopted(self, written_ids) -> None:
        """The optimizer wrote every adapter's bf16 values into the group images itself: they are current for the new parameters."""
        plist = getattr(self, "_lora_plist", None)
        if plist is None or getattr(self, "_li", None) is None:
            return
        if all(id(q) in written_ids for q in plist if q.requires_grad):
            self._li_ver = tuple(param_state_key(q) for q in plist)

    def _lora_fwd(self, key: str, x: torch.Tensor, y: torch.Tensor, tag: str) -> torch.Tensor:
        """y += lora_b(lora_a(x)) for a fused adapter group; returns t = lora_a(x) (kept for the backward)."""
        li = self._lora_step_images()
        A, Bm = li[key + ".A"], li[key + ".B"]
        t = self._buf("lora_t." + key.split(".")[0] + tag, (x.shape[0], A.shape[0]))
        self._skinny(x, A, t)
        f32 = y.dtype == torch.float32 and self.act == torch.bfloat16
        ops.gemm_nt(t, Bm, y, residual=y, epilogue=ops.EPI_RES_F32 if f32 else 0)
        return t

    def _kx_group_bwd(self, i: int, key: str, dy_full: torch.Tensor, N: int, x: torch.Tensor, t: torch.Tensor, dx: torch.Tensor,
                      swiglu_gu: Optional[torch.Tensor] = None):
        """Input gradient of a decoder GEMM with the adapters inside it: dy_full = [dy | dt] (dt = dy . B in the tail columns)
        times [W ; A] in ONE NN GEMM (dx = dy W + dt A), then the adapter weight gradients.  ``swiglu_gu`` (w2 only): the product is
        d(act) of the SwiGLU and ``dx`` the [rows, 2 F] gradient of the gate | up rows -- the SwiGLU backward runs in the GEMM epilogue."""
        li = self._lora_step_images()
        dy, dt = dy_full[:, :N], dy_full[:, N:]
        self._dgrad(dy, li[key + ".Bt"], dt)
        im = self._images()
        if swiglu_gu is not None:
            ops.gemm_nt(dy_full, im[key + ".yT"], dx, residual=swiglu_gu, epilogue=ops.EPI_SWIGLU_BWD)
        elif self._nt_dgrad(key):
            ops.gemm_nt(dy_full, im[key + ".yT"], dx)          # frozen base: the transposed image is free (built once)
        else:
            ops.gemm_nn(dy_full, im[key + ".y"], dx)
        self._lora_bwd(i, key, dy, x, t, None, dt=dt)

    def _lora_bwd(self, i: int, key: str, dy: torch.Tensor, x: torch.Tensor, t: torch.Tensor, dx: Optional[torch.Tensor], dt=None):
        """Adapter gradients of a fused group and the adapter term of the input gradient (dx += (dy B) A)."""
        m = self.m
        li = self._lora_step_images()
        r = m.lora_rank
        Bt, At = li[key + ".Bt"], li[key + ".At"]          # [Rp, Npad], [in, Rp]
        M, N = dy.shape
        Rp = t.shape[1]
        if dt is None:
            dt = self._buf("lora_dt", (M, Rp))
            self._dgrad(dy, Bt, dt)                          # dt = dy @ B
        g = key.split(".")[0]
        # dB is produced transposed ([Rp, N] = t^T dy): the strip with the adapter rank as its ROW index streams dy 1.2-1.5x
        # faster through the TN kernel than the [N, Rp] strip (tools/lora_skinny_bench.py); the r x N_j blocks are added
        # into the [N_j, r] gradient views below
        names = {"qkv": ["attention.wq", "attention.wk", "attention.wv"], "wo": ["attention.wo"],
                 "w13": ["feed_forward.w1", "feed_forward.w3"], "w2": ["feed_forward.w2"]}[g]
        vas = [self._views[f"layers.{i}.{nm}.lora_a.weight"] for nm in names]
        vbs = [self._views[f"layers.{i}.{nm}.lora_b.weight"] for nm in names]
        nr = len(names) * r
        gBt = self._buf("lora_gBt." + g, (Rp, N), torch.float32)
        self._wgrad(t, dy, gBt, "lb", store=True)
        # dA of the whole group straight into the flat gradient buffer: the group's lora_a gradients are adjacent = ONE [n r, in] matrix
        adjacent = all(vas[j + 1].data_ptr() == vas[j].data_ptr() + 4 * vas[j].numel() for j in range(len(vas) - 1))
        if adjacent and dt.stride(0) % 8 == 0:
            off = (vas[0].data_ptr() - self._flat.data_ptr()) // 4
            self._wgrad(dt[:, :nr], x, self._flat[off:off + nr * x.shape[1]].view(nr, x.shape[1]), "la", store=False)
        else:
            gA = self._buf("lora_gA." + g, (Rp, x.shape[1]), torch.float32)
            self._wgrad(dt, x, gA, "la", store=True)
            for j, va in enumerate(vas):
                ops.add2d(va, gA[j * r:(j + 1) * r])
        row0s, row = [], 0
        for vb in vbs:
            row0s.append(row)
            row += vb.shape[0]
        if gBt.is_cuda:
            ops.lora_gb_scatter(gBt, r, vbs, row0s)          # one launch: each module's [n_j, r] += its diagonal block transposed
        else:
            for j, vb in enumerate(vbs):
                vb.add_(gBt[j * r:(j + 1) * r, row0s[j]:row0s[j] + vb.shape[0]].t())
        if dx is not None:
            ops.gemm_nt(dt, At, dx, residual=dx)             # dx += dt @ A

    # ------------------------------------------------------------------ one decoder block (forward / recompute)
    def _block_forward(self, i: int, h: torch.Tensor, B: int, S: int, keep: bool, tag: str = "", h_out: Optional[torch.Tensor] = None):
        """One decoder block on the fp32 stream h [B*S, dim].
        keep=False: h is updated in place (checkpointing forward).
        keep=True : h is only read; h_mid goes to its own buffer and the intermediates the backward needs are
        returned; the block output is written to ``h_out`` if given (stored-activation forward, per-layer buffers
        selected by ``tag``) or not computed at all (recompute inside backward)."""
        m, a, im = self.m, self.m.args, self._images()
        H, Hkv, hd, dim, F = m.n_heads, m.n_kv_heads, m.head_dim, a.dim, m.ffn
        rows = B * S
        l = m.layers[i]
        spad = _pad64(S)
        kx = self._kext() > 0 and rows > 16 and hd in (64, 128) and self.fuse_qkv_rope
        li = self._lora_step_images() if kx else None

        def xbuf(name, cols, key):      # activation buffer with the adapter block t = x . A^T appended (K_ext): full, x view, t view
            ext = li[key + ".A"].shape[0]
            full = self._buf(name + tag, (rows, cols + ext))
            return full, full[:, :cols], full[:, cols:]
        if kx:
            xn_full, xn, t_qkv = xbuf("xn", dim, f"qkv.{i}")
            att_full, att, t_wo = xbuf("att", H * hd, f"wo.{i}")
        else:
            xn = self._buf("xn" + tag, (rows, dim))
            att = self._buf("att" + tag, (rows, H * hd))
        qkv = self._buf("qkv" + tag, (rows, (H + 2 * Hkv) * hd))
        qrot = self._buf("qrot" + tag, (rows, H * hd))
        kc = self._buf("kc" + tag, (B, Hkv, spad, hd))
        vc = self._buf("vc", (B, Hkv, hd, spad))
        lse = self._buf("lse" + tag, (B, H, S), torch.float32)
        ops.rmsnorm(h, l.attention_norm.weight, xn, a.norm_eps)
        lt = {}
        if not self.lora and self.act == torch.bfloat16 and hd in (64, 128) and rows > 16 and self.fuse_qkv_rope:
            # RoPE + K / V^T writes in the GEMM epilogue; v also token-major in its qkv columns for the attention backward
            ops.gemm_qkv_rope(xn, im[f"qkv.{i}"], qrot, kc, vc, m._cos_sin_dev(), B, S, H, Hkv, hd, 0, 0,
                              v_rows=qkv[:, (H + Hkv) * hd:])
        elif kx:
            # adapters inside the GEMM: t = x . A^T goes into the tail columns of the input, B sits in the tail columns of the image
            self._skinny(xn, li[f"qkv.{i}.A"], t_qkv)
            ops.gemm_qkv_rope(xn_full, im[f"qkv.{i}.x"], qrot, kc, vc, m._cos_sin_dev(), B, S, H, Hkv, hd, 0, 0,
                              v_rows=qkv[:, (H + Hkv) * hd:])
            lt["qkv"] = t_qkv
        elif self.lora and self.act == torch.bfloat16 and hd in (64, 128) and rows > 16 and self.fuse_qkv_rope:
            # adapters: the LoRA term t . B^T is written into the qkv buffer first and enters the fused GEMM as an additive term
            # before the rotation (the reference's order: linear output + modification, rounded, then RoPE)
            li = self._lora_step_images()
            A, Bm = li[f"qkv.{i}.A"], li[f"qkv.{i}.B"]
            t = self._buf("lora_t.qkv" + tag, (rows, A.shape[0]))
            self._skinny(xn, A, t)
            ops.gemm_nt(t, Bm, qkv)
            ops.gemm_qkv_rope(xn, im[f"qkv.{i}"], qrot, kc, vc, m._cos_sin_dev(), B, S, H, Hkv, hd, 0, 0,
                              v_rows=qkv[:, (H + Hkv) * hd:], delta=qkv)
            lt["qkv"] = t
        else:
            ops.gemm_nt(xn, im[f"qkv.{i}"], qkv)
            if self.lora:
                lt["qkv"] = self._lora_fwd(f"qkv.{i}", xn, qkv, tag)
            ops.rope_kvcache(qkv, qrot, kc, vc, m._cos_sin_dev(), B, S, H, Hkv, hd, 0, 0)
        ldo = att.stride(0)
        strides = (S * H * hd, H * hd, hd, Hkv * spad * hd, spad * hd, hd, Hkv * hd * spad, hd * spad, spad, S * ldo, ldo, hd)
        ops.attention_lse(qrot, kc, vc, att, lse, B, S, S, H, Hkv, hd, strides, True)
        res_flag = ops.EPI_RES_F32 if (self.act == torch.bfloat16 and self.stream == torch.float32) else 0     # bf16 stream: plain bf16 residual
        h_mid = self._buf("h_mid" + tag, (rows, dim), self.stream) if keep else h
        if kx:
            self._skinny(att, li[f"wo.{i}.A"], t_wo)
            ops.gemm_nt(att_full, im[f"wo.{i}.x"], h_mid, residual=h, epilogue=res_flag)
            lt["wo"] = t_wo
        else:
            ops.gemm_nt(att, im[f"wo.{i}"], h_mid, residual=h, epilogue=res_flag)
            if self.lora:
                lt["wo"] = self._lora_fwd(f"wo.{i}", att, h_mid, tag)
        gu = self._buf("gu" + tag, (rows, 2 * F))
        if kx:
            xn2_full, xn2, t_w13 = xbuf("xn2", dim, f"w13.{i}")
            act_full, actb, t_w2 = xbuf("act", F, f"w2.{i}")
        else:
            xn2 = self._buf("xn2" + tag, (rows, dim))
            actb = self._buf("act" + tag, (rows, F))
        ops.rmsnorm(h_mid, l.ffn_norm.weight, xn2, a.norm_eps)
        if kx:
            self._skinny(xn2, li[f"w13.{i}.A"], t_w13)
            ops.gemm_nt(xn2_full, im[f"w13.{i}.x"], gu)
            lt["w13"] = t_w13
        else:
            ops.gemm_nt(xn2, im[f"w13.{i}"], gu)
            if self.lora:
                lt["w13"] = self._lora_fwd(f"w13.{i}", xn2, gu, tag)
        ops.swiglu_fwd(gu, actb, F, interleaved=False)
        if kx:
            self._skinny(actb, li[f"w2.{i}.A"], t_w2)
            lt["w2"] = t_w2
        elif self.lora:                                   # t = lora_a(act) of w2 is needed by the backward even without the output
            li2 = self._lora_step_images()
            lt["w2"] = self._buf("lora_t.w2" + tag, (rows, li2[f"w2.{i}.A"].shape[0]))
            self._skinny(actb, li2[f"w2.{i}.A"], lt["w2"])
        kept = dict(xn=xn, qkv=qkv, qrot=qrot, kc=kc, att=att, lse=lse, h_mid=h_mid, xn2=xn2, gu=gu, act=actb, spad=spad, lt=lt)
        if keep and h_out is None:
            return kept                                   # recompute inside backward: the block output is not needed
        out = h_out if keep else h
        if kx:
            ops.gemm_nt(act_full, im[f"w2.{i}.x"], out, residual=h_mid, epilogue=res_flag)
            return kept
        ops.gemm_nt(actb, im[f"w2.{i}"], out, residual=h_mid, epilogue=res_flag)
        if self.lora:
            f32 = out.dtype == torch.float32 and self.act == torch.bfloat16
            ops.gemm_nt(lt["w2"], self._lora_step_images()[f"w2.{i}.B"], out, residual=out, epilogue=ops.EPI_RES_F32 if f32 else 0)
        return kept

    def _block_backward(self, i: int, h_in: torch.Tensor, dh: torch.Tensor, B: int, S: int):
        m, a, im = self.m, self.m.args, self._images()
        H, Hkv, hd, dim, F = m.n_heads, m.n_kv_heads, m.head_dim, a.dim, m.ffn
        rows = B * S
        l = m.layers[i]
        pre = f"layers.{i}."
        k = self._saved["kept"][i] if self._saved.get("kept") else self._block_forward(i, h_in, B, S, keep=True)
        # ---- FFN: out = h_mid + w2(silu(g) * u)
        lt = k.get("lt", {})
        kx = self._kext() > 0 and rows > 16 and hd in (64, 128) and self.fuse_qkv_rope       # as in _block_forward
        li = self._lora_step_images() if kx else None

        def ybuf(name, cols, key):      # gradient buffer with room for dt = dy . B behind it: full, dy view
            full = self._buf(name, (rows, cols + li[key + ".A"].shape[0]))
            return full, full[:, :cols]
        stream_lp = self.stream == torch.bfloat16 and self.act == torch.bfloat16     # dh itself is the bf16 operand of the GEMMs
        if stream_lp:
            dha = dh
            dha_full = self._dh_full if kx else None
        elif kx:
            dha_full, dha = ybuf("dh_act.x", dim, f"w2.{i}")
        else:
            dha = self._buf("dh_act", (rows, dim))
        fuse_cast = self.act == torch.bfloat16 and not stream_lp   # fp32 stream: the norm backward that produced dh also wrote its bf16 copy
        if not stream_lp and not (fuse_cast and getattr(self, "_dha_ready", False)):
            ops.cast(dh, dha)
        self._dha_ready = False
        if self._has(pre + "feed_forward.w2.weight"):
            self._wgrad(dha, k["act"], self._views[pre + "feed_forward.w2.weight"], "w2", (pre + "feed_forward.w2.weight",))
        if kx and self.fuse_swiglu_bwd and self._nt_dgrad(f"w2.{i}") and F % 8 == 0:
            # d(act) never reaches HBM: the input-gradient GEMM of w2 applies the SwiGLU backward in its epilogue (a3v_swiglu_bwd on
            # the bf16-rounded product, bit for bit) and writes d(gate) | d(up) straight into the w1|w3 gradient buffer
            dgu_full, dgu = ybuf("dgu.x", 2 * F, f"w13.{i}")
            self._kx_group_bwd(i, f"w2.{i}", dha_full, dim, k["act"], lt["w2"], dgu, swiglu_gu=k["gu"])
        elif (not self.lora and self.fuse_swiglu_bwd and self.nn_dgrad and self.act == torch.bfloat16 and dim % 64 == 0 and F % 8 == 0 and F >= 256
              and dha.stride(0) % 8 == 0 and 2 * rows * dha.stride(0) < 2 ** 31 and 2 * F * dim < 2 ** 31
              and tuple(im[f"w2.{i}"].shape) == (dim, F)):
            # full fine-tune: the same epilogue on the NN kernel (dX = dY . W on the forward image)
            dgu = self._buf("dgu", (rows, 2 * F))
            ops.gemm_nn(dha, im[f"w2.{i}"], dgu, residual=k["gu"], epilogue=ops.EPI_SWIGLU_BWD)
        elif (not self.lora and self.fuse_swiglu_bwd and not self.nn_dgrad and self.act == torch.bfloat16 and dim % 64 == 0 and F % 8 == 0
              and dha.stride(0) % 8 == 0 and tuple(im[f"w2.{i}.t"].shape) == (F, dim)):
            # full fine-tune with input gradients on the NT ring kernel (W^T images): the same epilogue there
            dgu = self._buf("dgu", (rows, 2 * F))
            ops.gemm_nt(dha, im[f"w2.{i}.t"], dgu, residual=k["gu"], epilogue=ops.EPI_SWIGLU_BWD)
        else:
            dact = self._buf("dact", (rows, F))
            if kx:
                self._kx_group_bwd(i, f"w2.{i}", dha_full, dim, k["act"], lt["w2"], dact)
                dgu_full, dgu = ybuf("dgu.x", 2 * F, f"w13.{i}")
            else:
                self._dgrad_w(dha, f"w2.{i}", dact)
                if self.lora:
                    self._lora_bwd(i, f"w2.{i}", dha, k["act"], lt["w2"], dact)
                dgu = self._buf("dgu", (rows, 2 * F))
            ops.swiglu_bwd(k["gu"], dact, dgu, F, interleaved=False)
        if self._has(pre + "feed_forward.w1.weight", pre + "feed_forward.w3.weight"):
            self._wgrad(dgu, k["xn2"], self._gview(pre + "feed_forward.w1.weight", pre + "feed_forward.w3.weight"), "w13",
                        (pre + "feed_forward.w1.weight", pre + "feed_forward.w3.weight"))
        dxn = self._buf("dxn", (rows, dim))
        if kx:
            self._kx_group_bwd(i, f"w13.{i}", dgu_full, 2 * F, k["xn2"], lt["w13"], dxn)
        else:
            self._dgrad_w(dgu, f"w13.{i}", dxn)
            if self.lora:
                self._lora_bwd(i, f"w13.{i}", dgu, k["xn2"], lt["w13"], dxn)
        ops.rmsnorm_bwd(k["h_mid"], l.ffn_norm.weight, dxn, dh, self._views.get(pre + "ffn_norm.weight"), a.norm_eps,
                        dh_lowp=dha if fuse_cast else None)
        # ---- attention: h_mid = h_in + wo(attn(rope(qkv(norm(h_in)))))
        if not fuse_cast and not stream_lp:
            ops.cast(dh, dha)
        if self._has(pre + "attention.wo.weight"):
            self._wgrad(dha, k["att"], self._views[pre + "attention.wo.weight"], "wo", (pre + "attention.wo.weight",))
        datt = self._buf("datt", (rows, H * hd))
        if kx:
            self._kx_group_bwd(i, f"wo.{i}", dha_full, dim, k["att"], lt["wo"], datt)
        else:
            self._dgrad_w(dha, f"wo.{i}", datt)
            if self.lora:
                self._lora_bwd(i, f"wo.{i}", dha, k["att"], lt["wo"], datt)
        D = self._buf("attn_D", (B, S, H), torch.float32)
        qkv = k["qkv"]
        ld = qkv.stride(0)
        vrows = qkv[:, (H + Hkv) * hd:]
        spad = k["spad"]
        if kx:
            dqkv_full, dqkv = ybuf("dqkv.x", (H + 2 * Hkv) * hd, f"qkv.{i}")
        else:
            dqkv = self._buf("dqkv", (rows, (H + 2 * Hkv) * hd))
        if self.act == torch.bfloat16 and hd in (64, 128) and self.packed_attn_bwd:
            # the MFMA kernels rotate dq / dk back and store [dq | dk | dv] straight into the fused-qkv gradient
            ops.attention_bwd_packed(k["qrot"], k["kc"], Hkv * spad * hd, spad * hd, vrows, S * ld, ld, hd, k["att"], datt, k["lse"], D,
                                     dqkv, m._cos_sin_dev(), B, S, H, Hkv, hd, True, 0)
        else:
            dq = self._buf("dq", (rows, H * hd))
            dk = self._buf("dk", (B, Hkv, S, hd))
            dv = self._buf("dv", (B, Hkv, S, hd))
            wsp = None
            if self.act == torch.bfloat16 and hd in (64, 128):
                wsp = self._buf("attn_bwd_ws", (ops.attention_bwd_workspace_bytes(B, S, H, Hkv, hd),), torch.uint8)
            att_c = k["att"] if k["att"].is_contiguous() else k["att"].contiguous()      # (K_ext keeps att inside a wider buffer)
            ops.attention_bwd(k["qrot"], k["kc"], Hkv * spad * hd, spad * hd, vrows, S * ld, ld, hd, att_c, datt, k["lse"], D,
                              dq, dk, dv, B, S, H, Hkv, hd, True, workspace=wsp)
            ops.rope_bwd_pack(dq, dk, dv, dqkv, m._cos_sin_dev(), B, S, H, Hkv, hd, 0)
        if self._has(pre + "attention.wq.weight", pre + "attention.wk.weight", pre + "attention.wv.weight"):
            self._wgrad(dqkv, k["xn"], self._gview(pre + "attention.wq.weight", pre + "attention.wv.weight"), "qkv",
                        (pre + "attention.wq.weight", pre + "attention.wk.weight", pre + "attention.wv.weight"))
        if kx:
            self._kx_group_bwd(i, f"qkv.{i}", dqkv_full, (H + 2 * Hkv) * hd, k["xn"], lt["qkv"], dxn)
        else:
            self._dgrad_w(dqkv, f"qkv.{i}", dxn)
            if self.lora:
                self._lora_bwd(i, f"qkv.{i}", dqkv, k["xn"], lt["qkv"], dxn)
        ops.rmsnorm_bwd(h_in, l.attention_norm.weight, dxn, dh, self._views.get(pre + "attention_norm.weight"), a.norm_eps,
                        dh_lowp=dha if fuse_cast else None)
        self._dha_ready = fuse_cast                        # the next block's backward finds its bf16 operand in place

    # ------------------------------------------------------------------ forward (loss) and backward
    @torch.no_grad()
    def forward_loss(self, examples: torch.Tensor, labels: torch.Tensor, image: Optional[torch.Tensor] = None,
                     qformer_feats=None, extra_feats=None) -> torch.Tensor:
        self._check_dtypes()
        self._li_checked = False
        m, a = self.m, self.m.args
        im = self._images()
        B, T = examples.shape
        n_img = 0 if image is None else (len(image) if isinstance(image, (list, tuple)) else 1)
        W = m.words_per_image * n_img if n_img else 0
        S = T + W
        rows = B * S
        dim, V = a.dim, a.vocab_size
        h = self._buf("h", (rows, dim), self.stream)
        if self.lora or self.act != torch.bfloat16:
            self.sync_optimizer()          # adapter / fp32 images are rebuilt from all parameters at once
        self.await_weights("embed")
        ops.embed_assemble(examples.contiguous(), m.tok_embeddings.weight, h, B, T, W, dim)
        vis = None
        if image is not None:
            self.await_weights("vision_proj")
            vis = self._encode_image_train(h, image, B, S, qformer_feats, extra_feats)
        hs = self._buf("h_saved", (m.n_layers + 1, rows, dim), self.stream)
        if self.recompute is None:
            F_, Hq = m.ffn, (m.n_heads + 2 * m.n_kv_heads) * m.head_dim
            esz = 2 if self.act == torch.bfloat16 else 4
            need = m.n_layers * rows * ((2 * dim + Hq + 2 * m.n_heads * m.head_dim + 3 * F_) * esz + dim * (4 if self.stream == torch.float32 else 2)) * 1.1
            free, _ = torch.cuda.mem_get_info(m._device)
            self.recompute = need > 0.5 * free
        kept = []
        if self.recompute:
            for i in range(m.n_layers):
                hs[i].copy_(h)                   # the block input (checkpoint, main_finetune.py:268-276)
                self.await_weights(f"layer{i}")
                self._block_forward(i, h, B, S, keep=False)
        else:
            # stored activations: block i reads hs[i] and writes hs[i + 1] (its w2 GEMM's residual epilogue stores there), so the
            # checkpoints cost no copy (was one 143-MB read + write per layer)
            hs[0].copy_(h)
            for i in range(m.n_layers):
                self.await_weights(f"layer{i}")
                kept.append(self._block_forward(i, hs[i], B, S, keep=True, tag=f".L{i}", h_out=hs[i + 1]))
            h = hs[m.n_layers]
        self.sync_optimizer()                    # "head" is last in forward_order: everything has landed from here on
        xt = self._buf("xn_text", (B * T, dim))
        hv = h.view(B, S, dim)
        for b in range(B):
            ops.rmsnorm(hv[b, W:], m.norm.weight, xt[b * T:(b + 1) * T], a.norm_eps)
        logits = self._buf("logits", (B * T, V))
        ops.gemm_nt(xt, im["out"], logits)
        lab = self._buf("lab", (B, T), torch.int64, zero=True)
        lab[:, :T - 1].copy_(labels[:, 1:])      # shift (meta.py:256-257); last position predicts nothing
        lab = lab.view(-1)
        n_valid = self._buf("n_valid", (1,), torch.int32)
        ops.count_valid(lab, n_valid)
        row_loss = self._buf("row_loss", (B * T,), torch.float32)
        ops.cross_entropy(logits, lab, row_loss)
        nv = n_valid.to(torch.float32)[0]
        loss = torch.where(nv > 0, row_loss.sum() / torch.clamp(nv, min=1.0), torch.zeros_like(nv))   # meta.py:259-262
        self._saved = dict(B=B, T=T, W=W, S=S, h=h, hs=hs, xt=xt, logits=logits, lab=lab, n_valid=n_valid, tokens=examples.contiguous(), vis=vis,
                           kept=kept if not self.recompute else None)
        return loss

    @torch.no_grad()
    def backward(self, grad_scale: float = 1.0) -> None:
        s = self._saved
        self._dha_ready = False
        self._li_checked = False
        assert s is not None, "backward() without forward_loss()"
        m, a = self.m, self.m.args
        self.sync_optimizer()                    # the update reads the gradients this backward overwrites
        im = self._images()
        self.ensure_grads()
        if not self.nn_dgrad and not self.lora and not getattr(self, "_nt_mem_checked", False):
            # the transposed images cost 2 B per matrix parameter (12.3 GiB at 7B, 23.9 at 13B): on a replica that would not leave
            # 8 GiB free after them (13B at DP 8 with persistent wire buckets is at 270 of 288 GiB), stay on the NN kernel
            self._nt_mem_checked = True
            need = 2 * sum(q.numel() for g_ in im.groups() if g_ != "vp" for q in im._params(g_)[0])
            if not getattr(self, "_opt_stepped", False):          # the AdamW moments (8 B per trainable parameter) are allocated by the first step
                need += 8 * sum(q.numel() for q in m.parameters() if q.requires_grad)
            free, _ = torch.cuda.mem_get_info(m._device)
            free += torch.cuda.memory_reserved(m._device) - torch.cuda.memory_allocated(m._device)
            # (a replica that recomputes its blocks is one sized against the HBM limit -- 13B at micro-batch 4: measured 452.5 / 453.7 ms
            # with the transposed images against 455.5 / 452.4 without, for 24 GiB: not taken there unless A3V_NN_DGRAD=0 asks for it)
            if free - need < 8 * 2 ** 30 or (self.recompute and "A3V_NN_DGRAD" not in os.environ):
                self.nn_dgrad = True
        if self.lora and not getattr(self, "_nt_mem_checked", False):
            # LoRA engines: the input gradients of the frozen matrices run on the NT kernel over transposed images that
            # are built once (2 B per base parameter: 12.3 GiB at 7B) and, with nn_dgrad off, the head's W^T too -- the same headroom
            # rule as the full fine-tune above: without 8 GiB left after them the NN kernel on the forward images is used instead
            self._nt_mem_checked = True
            if not self.nn_dgrad or self.lora_nt_dgrad not in ("", "0"):
                need = 2 * sum(q.numel() for g_ in im.groups() if g_ != "vp" for q in im._params(g_)[0])
                free, _ = torch.cuda.mem_get_info(m._device)
                free += torch.cuda.memory_reserved(m._device) - torch.cuda.memory_allocated(m._device)
                if free - need < 8 * 2 ** 30:
                    self.nn_dgrad = True
                    self.lora_nt_dgrad = "0"
        if self.sumsq_sink is not None:
            self.sumsq_sink.begin_backward()
        B, T, W, S = s["B"], s["T"], s["W"], s["S"]
        rows, dim, V = B * S, a.dim, a.vocab_size
        # ---- CE + LM head + final norm
        dlog = self._buf("dlogits", (B * T, V))
        ops.cross_entropy(s["logits"], s["lab"], self._buf("row_loss", (B * T,), torch.float32), dlog, s["n_valid"], grad_scale)
        if self._has("output.weight"):
            self._wgrad(dlog, s["xt"], self._views["output.weight"], "out", ("output.weight",))
        dxt = self._buf("dxn_text", (B * T, dim))
        self._dgrad_w(dlog, "out", dxt)
        kx = self._kext() > 0 and rows > 16 and m.head_dim in (64, 128) and self.fuse_qkv_rope       # as in _block_forward
        self._dh_full = None
        if self.stream == torch.bfloat16 and self.act == torch.bfloat16 and kx:
            # bf16 stream + adapters inside the GEMMs: dh IS the dy operand of the wo / w2 groups, so it lives in the first `dim`
            # columns of a buffer with room for dt = dy . B behind it (no copy of dh per layer)
            # (width of the wo / w2 groups' adapter block, pad64(r) -- NOT _kext(), the qkv group's pad64(3 r): the two differ from r = 22 on)
            self._dh_full = self._buf("dh.x", (rows, dim + self._kext_cols("w2.0")), self.stream, zero=True)
            dh = self._dh_full[:, :dim]
        else:
            dh = self._buf("dh", (rows, dim), self.stream, zero=True)
        hv, dhv = s["h"].view(B, S, dim), dh.view(B, S, dim)
        for b in range(B):
            ops.rmsnorm_bwd(hv[b, W:], m.norm.weight, dxt[b * T:(b + 1) * T], dhv[b, W:], self._views.get("norm.weight"), a.norm_eps)
        self._notify("head")
        # ---- decoder blocks, last to first (recompute from the saved block input)
        for i in range(m.n_layers - 1, -1, -1):
            self._block_backward(i, s["hs"][i], dh, B, S)
            self._notify(f"layer{i}")
        # ---- embeddings and projector
        if self._has("tok_embeddings.weight"):
            ops.embed_bwd(s["tokens"], dh, self._views["tok_embeddings.weight"], B, T, W, dim)
        self._notify("embed")
        if s["vis"] is not None:
            self._encode_image_backward(dh, s["vis"], B, S)
        # announced on EVERY backward, image or not: under accumulation an earlier micro-step of the cycle may have put projector
        # gradients there, and every rank must hand the same buckets to the collective on the boundary micro-step
        self._notify("vision_proj")
        self._saved = None

    def _notify(self, bucket: str):
        if self.on_layer_grads_ready is None:
            return
        for name, st, en in self._ranges:
            if name == bucket:
                self.on_layer_grads_ready(name, st, en)

    # ------------------------------------------------------------------ vision (frozen ViT, trainable projector)
    def _encode_image_train(self, h, image, B, S, qformer_feats, extra_feats):
        """Image words of the training forward (LLM/llama_ens5.py:377-458, 471-478): frozen CLIP ViT on the HIP path, then the two
        TRAINABLE projectors -- visual_proj over the concat [CLIP | ConvNeXt-XXL | DINOv2] (5632 wide in the reference configuration)
        and qformer_proj over the 32 Q-Former tokens.  The three out-of-scope encoders are frozen and run under no_grad in the
        reference (:399): their outputs are INPUTS here (``extra_feats`` / ``qformer_feats`` or the plugin's provider hooks), and
        only the projectors are differentiated (_encode_image_backward)."""
        m, a = self.m, self.m.args
        im = self._images()
        _, _, L = m._vit_geometry()
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        slots = tuple(range(len(images)))
        views = m._gather_views(images, B, m.clip.visual.conv1.weight.dtype)
        N = len(images) * a.n_views * B
        feats = m.clip_encode_image(views)                      # frozen, compute dtype of the clip params
        if extra_feats is None and getattr(m, "extra_feat_fns", None):
            extra_feats = [fn(views) for fn in m.extra_feat_fns]
        if qformer_feats is None and getattr(m, "qformer_fn", None) is not None:
            qformer_feats = m.qformer_fn(views)
        if a.extra_feat_dim:
            if extra_feats is None:
                raise ValueError("extra_feat_dim is set but no ConvNeXt / DINOv2 features were given (extra_feats= or provider hooks)")
            cat = self._buf("feats_cat", (N * L, a.vit_width + a.extra_feat_dim))
            cat[:, :a.vit_width] = feats                        # data movement only (concat, llama_ens5.py:436-440)
            o = a.vit_width
            for e in extra_feats:
                e2 = e.reshape(N * L, -1)
                cat[:, o:o + e2.shape[1]] = e2
                o += e2.shape[1]
            assert o == cat.shape[1], "extra feature widths do not add up to extra_feat_dim"
            feats = cat
        elif feats.dtype != self.act:
            f2 = self._buf("feats_act", tuple(feats.shape))
            ops.cast(feats, f2)
            feats = f2
        proj = self._buf("proj", (N * L, a.dim))
        ops.gemm_nt(feats, im["vp"], proj, bias=im["vp.b"])
        vp1 = getattr(m.visual_proj, "1")
        clip_map, qf_map, start_rows, end_rows = m._image_row_maps(B, S, slots)
        ops.layernorm(proj, vp1.weight, vp1.bias, h, row_map=clip_map)
        out = dict(feats=feats, proj=proj, clip_map=clip_map, start_rows=start_rows, end_rows=end_rows, N=N, L=L, slots=slots)
        if a.qformer_tokens:
            if qformer_feats is None:
                raise ValueError("qformer_tokens is set but no Q-Former features were given (qformer_feats= or provider hook)")
            Q = a.qformer_tokens
            qin = self._buf("qf_in", (N * Q, 768))
            qin.copy_(qformer_feats.reshape(N * Q, 768))
            qout = self._buf("qf_out", (N * Q, a.dim))
            ops.gemm_nt(qin, im["vq"], qout, bias=im["vq.b"])
            qp1 = getattr(m.qformer_proj, "1")
            ops.layernorm(qout, qp1.weight, qp1.bias, h, row_map=qf_map)
            out.update(qin=qin, qout=qout, qf_map=qf_map, Q=Q)
        tags = m._image_slots()
        for j, sl in enumerate(slots):
            ops.fill_rows(tags[sl][0].view(-1), h, start_rows[j])
            ops.fill_rows(tags[sl][1].view(-1), h, end_rows[j])
        return out

    def _encode_image_backward(self, dh, vis, B, S):
        m, a = self.m, self.m.args
        vp1 = getattr(m.visual_proj, "1")
        rowsv = vis["N"] * vis["L"]
        dproj = self._buf("dproj", (rowsv, a.dim))
        ops.layernorm_bwd(vis["proj"], vp1.weight, dh, vis["clip_map"], dproj, self._views["visual_proj.1.weight"],
                          self._views["visual_proj.1.bias"])
        self._wgrad(dproj, vis["feats"], self._views["visual_proj.0.weight"], "vp")
        ops.rows_sum(dproj, None, rowsv, self._views["visual_proj.0.bias"])
        if "qin" in vis:
            qp1 = getattr(m.qformer_proj, "1")
            rowsq = vis["N"] * vis["Q"]
            dq = self._buf("dqf", (rowsq, a.dim))
            ops.layernorm_bwd(vis["qout"], qp1.weight, dh, vis["qf_map"], dq, self._views["qformer_proj.1.weight"],
                              self._views["qformer_proj.1.bias"])
            self._wgrad(dq, vis["qin"], self._views["qformer_proj.0.weight"], "vq")
            ops.rows_sum(dq, None, rowsq, self._views["qformer_proj.0.bias"])
        names = [("start_img", "end_img"), ("start_depth_img", "end_depth_img")]
        for j, sl in enumerate(vis["slots"]):
            ops.rows_sum(dh, vis["start_rows"][j], vis["start_rows"][j].numel(), self._views[names[sl][0]].view(-1))
            ops.rows_sum(dh, vis["end_rows"][j], vis["end_rows"][j].numel(), self._views[names[sl][1]].view(-1))


class _Images:
    """dict-like view of the engine's compute-dtype weight images, built per group on demand."""

    def __init__(self, eng: "TrainEngine"):
        self.eng = eng
        self.store: Dict[str, torch.Tensor] = {}
        self.ver: Dict[str, tuple] = {}
        self.tver: Dict[str, tuple] = {}      # version of each transposed image (built lazily, per key)
        self._sinks = None                    # id(param) -> (image key, first row), built on first use

    def _group(self, key: str):
        if key.startswith(("qkv.", "wo.", "w13.", "w2.")):
            return "L" + key.split(".")[1]
        return "vp" if key.startswith(("vp", "vq")) else "out"

    def _fused(self, ps):
        """The fused matrix of consecutive parameters: a VIEW of the flat parameter buffer under ZeRO-1 (always current, no copy: the
        image IS the parameter storage), else their concatenation."""
        v = self.eng._pview(list(ps)) if self.eng.zero1_world and not self.eng.lora else None
        if v is not None:
            return v
        return torch.cat(list(ps), dim=0) if len(ps) > 1 else ps[0]

    def _both(self, key, w):      # forward image W [N,K] now; W^T [K, N padded to 64] only when somebody asks for key + ".t"
        ext = self.eng._kext_cols(key)
        if ext:                   # LoRA inside the GEMMs: one [N + Rp, K + Rp] image = [[W, B], [A, 0]]; forward reads [W | B] (rows 0..N),
            N, K = w.shape        # the input gradient reads [W ; A] (columns 0..K); A / B blocks written by _lora_step_images / a3v_lora_refresh
            full = torch.zeros(N + ext, K + ext, dtype=self.eng.act, device=w.device)
            full[:N, :K] = w.to(self.eng.act)
            self.store[key] = full[:N, :K]
            self.store[key + ".x"] = full[:N]
            self.store[key + ".y"] = full[:, :K]
            if self.eng._nt_dgrad(key):
                # [W^T | A^T]  [K, N + Rp]: dx = [dy | dt] . (this)^T on the NT ring kernel; W is frozen, the A^T block is re-written
                # with the adapters (a3v_adamw_multi / a3v_lora_refresh)
                ft = torch.zeros(K, N + ext, dtype=self.eng.act, device=w.device)
                ft[:, :N] = full[:N, :K].t()
                self.store[key + ".yT"] = ft
        else:
            self.store[key] = w.to(self.eng.act).contiguous()

    def _transposed(self, key: str, ver) -> torch.Tensor:
        if self.tver.get(key) != ver:
            act = self.eng.act
            wa = self.store[key[:-2]]
            N, K = wa.shape
            Np = _pad64(N)
            if Np != N:
                wp = torch.zeros(Np, K, dtype=act, device=wa.device)
                wp[:N] = wa
            else:
                wp = wa
            wt = self.store.get(key)
            if wt is None or wt.shape != (K, Np) or wt.dtype != act:
                # row pitch off the powers of two: a3v_adamw_scaled_t writes 64 rows of this image per tile, and at an 8-KiB pitch they
                # all fall on the same HBM channels (w2 / wo at 7B: 281 vs 255 us per update)
                ld = Np + 64 if (Np * wa.element_size()) % 4096 == 0 else Np
                wt = torch.empty(K, ld, dtype=act, device=wa.device)[:, :Np]
            ops.transpose(wp, wt, Np, K, Np)
            self.store[key] = wt
            self.tver[key] = ver
        return self.store[key]

    def _params(self, g: str):
        """(parameters, [(image key, first row)] in the same order) of image group g."""
        m = self.eng.m
        if g.startswith("L"):
            i = int(g[1:])
            a, f = m.layers[i].attention, m.layers[i].feed_forward
            H, Hkv, hd, F = m.n_heads, m.n_kv_heads, m.head_dim, m.ffn
            return ((a.wq.weight, a.wk.weight, a.wv.weight, a.wo.weight, f.w1.weight, f.w3.weight, f.w2.weight),
                    ((f"qkv.{i}", 0), (f"qkv.{i}", H * hd), (f"qkv.{i}", (H + Hkv) * hd), (f"wo.{i}", 0), (f"w13.{i}", 0), (f"w13.{i}", F),
                     (f"w2.{i}", 0)))
        if g == "out":
            return (m.output.weight,), (("out", 0),)
        vp0 = getattr(m.visual_proj, "0")
        if getattr(m.args, "qformer_tokens", 0):
            qp0 = getattr(m.qformer_proj, "0")
            return (vp0.weight, vp0.bias, qp0.weight, qp0.bias), (("vp", 0), ("vp.b", 0), ("vq", 0), ("vq.b", 0))
        return (vp0.weight, vp0.bias), (("vp", 0), ("vp.b", 0))

    def _key(self, ps) -> tuple:
        return tuple(param_state_key(q) for q in ps) + (self.eng.act, str(self.eng.m._device))

    def groups(self):
        m = self.eng.m
        return [f"L{i}" for i in range(m.n_layers)] + ["out"] + (["vp"] if getattr(m, "visual_proj", None) is not None else [])

    def sink(self, p) -> Optional[torch.Tensor]:
        """bf16 destination of parameter p inside its (already built) forward image, or None."""
        if self.eng.act != torch.bfloat16:
            return None
        if self._sinks is None:
            self._sinks = {}
            for g in self.groups():
                ps, where = self._params(g)
                for q, (key, row) in zip(ps, where):
                    self._sinks[id(q)] = (key, row)
        ent = self._sinks.get(id(p))
        if ent is None:
            return None
        img = self.store.get(ent[0])
        if img is None or img.dtype != torch.bfloat16:
            return None
        return img[ent[1]:ent[1] + p.shape[0]] if img.dim() == 2 else img

    def sink_t(self, p):
        """(columns of parameter p inside the transposed image of its fused matrix [K, Np], Np) -- only once that image exists (the
        first backward built it) and p is a 64-aligned block of it."""
        if self.eng.act != torch.bfloat16 or p.dim() != 2 or (p.shape[0] & 63) or (p.shape[1] & 63):
            return None
        self.sink(p)                                   # (builds the parameter -> (key, row) map)
        ent = self._sinks.get(id(p))
        if ent is None or ent[0].endswith(".b"):
            return None
        wt = self.store.get(ent[0] + ".t")
        if wt is None or wt.dtype != torch.bfloat16 or wt.dim() != 2 or wt.shape[0] != p.shape[1] or ent[1] + p.shape[0] > wt.shape[1] \
                or (ent[1] & 3) or self.tver.get(ent[0] + ".t") is None:
            return None
        return wt[:, ent[1]:ent[1] + p.shape[0]], int(wt.stride(0))

    def adopted(self, written_ids, written_t_ids=()) -> None:
        """The optimizer wrote the bf16 values of these parameters into their images: groups written completely are current (and
        the transposed image of a fused matrix whose every parameter also went through ``a3v_adamw_scaled_t``)."""
        for g in self.groups():
            ps, where = self._params(g)
            if g in self.ver and all(id(q) in written_ids for q in ps):
                old = self.ver[g]
                self.ver[g] = self._key(ps)
                for key in {k for k, _ in where}:
                    if self.tver.get(key + ".t") == old and all(id(q) in written_t_ids for q, (k, _) in zip(ps, where) if k == key):
                        self.tver[key + ".t"] = self.ver[g]

    def __getitem__(self, key: str) -> torch.Tensor:
        eng, m = self.eng, self.eng.m
        g = self._group(key)
        ps, _ = self._params(g)
        if g.startswith("L"):
            i = int(g[1:])
        ver = self._key(ps)
        if self.ver.get(g) != ver:
            with torch.no_grad():
                if g.startswith("L"):
                    self._both(f"qkv.{i}", self._fused(ps[0:3]))
                    self._both(f"wo.{i}", self._fused(ps[3:4]))
                    self._both(f"w13.{i}", self._fused(ps[4:6]))
                    self._both(f"w2.{i}", self._fused(ps[6:7]))
                elif g == "out":
                    self._both("out", self._fused(ps[0:1]))
                else:
                    self._both("vp", ps[0])
                    self.store["vp.b"] = ps[1].to(eng.act)
                    if len(ps) == 4:
                        self._both("vq", ps[2])
                        self.store["vq.b"] = ps[3].to(eng.act)
            self.ver[g] = ver
        if key.endswith(".t"):
            return self._transposed(key, ver)
        return self.store[key]


class _StepLoss(torch.autograd.Function):
    """loss = engine.forward_loss(...); backward runs the HIP backward and fills param.grad itself."""

    @staticmethod
    def forward(ctx, anchor, engine, examples, labels, image):
        ctx.engine = engine
        return engine.forward_loss(examples, labels, image)

    @staticmethod
    def backward(ctx, grad_out):
        # the incoming gradient of the loss is the caller's 1 / accum_iter: a trainer that knows it sets
        # engine.static_grad_scale and spares the host read of a device scalar on every micro-step
        hint = getattr(ctx.engine, "static_grad_scale", None)
        ctx.engine.backward(float(grad_out) if hint is None else float(hint))
        return torch.zeros((), device=grad_out.device), None, None, None, None


def step_loss(engine: TrainEngine, anchor: torch.Tensor, examples, labels, image=None) -> torch.Tensor:
    return _StepLoss.apply(anchor, engine, examples, labels, image)


def hbm_budget(dim: int, n_layers: int, n_heads: int, ffn: int, vocab: int, batch: int, seq: int, text: int, n_kv_heads: Optional[int] = None,
               vit_params: int = 304_000_000, proj_in: int = 1024, world: int = 1, wire_bytes: int = 2, recompute: bool = False,
               stream_bytes: int = 2, zero1: bool = False, transposed_images: bool = False) -> Dict[str, int]:
    """Bytes of one pure-DP replica of the FULL fine-tune (every rank of a DP job holds exactly this; SURVEY 7 "13B full fine-tune
    memory", main_finetune.py:241-276): what ``TrainEngine`` + ``FusedAdamW`` + ``dp.GradReducer`` allocate, from their own layouts.
    fp32 masters / flat gradient buffer / two AdamW moments of every trainable parameter, the bf16 GEMM images of the decoder and
    head matrices, the frozen bf16 ViT, the reducer's persistent wire buckets (``wire_bytes`` per gradient element when world > 1
    and the wire dtype is not fp32), the residual-stream checkpoints [L+1, rows, dim] (``stream_bytes`` per element: 2 = the bf16 stream
    of round 3 on, 4 = the fp32 stream of rounds 1-2), the block intermediates (per layer when activations are stored, once when
    blocks are recomputed), and the CE buffers; ``transposed_images``: + the W^T images of the round-5 default.  Checked against the measured peaks of the bench
    (tests/test_dp_cpu.py::test_dp_replica_fits_288_gib)."""
    hkv = n_kv_heads or n_heads
    hd = dim // n_heads
    qkv = (n_heads + 2 * hkv) * hd
    p_layer = dim * qkv + dim * dim + 3 * dim * ffn
    p_mat = n_layers * p_layer + dim * vocab                      # matrices that get a bf16 GEMM image
    p_train = p_mat + dim * vocab + (2 * n_layers + 1) * dim + proj_in * dim + 4 * dim     # + embeddings, norms, projector, tags
    rows = batch * seq
    spad = (seq + 63) // 64 * 64
    sb = stream_bytes
    block = rows * (2 * dim * 2 + qkv * 2 + 2 * n_heads * hd * 2 + 3 * ffn * 2 + dim * sb) + batch * hkv * spad * hd * 2 + batch * n_heads * seq * 4
    bwd_ws = rows * ((dim * 2 if sb == 4 else 0) + ffn * 2 + 2 * ffn * 2 + dim * 2 + n_heads * hd * 2 + qkv * 2 + dim * sb) + batch * hkv * hd * spad * 2
    if zero1:
        # ZeRO-1 (zero1.Zero1Optimizer, TrainEngine(zero1_world=world)): the big matrices (decoder linears, embeddings, head) exist once in
        # bf16 (parameters = GEMM images = all-gather destination), their fp32 masters, both moments, the fp32 gradient slice and the bf16
        # output slice only for 1 / world of them; ONE wire bucket of the largest layer instead of persistent buckets for all
        p_big = p_mat + dim * vocab
        p_small = p_train - p_big
        layer_big = p_layer if n_layers else 0
        out = {"masters_fp32": 4 * p_small + 4 * p_big // world, "grads_fp32": 4 * p_train + 4 * p_big // world,
               "adamw_moments_fp32": 8 * p_small + 8 * p_big // world, "images_bf16": 2 * p_big + 2 * p_big // world,
               "vit_bf16": 2 * vit_params,
               "wire_buckets": (wire_bytes * max(layer_big, dim * vocab) + wire_bytes * max(layer_big, dim * vocab) // world) if wire_bytes != 4 else 0}
    else:
        out = {
            "masters_fp32": 4 * p_train, "grads_fp32": 4 * p_train, "adamw_moments_fp32": 8 * p_train, "images_bf16": 2 * p_mat,
            "vit_bf16": 2 * vit_params,
            "wire_buckets": wire_bytes * p_train if world > 1 and wire_bytes != 4 else 0}
    if transposed_images and not zero1:
        out["images_t_bf16"] = 2 * p_mat          # W^T of every decoder / head matrix (round 5: input gradients on the NT kernel, a3v_adamw_scaled_t)
    out.update({
        "stream_checkpoints": (n_layers + 1) * rows * dim * sb + rows * dim * sb,
        "block_activations": block * (1 if recompute else n_layers),
        "backward_workspace": bwd_ws,
        "ce_buffers": batch * text * vocab * 2 * 2 + batch * text * dim * 2 * 2,
    })
    out["total"] = sum(out.values())
    return out
