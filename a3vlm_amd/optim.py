"""AdamW on the HIP kernel ``a3v_adamw`` with the state layout of ``torch.optim.AdamW`` (``step`` / ``exp_avg`` /
``exp_avg_sq`` per parameter), so optimizer checkpoints written by either (util/misc.py:324-569 layout, ``checkpoint.py``)
load into the other.  Reference: the trainer builds ``torch.optim.AdamW(param_groups, lr, betas=(0.9, 0.95))``
(main_finetune.py:138 of this package mirrors accessory/main_finetune.py) and calls ``optimizer.step()`` once per
accumulation cycle (engine_finetune.py:63).

One launch per parameter tensor streams p, g, m, v once (28 B per parameter).  ``image_of`` (optional) maps a parameter to a
bf16 tensor of the same shape that receives the rounded updated values in the same pass (the compute-dtype operand of the next
step's GEMMs).

``step(overlap=True)`` (with ``engine``): the launches go to the optimizer's own stream in the order the forward first touches the
parameters (``TrainEngine.forward_order``), one event per gradient bucket; ``TrainEngine.forward_loss`` of the next step waits
bucket by bucket, so the 28-B/parameter HBM stream of the update runs under the MFMA-bound GEMMs of the next forward instead of
in front of it.  Same arithmetic, same order per parameter.  The caller's contract: between ``step(overlap=True)`` and the next
``forward_loss`` / ``backward`` only the host and ``zero_grad(set_to_none=True)`` touch parameters or gradients; anything else
(checkpoints, evaluation through the model, a stock ``zero_grad(set_to_none=False)``) calls ``engine.sync_optimizer()`` first
(``MetaModel.zero_grad(set_to_none=False)``, ``MetaModel.state_dict`` and this optimizer's ``state_dict`` do so themselves).

Skipped steps: a negative or NaN ``grad_scale`` makes ``a3v_adamw_scaled`` a no-op on masters, moments and bf16 images (the trainer's
device-side non-finite guard).  The host cannot know that without reading the flag, so ``state[p]["step"]`` counts LAUNCHES, not
applied updates, and the weight images are marked current either way (they are: nothing changed).  The trainer reads the flag at
its logging / checkpoint / epoch boundaries and exits there, before anything is saved, so a skipped step is never followed by a
step whose bias correction would be off by one."""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch

from . import lib as _l


class FusedAdamW(torch.optim.Optimizer):
    bumps_version = True        # parameters' Tensor._version is incremented by step(): version-keyed caches need no extra hook

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 image_of: Optional[Callable[[torch.Tensor], Optional[torch.Tensor]]] = None, engine=None):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        # ``engine`` (a TrainEngine): the bf16 values go straight into the engine's fused weight images, which are then marked
        # current -- the next step does not re-cast 6.7 G fp32 parameters (cat + cast = 14 B per parameter) to rebuild them
        self.engine = engine
        self.image_of = image_of if engine is None else engine.image_sink
        self._stream: Optional[torch.cuda.Stream] = None
        # small tensors (adapters, norm weights, ...) of a param group go through ONE a3v_adamw_multi launch per step instead of one
        # launch each; A3V_ADAMW_MULTI=0 (A/B runs, equality tests) keeps the per-tensor launches
        self.multi_threshold = int(os.environ.get("A3V_ADAMW_MULTI", str(1 << 20)))
        self._tables = {}                      # param-group index -> (key, device table, max_n, [params])

    def state_dict(self):
        if self.engine is not None:
            self.engine.sync_optimizer()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """The device tables of the multi-tensor launch hold raw pointers of the moment tensors, which a load replaces."""
        if self.engine is not None:
            self.engine.sync_optimizer()
        super().load_state_dict(state_dict)
        self._tables.clear()

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_tables"):
            self._tables.clear()

    def _ordered(self):
        """[(bucket or None, group, parameter)] in forward order (parameters the engine does not know first)."""
        rank, name = {}, {}
        for r, (bucket, plist) in enumerate(self.engine.forward_order()):
            for _, q in plist:
                rank[id(q)], name[id(q)] = r, bucket
        items = [(rank.get(id(p), -1), name.get(id(p)), g, p) for g in self.param_groups for p in g["params"]]
        items.sort(key=lambda it: it[0])
        return [(b, g, p) for _, b, g, p in items]

    @torch.no_grad()
    def step(self, closure=None, grad_scale: Optional[torch.Tensor] = None, overlap: bool = False):
        """``grad_scale``: optional fp32 device scalar every gradient is multiplied by as it is read (the clip coefficient of
        ``dp.clip_grad_norm(..., defer=True)``): same update as ``grad.mul_(coef)`` + ``step()``; ``p.grad`` is left unscaled.
        ``overlap``: see the module docstring."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _l.load()
        overlap = bool(overlap) and self.engine is not None
        if grad_scale is not None and (not grad_scale.is_cuda or grad_scale.dtype != torch.float32 or grad_scale.numel() != 1):
            raise RuntimeError("grad_scale must be a one-element fp32 device tensor")
        gs_ptr = grad_scale.data_ptr() if grad_scale is not None else None
        order = self._ordered() if overlap else [(None, g, p) for g in self.param_groups for p in g["params"]]
        for _, _, p in order:                  # states first: their zero fills are ordered before the hand-over to the other stream
            if p.grad is None:
                continue
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() \
                    or not p.grad.is_contiguous():
                raise RuntimeError("FusedAdamW takes contiguous fp32 device parameters and gradients (no CPU fallback)")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)          # torch.optim.AdamW's (non-capturable) layout
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        cur = torch.cuda.current_stream()
        if overlap:
            if self._stream is None:
                self._stream = torch.cuda.Stream(priority=int(os.environ.get("A3V_ADAMW_STREAM_PRIO", "-1")))
            self.engine.sync_optimizer()                       # a previous overlapped step nobody consumed
            self._stream.wait_stream(cur)                      # gradients, states and the clip coefficient are ready
            if grad_scale is not None:
                grad_scale.record_stream(self._stream)
            stream = self._stream.cuda_stream
        else:
            stream = cur.cuda_stream
        written = set()
        written_t = set()
        adapters = set()
        last = None
        small = {}                             # id(group) -> [(p, sink)] handled by the multi-tensor launch (not with overlap: per-bucket events)
        if not overlap and self.multi_threshold > 0:
            for gi, group in enumerate(self.param_groups):
                lst = []
                for p in group["params"]:
                    if p.grad is None or p.numel() > self.multi_threshold:
                        continue
                    img = self.image_of(p) if self.image_of is not None else None
                    sink = None
                    if img is not None and self.engine is not None and hasattr(self.engine, "image_sink_t") and self.engine.image_sink_t(p) is not None:
                        continue                     # a matrix whose transposed image is in use: the per-tensor a3v_adamw_scaled_t launch below
                    if img is not None:
                        if img.dtype != torch.bfloat16 or img.shape != p.shape or not img.is_contiguous():
                            raise RuntimeError("image_of must return a contiguous bf16 tensor of the parameter's shape")
                        sink = (img.data_ptr(), p.shape[-1] if p.dim() > 1 else p.numel(), 1, 0, 0, 0)
                    elif self.engine is not None and hasattr(self.engine, "adapter_sink"):
                        sink = self.engine.adapter_sink(p)          # (d1, s1r, s1c, d2, s2r, s2c) inside the fused LoRA group images
                    st = self.state[p]
                    # a3v_adamw_multi does 16-byte vector accesses on p / g / m / v (and 8-byte ones on a contiguous image) through a table
                    # the C entry cannot check: anything misaligned or non-contiguous takes the per-tensor launch, which checks for itself
                    ok = all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
                    if img is not None and img.data_ptr() % 8:
                        ok = False
                    if not ok:
                        continue
                    lst.append((p, sink, img is not None))
                if len(lst) >= 2:
                    small[gi] = lst
        in_multi = {id(p) for lst in small.values() for p, _, _ in lst}
        # only adapters whose bf16 values the multi launch really writes into the fused images count as adopted by the engine
        adapters = {id(p) for lst in small.values() for p, sink, same in lst if sink is not None and not same}
        for bucket, group, p in order:
            if overlap and bucket != last:
                if last is not None:
                    self.engine._weights_ready[last] = self._stream.record_event()
                last = bucket
            if p.grad is None or id(p) in in_multi:
                continue
            b1, b2 = group["betas"]
            st = self.state[p]
            st["step"] += 1
            img = self.image_of(p) if self.image_of is not None else None
            if img is not None and (img.dtype != torch.bfloat16 or img.shape != p.shape or not img.is_contiguous()):
                raise RuntimeError("image_of must return a contiguous bf16 tensor of the parameter's shape")
            imgt = self.engine.image_sink_t(p) if (img is not None and self.engine is not None and hasattr(self.engine, "image_sink_t")) else None
            if imgt is not None and p.is_contiguous():
                # full fine-tune with input gradients on the NT kernel: the same pass also writes p's columns of the transposed image
                rc = lib.a3v_adamw_scaled_t(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                            p.shape[0], p.shape[1], float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                            float(group["weight_decay"]), int(st["step"].item()), img.data_ptr(), imgt[0].data_ptr(),
                                            imgt[1], gs_ptr, stream)
                _l.check(rc, "a3v_adamw_scaled_t")
                written_t.add(id(p))
            else:
                rc = lib.a3v_adamw_scaled(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                          float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                          int(st["step"].item()), img.data_ptr() if img is not None else None, gs_ptr, stream)
                _l.check(rc, "a3v_adamw_scaled")
            # the kernel wrote through raw pointers: tell autograd / version-keyed caches (the engine's bf16 weight images)
            torch.autograd.graph.increment_version(p)
            if img is not None:
                written.add(id(p))
        for gi, lst in small.items():
            group = self.param_groups[gi]
            b1, b2 = group["betas"]
            steps = set()
            for p, _, _ in lst:
                st = self.state[p]
                st["step"] += 1
                steps.add(int(st["step"].item()))
            if len(steps) != 1:
                raise RuntimeError("FusedAdamW multi-tensor launch: the small tensors of a param group must share their step count")
            # the table holds raw pointers of the moments too: load_state_dict() replaces those tensors
            key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), sink)
                        for p, sink, _ in lst)
            ent = self._tables.get(gi)
            if ent is None or ent[0] != key:
                rows = []
                for p, sink, _ in lst:
                    st = self.state[p]
                    cols = p.shape[-1] if p.dim() > 1 else p.numel()
                    d = sink if sink is not None else (0, 0, 0, 0, 0, 0)
                    rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), cols,
                                 d[0], d[1], d[2], d[3], d[4], d[5]])
                tab = torch.tensor(rows, dtype=torch.int64).to(lst[0][0].device)
                ent = self._tables[gi] = (key, tab, max(p.numel() for p, _, _ in lst))
            rc = lib.a3v_adamw_multi(ent[1].data_ptr(), len(lst), ent[2], float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                     float(group["weight_decay"]), steps.pop(), gs_ptr, stream)
            _l.check(rc, "a3v_adamw_multi")
            for p, sink, same_shape in lst:
                torch.autograd.graph.increment_version(p)
                if same_shape:
                    written.add(id(p))
        if overlap:
            self.engine._weights_ready[last if last is not None else "head"] = self._stream.record_event()
        if self.engine is not None:
            if written_t:
                self.engine.images_adopted(written, written_t)
            else:
                self.engine.images_adopted(written)
            if adapters and hasattr(self.engine, "adapters_adopted"):
                self.engine.adapters_adopted(adapters)
        return loss
