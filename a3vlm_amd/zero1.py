"""ZeRO-1 for the full fine-tune of BASELINE configs[3] (Llama-2-13B, DP = 8): the reference trains it under
``FSDP(ShardingStrategy.SHARD_GRAD_OP)`` with ``MixedPrecision(param_dtype=bf16, reduce_dtype=bf16)`` (main_finetune.py:241-263) --
gradients are reduce-scattered, every rank owns 1/N of the fp32 parameters and of the AdamW state, and the bf16 compute parameters are
all-gathered.  Pure DP replicas hold 16 B per parameter (fp32 master + gradient + two moments) on EVERY rank: 13 B parameters only fit
at micro-batch 4 with recompute.  Here (the ZeRO-1 fall-back it names):

  * the big matrices (decoder linears, embeddings, LM head) live ONLY as the bf16 values every rank multiplies with, in one flat buffer
    laid out like the engine's flat fp32 gradient buffer (``TrainEngine(zero1_world=N)``: the parameters and the engine's GEMM images are
    views of it -- no fp32 replica, no separate image copy);
  * per gradient bucket (one decoder layer): ``reduce_scatter`` of the bucket (bf16 wire = the reference's ``reduce_dtype``, issued on the
    side stream the moment the layer's backward has produced it, like ``dp.GradReducer``'s all-reduce) -> this rank's 1/N slice of the
    averaged gradient in fp32;
  * AdamW on the slice: fp32 master + two moments of 1/N of the parameters (``a3v_adamw_scaled`` with the bf16 image sink = the slice of
    the flat parameter buffer's wire copy), global-norm clip coefficient applied as the gradient is read;
  * ``all_gather`` of the updated bf16 slices straight into the flat parameter buffer: every rank's parameters and GEMM images are current.
    Wire bytes per step = reduce-scatter + all-gather = one ring all-reduce of bf16 gradients.
  * small parameters (norm weights, projector, tags: fp32, replicated) keep the plain path: their gradient ranges are all-reduced and a
    ``FusedAdamW`` updates them identically on every rank.

The collective calls are ``torch.distributed`` (RCCL on the GPU; gloo in the CPU tests, which pass their own ``update`` function: there is
no CPU AdamW in the product)."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch


def _scale_cast(src: torch.Tensor, dst: torch.Tensor, scale: float) -> None:
    from .dp import _scale_cast as sc
    sc(src, dst, scale)


def _hip_adamw(master, grad, m, v, out, lr, b1, b2, eps, wd, step, grad_scale) -> None:
    """One segment of the local shard on the device: a3v_adamw_scaled with the updated master's bf16 rounding written to ``out``."""
    from . import lib as _l
    if not master.is_cuda:
        raise RuntimeError("Zero1Optimizer: the AdamW update runs on the device (pass update=... only in the CPU tests)")
    rc = _l.load().a3v_adamw_scaled(master.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), master.numel(), float(lr), float(b1), float(b2),
                                   float(eps), float(wd), int(step), out.data_ptr() if out.dtype == torch.bfloat16 else None,
                                   grad_scale.data_ptr() if grad_scale is not None else None, torch.cuda.current_stream().cuda_stream)
    _l.check(rc, "a3v_adamw_scaled")
    if out.dtype != torch.bfloat16:
        out.copy_(master)


class Zero1Optimizer:
    """Sharded AdamW over the engine's flat buffers.  Engine interface (``TrainEngine(zero1_world=N)`` or a test double):
    ``flat_grads()``, ``flat_params()``, ``zero1_buckets()`` -> [(name, start, end, shard_start, shard_end, [(seg_start, seg_end,
    weight_decay)])] with ``shard_end - shard_start`` a multiple of 64 * world, ``zero1_mark_fresh()``, and the ``on_layer_grads_ready`` hook.
    ``small``: an optimizer for the replicated (fp32) parameters, stepped with the same clip coefficient."""

    zero1 = True

    def __init__(self, engine, dist, lr: float = 1e-3, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.0,
                 reduce_dtype: Optional[torch.dtype] = torch.bfloat16, small=None, update: Optional[Callable] = None, group=None):
        self.eng, self.dist, self.group, self.small = engine, dist, group, small
        self.world = dist.get_world_size(group) if group is not None else dist.get_world_size()
        self.rank = dist.get_rank(group) if group is not None else dist.get_rank()
        self.update = update or _hip_adamw
        self.reduce_dtype = reduce_dtype
        self.enabled = True                       # False on the micro-steps of an accumulation window that do not end it (no_sync)
        self.stub_collective = False              # bench: everything but the collectives
        self._group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "params": []}
        self.step_count = 0
        try:
            self._avg = dist.get_backend(group) == "nccl" and hasattr(dist.ReduceOp, "AVG")
        except Exception:
            self._avg = False
        self.wire_bytes_last_step = 0
        self._wire_bytes = 0
        self._stream = None
        self._wire: Optional[torch.Tensor] = None
        self._reduced: set = set()
        flat_p = engine.flat_params()
        dev = flat_p.device
        self.buckets: List[dict] = []
        for name, s, e, ss, se, segs in engine.zero1_buckets(weight_decay):
            n = se - ss
            assert n % (64 * self.world) == 0 and s <= ss <= se <= e, (name, s, e, ss, se)
            nb = n // self.world
            lo = ss + self.rank * nb
            b = {"name": name, "range": (s, e), "shard": (ss, se), "mine": (lo, lo + nb), "segs": [], "n": nb}
            if nb:
                b["master"] = flat_p[lo:lo + nb].float()                       # fp32 masters start as the bf16 values (the reference promotes
                b["m"] = torch.zeros(nb, dtype=torch.float32, device=dev)      # bf16-loaded weights the same way, util/tensor_type.py:60-66)
                b["v"] = torch.zeros(nb, dtype=torch.float32, device=dev)
                b["g"] = torch.zeros(nb, dtype=torch.float32, device=dev)
                b["out"] = torch.empty(nb, dtype=flat_p.dtype, device=dev)
                b["out"].copy_(flat_p[lo:lo + nb])
                for a, z, wd in segs:                                           # parameter pieces inside my slice (padding is never updated)
                    a2, z2 = max(a, lo), min(z, lo + nb)
                    if z2 > a2:
                        b["segs"].append((a2 - lo, z2 - lo, wd))
            self.buckets.append(b)
        self._by_name = {b["name"]: b for b in self.buckets}
        engine.on_layer_grads_ready = self._on_ready

    # torch.optim-like surface used by the trainer (LR schedule, zero_grad)
    @property
    def param_groups(self):
        return ([] if self.small is None else self.small.param_groups) + [self._group]

    def zero_grad(self, set_to_none: bool = True) -> None:
        if self.small is not None:
            self.small.zero_grad(set_to_none=set_to_none)

    def shard_bytes(self) -> int:
        return sum(b["n"] * (4 * 4 + b["out"].element_size()) for b in self.buckets if b["n"])

    # ------------------------------------------------------------------ gradient exchange (called as each bucket's backward completes)
    def _side_stream(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def _on_ready(self, name: str, start: int, end: int) -> None:
        if not self.enabled or end <= start:
            return
        flat = self.eng.flat_grads()
        if flat.is_cuda:
            st = self._side_stream(flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            st.wait_event(ev)
            with torch.cuda.stream(st):
                self._exchange(name, start, end)
        else:
            self._exchange(name, start, end)

    def _exchange(self, name: str, start: int, end: int) -> None:
        dist, flat = self.dist, self.eng.flat_grads()
        b = self._by_name.get(name)
        ss, se = (b["shard"] if b is not None else (start, start))
        # replicated pieces of the bucket (small fp32 parameters): plain average on every rank
        for a, z in ((start, ss), (se, end)):
            if z > a and self.world > 1 and not self.stub_collective:
                seg = flat[a:z]
                if self._avg:
                    dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    seg.mul_(1.0 / self.world)
                    dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
                self._wire_bytes += seg.numel() * 4
        if b is None or not b["n"]:
            return
        seg = flat[ss:se]
        wd = self.reduce_dtype or torch.float32
        scale = 1.0 if self._avg else 1.0 / self.world
        if wd != torch.float32:
            if self._wire is None or self._wire.numel() < seg.numel() or self._wire.dtype != wd:
                self._wire = torch.empty(max(x["shard"][1] - x["shard"][0] for x in self.buckets), dtype=wd, device=seg.device)
            src = self._wire[:seg.numel()]
            _scale_cast(seg, src, scale)                                        # one pass: pre-scale + wire cast
            if "gw" not in b:
                b["gw"] = torch.empty(b["n"], dtype=wd, device=seg.device)
            out = b["gw"]
        else:
            if scale != 1.0:
                seg.mul_(scale)
            src, out = seg, b["g"]
        self._wire_bytes += src.numel() * src.element_size()
        if self.world > 1 and not self.stub_collective:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            dist.reduce_scatter_tensor(out, src, op=op, group=self.group)
        else:
            out.copy_(src[self.rank * b["n"]:(self.rank + 1) * b["n"]])
        if out is not b["g"]:
            _scale_cast(out, b["g"], 1.0)                                       # widen my slice of the averaged gradient
        self._reduced.add(name)

    def finish(self) -> None:
        """Join the side stream; buckets the backward never announced (parameters without a gradient this step) are exchanged now."""
        for b in self.buckets:
            if b["n"] and b["name"] not in self._reduced and self.enabled:
                self._on_ready(b["name"], *b["range"])
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        self.wire_bytes_last_step, self._wire_bytes = self._wire_bytes, 0

    def any_rank(self, flag: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return flag
        if self.stub_collective:
            return flag
        t = flag.to(torch.float32).reshape(1)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t[0] > 0

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the AVERAGED gradient: sum over ranks of the squared slices + the replicated pieces (identical everywhere)."""
        flat = self.eng.flat_grads()
        loc = torch.zeros((), dtype=torch.float32, device=flat.device)
        rep = torch.zeros((), dtype=torch.float32, device=flat.device)
        for b in self.buckets:
            if b["n"]:
                loc = loc + torch.linalg.vector_norm(b["g"]) ** 2
            (s, e), (ss, se) = b["range"], b["shard"]
            for a, z in ((s, ss), (se, e)):
                if z > a:
                    rep = rep + torch.linalg.vector_norm(flat[a:z]) ** 2
        if self.world > 1 and not self.stub_collective:
            loc = loc.reshape(1)
            self.dist.all_reduce(loc, op=self.dist.ReduceOp.SUM, group=self.group)
            loc = loc[0]
        elif self.world > 1:
            loc = loc * self.world               # stubbed collectives (bench emulation of one shard): the other slices are assumed alike
        return torch.sqrt(loc + rep)

    def clip_coef(self, max_norm: float) -> Tuple[torch.Tensor, torch.Tensor]:
        norm = self.grad_norm()
        return norm, torch.clamp(max_norm / (norm + 1e-6), max=1.0).to(torch.float32)

    # ------------------------------------------------------------------ update
    @torch.no_grad()
    def step(self, grad_scale: Optional[torch.Tensor] = None, overlap: bool = False) -> None:
        self.step_count += 1
        g = self._group
        b1, b2 = g["betas"]
        flat_p = self.eng.flat_params()
        for b in self.buckets:
            if not b["n"]:
                continue
            for a, z, wd in b["segs"]:
                self.update(b["master"][a:z], b["g"][a:z], b["m"][a:z], b["v"][a:z], b["out"][a:z], g["lr"], b1, b2, g["eps"], wd, self.step_count, grad_scale)
            ss, se = b["shard"]
            if self.world > 1 and not self.stub_collective:
                self.dist.all_gather_into_tensor(flat_p[ss:se], b["out"], group=self.group)
            else:
                lo = b["mine"][0]
                flat_p[lo:lo + b["n"]].copy_(b["out"])
            self._wire_bytes += (se - ss) * b["out"].element_size()
        if self.small is not None:
            if grad_scale is not None and hasattr(self.small, "engine"):
                self.small.step(grad_scale=grad_scale)
            else:
                self.small.step()
        self._reduced.clear()
        self.eng.zero1_mark_fresh()                  # parameters changed in place: images are views, gradients restart from zero / store

    # ------------------------------------------------------------------ state (per-rank shard files, like FSDP's sharded optimizer state)
    def state_dict(self) -> Dict:
        return {"zero1": {"world": self.world, "rank": self.rank, "step": self.step_count,
                          "buckets": {b["name"]: {k: b[k].detach().cpu() for k in ("master", "m", "v")} for b in self.buckets if b["n"]}},
                "small": self.small.state_dict() if self.small is not None else None}

    def load_state_dict(self, sd: Dict) -> None:
        z = sd["zero1"]
        if z["world"] != self.world or z["rank"] != self.rank:
            raise RuntimeError(f"ZeRO-1 state of rank {z['rank']}/{z['world']} loaded into rank {self.rank}/{self.world}")
        self.step_count = int(z["step"])
        flat_p = self.eng.flat_params()
        for b in self.buckets:
            if not b["n"]:
                continue
            st = z["buckets"][b["name"]]
            for k in ("master", "m", "v"):
                b[k].copy_(st[k])
            b["out"].copy_(b["master"])
            ss, se = b["shard"]
            if self.world > 1:
                self.dist.all_gather_into_tensor(flat_p[ss:se], b["out"], group=self.group)
            else:
                flat_p[b["mine"][0]:b["mine"][1]].copy_(b["out"])
        if self.small is not None and sd.get("small") is not None:
            self.small.load_state_dict(sd["small"])
        self.eng.zero1_mark_fresh()
