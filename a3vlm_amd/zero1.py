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

``step(overlap=True)`` (the trainer's default under DP): bucket i's all-gather is issued on a side stream the moment its AdamW launches
are queued -- it runs under the AdamW of buckets i+1.. and, for the tail, under the next forward, which waits bucket by bucket on first
use (``engine.await_weights``).  Same values as the serial step, bit for bit (``tests/test_zero1_cpu.py``).

Checkpoints: ``state_dict()`` is the rank's slice (cheap, rank-local); ``full_state_dict()`` gathers the slices bucket by bucket to rank 0
and is what ``checkpoint.save_checkpoint`` writes as ``consolidated.00-of-01.optimizer.pth`` -- one world-size-independent file, as the
reference's ``FSDP.full_optim_state_dict`` path does (util/misc.py:395-403); ``load_state_dict`` takes either form and re-slices the full
one for any world size.  A step skipped by a negative ``grad_scale`` still advances ``step_count`` (the bias correction), exactly as
``FusedAdamW`` advances every parameter's ``step``: both counts are saved together, so they cannot disagree after a resume.

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
        self._gather_stream = None
        self._pending: Dict[str, object] = {}     # bucket -> all-gather still in flight (overlapped step): a CUDA event or a dist Work
        flat_p = engine.flat_params()
        dev = flat_p.device
        self.buckets: List[dict] = []
        for name, s, e, ss, se, segs in engine.zero1_buckets(weight_decay):
            n = se - ss
            assert n % (64 * self.world) == 0 and s <= ss <= se <= e, (name, s, e, ss, se)
            nb = n // self.world
            lo = ss + self.rank * nb
            b = {"name": name, "range": (s, e), "shard": (ss, se), "mine": (lo, lo + nb), "segs": [], "n": nb,
                 "used": max([z for _, z, _ in segs], default=ss) - ss}       # parameters end here; the rest of the span is padding to 64 N
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
        self._reduced.add(name)
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

    def finish(self) -> None:
        """Join the side stream; buckets the backward never announced (parameters without a gradient this step) are exchanged now."""
        if self.enabled:
            for name, s, e in self._ranges():      # replicated-only buckets (projector, tags) included: every rank hands over the same set
                if name not in self._reduced:
                    self._on_ready(name, s, e)
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
        # the replicated pieces: EVERY gradient range of the engine minus the sharded spans -- also the buckets that have no sharded
        # span at all (vision_proj: projector matrices and image tags; round 4 left them out of the norm)
        for name, s, e in self._ranges():
            b = self._by_name.get(name)
            ss, se = b["shard"] if b is not None and b["n"] else (s, s)
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
    def _ranges(self):
        """(name, start, end) of every gradient bucket of the engine (sharded or not)."""
        if hasattr(self.eng, "grad_ranges"):
            return list(self.eng.grad_ranges())
        return [(b["name"],) + tuple(b["range"]) for b in self.buckets]

    def _order(self) -> List[dict]:
        """Buckets in the order the next forward first touches them (the tail of the gathers hides under its first layers)."""
        if hasattr(self.eng, "forward_order"):
            rank = {it[0]: i for i, it in enumerate(self.eng.forward_order())}
            return sorted(self.buckets, key=lambda b: rank.get(b["name"], len(rank)))
        return list(self.buckets)

    def await_params(self, name: str) -> None:
        """The caller's stream (or the host, on CPU) waits for this bucket's all-gather of an overlapped step."""
        h = self._pending.pop(name, None)
        if h is None:
            return
        if isinstance(h, torch.cuda.Event):
            torch.cuda.current_stream().wait_event(h)
        else:
            h.wait()

    def sync_params(self) -> None:
        """Every all-gather of an overlapped step has landed (the optimizer's own handles and the events handed to the engine)."""
        for name in list(self._pending):
            self.await_params(name)
        if hasattr(self.eng, "sync_optimizer"):
            self.eng.sync_optimizer()

    @torch.no_grad()
    def step(self, grad_scale: Optional[torch.Tensor] = None, overlap: bool = False) -> None:
        self.sync_params()                               # gathers of a previous overlapped step nobody consumed
        self.step_count += 1
        g = self._group
        b1, b2 = g["betas"]
        flat_p = self.eng.flat_params()
        real = self.world > 1 and not self.stub_collective
        overlap = bool(overlap) and real
        ready = getattr(self.eng, "_weights_ready", None)
        for b in (self._order() if overlap else self.buckets):
            if not b["n"]:
                continue
            for a, z, wd in b["segs"]:
                self.update(b["master"][a:z], b["g"][a:z], b["m"][a:z], b["v"][a:z], b["out"][a:z], g["lr"], b1, b2, g["eps"], wd, self.step_count, grad_scale)
            ss, se = b["shard"]
            if not real:
                lo = b["mine"][0]
                flat_p[lo:lo + b["n"]].copy_(b["out"])
            elif not overlap:
                self.dist.all_gather_into_tensor(flat_p[ss:se], b["out"], group=self.group)
            elif flat_p.is_cuda:
                # side stream: waits for this bucket's AdamW launches only; the compute stream goes on with the next bucket
                if self._gather_stream is None:
                    self._gather_stream = torch.cuda.Stream(device=flat_p.device)
                gs = self._gather_stream
                gs.wait_event(torch.cuda.current_stream().record_event())
                with torch.cuda.stream(gs):
                    self.dist.all_gather_into_tensor(flat_p[ss:se], b["out"], group=self.group)
                    ev = gs.record_event()
                if ready is not None:
                    ready[b["name"]] = ev                # TrainEngine.await_weights / sync_optimizer wait on it
                else:
                    self._pending[b["name"]] = ev
            else:
                self._pending[b["name"]] = self.dist.all_gather_into_tensor(flat_p[ss:se], b["out"], group=self.group, async_op=True)
            self._wire_bytes += (se - ss) * b["out"].element_size()
        if self.small is not None:
            if grad_scale is not None and hasattr(self.small, "engine"):
                self.small.step(grad_scale=grad_scale)
            else:
                self.small.step()
        self._reduced.clear()
        self.eng.zero1_mark_fresh()                  # parameters changed in place: images are views, gradients restart from zero / store

    def resync_from_params(self) -> None:
        """Re-seed the fp32 masters (and the bf16 sink) from the CURRENT parameter values -- after model weights were loaded behind the
        optimizer's back (a resume without optimizer state: the constructor's snapshot would otherwise overwrite the loaded weights at
        the first all-gather).  Moments are kept."""
        self.sync_params()
        flat_p = self.eng.flat_params()
        for b in self.buckets:
            if b["n"]:
                lo = b["mine"][0]
                b["master"].copy_(flat_p[lo:lo + b["n"]])
                b["out"].copy_(flat_p[lo:lo + b["n"]])

    # ------------------------------------------------------------------ state
    def state_dict(self) -> Dict:
        """This rank's slice (rank-local, no collective)."""
        self.sync_params()
        return {"zero1": {"world": self.world, "rank": self.rank, "step": self.step_count,
                          "buckets": {b["name"]: {k: b[k].detach().cpu() for k in ("master", "m", "v")} for b in self.buckets if b["n"]}},
                "small": self.small.state_dict() if self.small is not None else None}

    def full_state_dict(self) -> Optional[Dict]:
        """COLLECTIVE: the whole sharded state gathered bucket by bucket; rank 0 returns it (CPU tensors over each bucket's sharded span,
        padding included), the other ranks return None.  World-size independent: any DP size can load it."""
        self.sync_params()
        full = {}
        for b in self.buckets:
            if not b["n"]:
                continue
            ent = {}
            for k in ("master", "m", "v"):
                if self.world > 1 and not self.stub_collective:
                    # gathered onto rank 0 only (an all-gather put n x world fp32 on EVERY rank -- 3 x 52 GB at 13B -- for a
                    # file one rank writes); rank 0 holds one bucket's n x world at a time
                    parts = [torch.empty_like(b[k]) for _ in range(self.world)] if self.rank == 0 else None
                    dst = self.dist.get_global_rank(self.group, 0) if self.group is not None and hasattr(self.dist, "get_global_rank") else 0
                    self.dist.gather(b[k].contiguous(), parts, dst=dst, group=self.group)
                    if self.rank == 0:
                        ent[k] = torch.cat([q.cpu() for q in parts])
                    del parts
                elif self.rank == 0:
                    ent[k] = b[k].detach().cpu()
            if self.rank == 0:
                full[b["name"]] = ent
        if self.rank != 0:
            return None
        # (a span is padded to a multiple of 64 x world at its END only: the first `used` elements have the same layout at every DP size)
        return {"zero1_full": {"step": self.step_count, "world": self.world, "used": {b["name"]: b["used"] for b in self.buckets if b["n"]},
                               "buckets": full},
                "small": self.small.state_dict() if self.small is not None else None}

    def load_state_dict(self, sd: Dict) -> None:
        self.sync_params()
        flat_p = self.eng.flat_params()
        if "zero1_full" in sd:
            # every rank reads ITS slice of the same (mmap-ed) file; the parameters are then rebuilt by the all-gather of the slices' compute
            # images, exactly as a step does (reading every master on every rank was 52 GB of host reads + H2D per rank at 13B)
            z = sd["zero1_full"]
            self.step_count = int(z["step"])
            for b in self.buckets:
                if not b["n"]:
                    continue
                st, (ss, se) = z["buckets"][b["name"]], b["shard"]
                n_st = st["master"].numel()
                if z["used"][b["name"]] != b["used"] or n_st < b["used"]:
                    raise RuntimeError(f"ZeRO-1 state of bucket {b['name']}: {z['used'][b['name']]} parameter elements saved, {b['used']} here")
                lo = self.rank * b["n"]
                hi = min(lo + b["n"], n_st)                 # the saving run's span may be shorter or longer: only padding differs
                for k in ("master", "m", "v"):
                    if hi > lo:
                        b[k][:hi - lo].copy_(st[k][lo:hi])
                b["out"].copy_(b["master"])
                if self.world > 1 and not self.stub_collective:
                    self.dist.all_gather_into_tensor(flat_p[ss:se], b["out"], group=self.group)
                elif self.world > 1:                        # one-GPU emulation of a shard (bench): no peers to gather from
                    nc = min(n_st, se - ss)
                    flat_p[ss:ss + nc].copy_(st["master"][:nc])
                else:
                    flat_p[b["mine"][0]:b["mine"][1]].copy_(b["out"])
        else:
            z = sd["zero1"]
            if z["world"] != self.world or z["rank"] != self.rank:
                raise RuntimeError(f"ZeRO-1 slice state of rank {z['rank']}/{z['world']} loaded into rank {self.rank}/{self.world} "
                                   "(load the consolidated optimizer file to change the DP size)")
            self.step_count = int(z["step"])
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
