"""Data-parallel fine-tuning over RCCL/xGMI (one process per GPU, ``torch.distributed`` backend "nccl").

Replaces the reference's FSDP(SHARD_GRAD_OP) wrap (main_finetune.py:241-263) by pure DP replicas:
  * ``GradReducer`` -- bucketed gradient AVERAGE (= FSDP's gradient reduction) of the training engine's flat fp32 gradient
    buffer.  Buckets are the engine's per-layer ranges; a bucket is handed to RCCL on a dedicated side HIP stream the moment
    that layer's backward has produced it (``TrainEngine.on_layer_grads_ready``), so the transfer overlaps the
    back-propagation of the earlier layers.  Gradient accumulation = ``reducer.enabled = False`` on every micro-step but the
    LAST of a cycle (the reference's ``no_sync``, util/misc.py:311-313): the flat buffer accumulates in place, so on the last
    micro-step a layer's bucket already holds the cycle's sum when its backward finishes, and the reduction overlaps that
    backward exactly as without accumulation.
    Passes over the gradient buffer: fp32 wire = none besides the collective (``ReduceOp.AVG`` on RCCL); bf16 wire (the
    reference's FSDP ``reduce_dtype``) = one fused scale+cast kernel into a persistent bf16 bucket before, one widening cast
    after (``a3v_scale_cast``).
  * ``FinetuneDistSampler`` -- the reference's sharding contract (data/alpaca.py:246-328), pinned by
    ``tests/golden/host_tiny.json``.
  * ``clip_grad_norm`` -- global L2 clip with the reference's coefficient (util/clip_grad.py:187-193); no collective is
    needed in pure DP because post-reduce gradients are identical on every rank.
xGMI is point-to-point: large (per-layer, ~0.8 GB fp32) buckets keep every link busy with few launches.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch


def _scale_cast(src: torch.Tensor, dst: torch.Tensor, scale: float) -> None:
    """dst = (dst.dtype)(src * scale) in ONE pass (HIP kernel on device tensors; torch on the CPU tensors of the gloo tests)."""
    if src.is_cuda:
        from . import ops
        ops.scale_cast(src, dst, scale)
    else:
        torch.mul(src, scale, out=dst) if dst.dtype == src.dtype else dst.copy_(src * scale)


class _DoneWork:
    def wait(self):
        return True


class _NoCollective:
    """``GradReducer.stub_collective``: the process group's interface with ``all_reduce`` doing nothing (bench.py's
    exposed-all-reduce measurement; never used by the trainer)."""

    def __init__(self, dist):
        self.ReduceOp = dist.ReduceOp

    def all_reduce(self, *a, **k):
        return _DoneWork()


class GradReducer:
    def __init__(self, engine, dist, reduce_dtype: Optional[torch.dtype] = None, group=None, reduce_single_rank: bool = False):
        self.eng = engine
        self.reduce_single_rank = reduce_single_rank      # tests: run the whole bucket path on a one-rank group
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else dist.get_world_size()
        self.enabled = True
        self.reduce_dtype = reduce_dtype          # None: reduce the fp32 buffer in place; bf16: halve the wire bytes
        self._pending: List = []
        self._keys: List[Tuple[int, int]] = []
        self._stream = None
        self._wire: Dict[Tuple[int, int], torch.Tensor] = {}      # persistent low-precision buckets, keyed by flat range
        try:                                       # RCCL averages in the collective; gloo (CPU tests) only sums
            self._avg = dist.get_backend(group) == "nccl" and hasattr(dist.ReduceOp, "AVG")
        except Exception:
            self._avg = False
        self.sumsq: Optional["GradSquareSums"] = None       # set by GradSquareSums(engine, reducer=self): sums over the REDUCED buckets
        self.stub_collective = False               # bench: everything but the collective itself (exposed all-reduce time = the difference)
        self.wire_bytes_last_step = 0              # bytes handed to the collective since the last finish()
        self._wire_bytes = 0
        engine.on_layer_grads_ready = self._on_ready

    def _side_stream(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def _on_ready(self, name: str, start: int, end: int) -> None:
        if not self.enabled or (self.world == 1 and not self.reduce_single_rank) or end <= start:      # (empty: all its parameters frozen)
            return
        seg = self.eng.flat_grads()[start:end]
        if seg.is_cuda:
            st = self._side_stream(seg.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(seg.device))
            st.wait_event(ev)                      # the bucket's producers have finished
            with torch.cuda.stream(st):
                self._launch(seg, (start, end))
        else:
            self._launch(seg, (start, end))

    def _launch(self, seg: torch.Tensor, key: Tuple[int, int]) -> None:
        dist = self.dist
        self._keys.append(key)
        if self.stub_collective:
            dist = _NoCollective(dist)
        if self.reduce_dtype is not None and self.reduce_dtype != seg.dtype:
            low = self._wire.get(key)
            if low is None or low.device != seg.device:
                low = self._wire[key] = torch.empty(seg.numel(), dtype=self.reduce_dtype, device=seg.device)
            _scale_cast(seg, low, 1.0 if self._avg else 1.0 / self.world)
            self._wire_bytes += low.numel() * low.element_size()
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            work = dist.all_reduce(low, op=op, group=self.group, async_op=True)
        elif self._avg:
            low = None
            self._wire_bytes += seg.numel() * seg.element_size()
            work = dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            low = None
            self._wire_bytes += seg.numel() * seg.element_size()
            seg.mul_(1.0 / self.world)
            work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if seg.is_cuda:
            # on a device stream Work.wait() only makes THIS (side) stream wait for the collective: the widening pass and the
            # clip's sums of squares are queued behind it right away and overlap the backward like the collective itself
            work.wait()
            if low is not None:
                _scale_cast(low, seg, 1.0)
            if self.sumsq is not None:
                self.sumsq.add(seg, key)
            self._pending.append((None, None, None))
        else:
            self._pending.append((work, seg if low is not None else None, low))

    def finish(self) -> None:
        """Join every outstanding bucket (call before clipping / optimizer.step)."""
        host_side = False
        for work, seg, low in self._pending:
            if work is None:                       # device bucket: everything is already queued on the side stream
                continue
            host_side = True
            work.wait()
            if seg is not None:
                seg.copy_(low)
        if self.sumsq is not None and host_side:
            eng_flat = self.eng.flat_grads()
            for key in self._keys:
                self.sumsq.add(eng_flat[key[0]:key[1]], key)
        self._pending.clear()
        self._keys.clear()
        self.wire_bytes_last_step, self._wire_bytes = self._wire_bytes, 0
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def any_rank(self, flag: torch.Tensor) -> torch.Tensor:
        """Logical OR of a per-rank boolean device scalar over all ranks (one tiny MAX all-reduce, no host read).  The trainer's
        non-finite flag must be GLOBAL before the optimizer step: a NaN loss on rank A reaches rank B through the averaged
        gradients, so B has to skip the same update A skips (the reference exits before backward on the rank that sees it,
        engine_finetune.py:54-58, and the job dies with it)."""
        if self.world == 1:
            return flag
        t = flag.to(torch.float32).reshape(1)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t[0] > 0

    def reduce_all_now(self) -> None:
        """Reduce every bucket at once (no overlap): for callers that ran the whole cycle with ``enabled = False``."""
        was = self.enabled
        self.enabled = True
        for name, st, en in self.eng.grad_ranges():
            self._on_ready(name, st, en)
        self.enabled = was
        self.finish()


class GradSquareSums:
    """Sum of squares of the flat gradient buffer, bucket by bucket, for the global-norm clip (reference ``util/clip_grad.py:59-210``).

    Without it the clip reads the whole buffer (27 GB at 7B) after the backward, serially, before the optimizer can start.  Here
    every bucket's partial sums are taken the moment the bucket is final -- on a side stream while the MFMA-bound backward of the
    earlier layers still runs (one rank), or right behind the bucket's all-reduce on the reducer's stream (DP: the norm is over the
    REDUCED gradients, identical on every rank, so clipping needs no collective).  Buckets that were not announced in a step (no
    image in the batch, ...) are summed at ``norm()`` time.  ``enabled`` follows the reducer's convention: False on the micro-steps
    of an accumulation window that do not end it."""

    def __init__(self, engine, reducer: Optional[GradReducer] = None, fused: bool = True):
        self.eng, self.enabled = engine, True
        self.ranges, self.slot, self.part = None, None, None      # laid out at the first bucket (the engine allocates its buffer lazily)
        self.seen = set()
        self._stream = None
        # one rank: the big weight-gradient GEMMs add up the squares of what they store in their own epilogue (a3v_gemm_tn_sumsq), so
        # only the small rest of a bucket (norm weights, ...) is read again; under DP the sums must be over the REDUCED gradients
        self._fused_buf: Dict[int, torch.Tensor] = {}             # grad data_ptr -> its slots (views of _fused_all)
        self._fused_all: Optional[torch.Tensor] = None
        self._fused_used = 0
        self._covered: List[Tuple[int, int]] = []                 # flat ranges whose sums came out of a GEMM this step
        self.reducer, self._fused, self._chained, self._attached = reducer, fused, None, False
        self.attach()

    def attach(self) -> None:
        """Link into the gradient path: behind the reducer's buckets (DP: sums over the REDUCED gradients, and the engine's fused
        GEMM-epilogue sums are switched OFF because they would be over the local ones), or onto the engine's bucket hook + fused
        epilogue sums (one rank).  Idempotent; ``detach()`` undoes it (the trainer detaches at the end of every epoch, so a later
        epoch with another reducer -- or none -- never inherits this object's hooks)."""
        if self._attached:
            return
        engine, reducer = self.eng, self.reducer
        self.seen.clear()
        self._covered.clear()
        if reducer is not None and (reducer.world > 1 or reducer.reduce_single_rank):
            reducer.sumsq = self
            if hasattr(engine, "sumsq_sink"):
                engine.sumsq_sink = None
        else:
            self._chained = engine.on_layer_grads_ready
            engine.on_layer_grads_ready = self._on_ready
            if self._fused and hasattr(engine, "sumsq_sink"):
                engine.sumsq_sink = self
        self._attached = True

    def detach(self) -> None:
        if not self._attached:
            return
        engine, reducer = self.eng, self.reducer
        if reducer is not None and reducer.sumsq is self:
            reducer.sumsq = None
        if engine.on_layer_grads_ready == self._on_ready:
            engine.on_layer_grads_ready = self._chained
        if getattr(engine, "sumsq_sink", None) is self:
            engine.sumsq_sink = None
        self._chained = None
        self.seen.clear()
        self._covered.clear()
        self._attached = False

    def begin_backward(self) -> None:
        """A new backward starts (called by the engine): whatever an earlier backward left behind without a ``norm()`` -- covered
        ranges of fused slots, buckets marked seen -- belongs to another step."""
        self._covered.clear()
        self.seen.clear()

    def _layout(self):
        if self.part is None:
            from . import ops
            flat = self.eng.flat_grads()
            self.ranges = [(st, en) for _, st, en in self.eng.grad_ranges() if en > st]
            self.slot = {key: i for i, key in enumerate(self.ranges)}
            self.part = torch.zeros(len(self.ranges), ops.SUMSQ_SLOTS if flat.is_cuda else 1, dtype=torch.float32, device=flat.device)

    def wgrad_slots(self, grad: torch.Tensor, M: int, N: int) -> Optional[torch.Tensor]:
        """Called by the engine for a weight-gradient GEMM that writes ``grad`` [M, N] (a view of the flat buffer): the slots its
        epilogue fills, or None when sums are not wanted for this micro-step."""
        if not self.enabled or not grad.is_cuda:
            return None
        from . import ops
        flat = self.eng.flat_grads()
        off = (grad.data_ptr() - flat.data_ptr()) // 4
        if off < 0 or off + grad.numel() > flat.numel():
            return None
        buf = self._fused_buf.get(grad.data_ptr())
        if buf is None:
            n = ops.gemm_tn_sumsq_slots(M, N)
            if self._fused_all is None:
                self._fused_all = torch.zeros(4 << 20, dtype=torch.float32, device=grad.device)       # 16 MB: ~10x the 7B need
            if self._fused_used + n > self._fused_all.numel():
                return None
            buf = self._fused_all[self._fused_used:self._fused_used + n]
            self._fused_used += n
            self._fused_buf[grad.data_ptr()] = buf
        self._covered.append((off, off + grad.numel()))
        return buf

    def _uncovered(self, start: int, end: int) -> List[Tuple[int, int]]:
        """[start, end) minus the ranges whose sums a GEMM produced this step."""
        out, cur = [], start
        for a, b in sorted(r for r in self._covered if r[1] > start and r[0] < end):
            if a > cur:
                out.append((cur, min(a, end)))
            cur = max(cur, b)
        if cur < end:
            out.append((cur, end))
        return out

    def add(self, seg: torch.Tensor, key: Tuple[int, int]) -> None:
        """Partial sums of one bucket on the CURRENT stream."""
        self._layout()
        i = self.slot.get(key)
        if i is None:
            return
        if key in self.seen:                       # the same bucket again without a norm() in between: a new step began
            self.seen.clear()
        if seg.is_cuda:
            from . import ops
            rest = self._uncovered(key[0], key[1]) if self._covered else [key]
            if rest == [key]:
                ops.sumsq_partials(seg, self.part[i])
            else:
                # the bucket's big matrices were summed by their GEMMs: read only what is left (16-byte aligned pieces; a bucket has a
                # handful of them), accumulated into the bucket's first slot row through a scratch row
                self.part[i].zero_()
                flat = self.eng.flat_grads()
                for a_, b_ in rest:
                    if b_ > a_:
                        piece = flat[a_:b_]
                        if piece.data_ptr() % 16 == 0:
                            ops.sumsq_partials(piece, self._scratch_row())
                            self.part[i] += self._scratch
                        else:
                            self.part[i, 0] += (piece.double() ** 2).sum().float()
        else:
            self.part[i, 0] = (seg.double() ** 2).sum().float()
        self.seen.add(key)

    def _scratch_row(self) -> torch.Tensor:
        if getattr(self, "_scratch", None) is None:
            self._scratch = torch.zeros_like(self.part[0])
        return self._scratch

    def _on_ready(self, name: str, start: int, end: int) -> None:
        if self._chained is not None:
            self._chained(name, start, end)
        if not self.enabled or end <= start:
            return
        seg = self.eng.flat_grads()[start:end]
        if seg.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=seg.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(seg.device))
            self._stream.wait_event(ev)
            with torch.cuda.stream(self._stream):
                self.add(seg, (start, end))
        else:
            self.add(seg, (start, end))

    def norm(self) -> torch.Tensor:
        """sqrt(sum over all buckets) as a device scalar on the current stream; resets the per-step bookkeeping."""
        self._layout()
        flat = self.eng.flat_grads()
        if self._stream is not None:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        for key in self.ranges:
            if key not in self.seen:
                self.add(flat[key[0]:key[1]], key)
        self.seen.clear()
        total = self.part.sum(dtype=torch.float32)
        if self._covered:
            total = total + self._fused_all[:self._fused_used].sum(dtype=torch.float32)
            self._fused_all[:self._fused_used].zero_()          # the next step's GEMMs write (only) the slots of tiles inside their output
            self._covered.clear()
        return total.sqrt()


def clip_grad_norm(parameters, max_norm: float, flat: Optional[torch.Tensor] = None, defer: bool = False,
                   sumsq: Optional[GradSquareSums] = None):
    """util/clip_grad.py:59-210 for pure DP: fp32 global L2 norm, coef = max_norm/(norm+1e-6) clamped to 1, every gradient
    multiplied by coef.  Returns the norm.

    ``flat``: the training engine's flat gradient buffer (``TrainEngine.flat_grads()``); when every gradient is a view into
    it the norm is ONE reduction over the buffer instead of one per parameter (its padding is zero).  ``defer=True`` returns
    ``(norm, coef)`` and leaves the gradients untouched: the caller hands ``coef`` (a device scalar) to
    ``FusedAdamW.step(grad_scale=coef)``, which multiplies as it reads -- no separate read + write of every gradient.
    ``sumsq``: a ``GradSquareSums`` fed during the backward; the norm is then the square root of its partial sums."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        z = torch.zeros(())
        return (z, torch.ones(())) if defer else z
    in_flat = flat is not None and flat.dtype == torch.float32 and all(
        g.dtype == torch.float32 and g.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for g in grads)
    if in_flat and sumsq is not None:
        norm = sumsq.norm()                    # per-bucket partial sums taken while the backward ran (GradSquareSums)
    elif in_flat:
        norm = torch.linalg.vector_norm(flat)
    else:
        norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float()) for g in grads]))
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    if defer:
        return norm, coef.to(torch.float32)
    if in_flat:
        flat.mul_(coef)
    else:
        for g in grads:
            g.mul_(coef.to(g.dtype))
    return norm


class FinetuneDistSampler(torch.utils.data.Sampler):
    """Which dataset indices a DP rank sees, in which order (the sharding contract of the path's training side).

    * A *global batch* = ``batch_size * num_replicas * acc_grad`` consecutive members of ONE dataset group (``dataset.groups()``),
      so every optimizer step is homogeneous in data type; a group's ragged tail is dropped.
    * Per epoch the global batches (not the samples) are permuted with ``numpy.random.default_rng(seed + epoch)``.
    * Inside the flattened order, micro-batch ``k`` of rank ``r`` is the ``batch_size`` indices starting at
      ``(k * num_replicas + r) * batch_size``.
    * ``set_epoch(epoch, start_iter)`` resumes inside an epoch: the first ``start_iter`` micro-batches are skipped
      (``__len__`` keeps reporting the full epoch, which the trainer's LR schedule relies on)."""

    def __init__(self, dataset, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True,
                 seed: int = 0, batch_size=None, acc_grad: int = 1) -> None:
        if num_replicas is None or rank is None or not 0 <= rank < num_replicas:
            raise ValueError(f"Invalid num_replicas ({num_replicas}) or rank ({rank})")
        assert batch_size is not None
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.batch_size, self.acc_grad, self.shuffle, self.seed = batch_size, acc_grad, shuffle, seed
        self.epoch, self.start_iter = 0, 0
        per_step = self._per_step
        self.group_indices = [list(members[:len(members) - len(members) % per_step]) for members in dataset.groups()]
        self.total_size = sum(len(members) for members in self.group_indices)
        self.num_samples = self.total_size // num_replicas

    @property
    def _per_step(self) -> int:
        return self.batch_size * self.num_replicas * self.acc_grad

    def _epoch_order(self) -> np.ndarray:
        """All kept indices of the epoch as an array [global batch][per_step]."""
        per_step = self._per_step
        table = np.asarray([i for members in self.group_indices for i in members], dtype=np.int64).reshape(-1, per_step)
        if self.shuffle and len(table):
            order = list(range(len(table)))
            np.random.default_rng(self.seed + self.epoch).shuffle(order)     # a Python list: the generic Fisher-Yates path
            table = table[order]
        return table

    def __iter__(self) -> Iterator[int]:
        table = self._epoch_order()
        assert table.size == self.total_size
        # [global batch, micro-step of the accumulation cycle, rank, sample of the micro-batch] -> this rank's column
        mine = table.reshape(-1, self.acc_grad, self.num_replicas, self.batch_size)[:, :, self.rank, :].reshape(-1)
        assert mine.size == self.num_samples
        return iter(mine[self.start_iter * self.batch_size:].tolist())

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int, start_iter: int = 0) -> None:
        self.epoch, self.start_iter = epoch, start_iter
