"""Data-parallel fine-tuning over RCCL/xGMI (one process per GPU, ``torch.distributed`` backend "nccl").

Replaces the reference's FSDP(SHARD_GRAD_OP) wrap (main_finetune.py:241-263) by pure DP replicas:
  * ``GradReducer`` -- bucketed SUM all-reduce (pre-scaled by 1/world = FSDP's gradient average) of the
    training engine's flat fp32 gradient buffer.  Buckets are the engine's per-layer ranges; a bucket is
    handed to RCCL on a dedicated side HIP stream the moment that layer's backward has produced it
    (``TrainEngine.on_layer_grads_ready``), so the transfer overlaps the back-propagation of the earlier
    layers.  Gradient accumulation = ``reducer.enabled = False`` on non-boundary micro-steps (the
    reference's ``no_sync``, util/misc.py:311-313).
  * ``FinetuneDistSampler`` -- the reference's sharding contract (data/alpaca.py:246-328).
  * ``clip_grad_norm`` -- global L2 clip with the reference's coefficient (util/clip_grad.py:187-193); no
    collective is needed in pure DP because post-reduce gradients are identical on every rank.
xGMI is point-to-point: large (per-layer, ~0.8 GB fp32) buckets keep every link busy with few launches.
"""
from __future__ import annotations

import copy
from typing import Iterator, List, Optional

import numpy as np
import torch


class GradReducer:
    def __init__(self, engine, dist, reduce_dtype: Optional[torch.dtype] = None, group=None):
        self.eng = engine
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else dist.get_world_size()
        self.enabled = True
        self.reduce_dtype = reduce_dtype          # None: reduce the fp32 buffer in place; bf16: halve wire bytes
        self._pending: List = []
        self._stream = None
        engine.on_layer_grads_ready = self._on_ready

    def _side_stream(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def _on_ready(self, name: str, start: int, end: int) -> None:
        if not self.enabled or self.world == 1:
            return
        seg = self.eng.flat_grads()[start:end]
        if seg.is_cuda:
            st = self._side_stream(seg.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(seg.device))
            st.wait_event(ev)                      # the bucket's producers have finished
            with torch.cuda.stream(st):
                self._launch(seg)
        else:
            self._launch(seg)

    def _launch(self, seg: torch.Tensor) -> None:
        seg.mul_(1.0 / self.world)
        if self.reduce_dtype is not None and self.reduce_dtype != seg.dtype:
            low = seg.to(self.reduce_dtype)
            work = self.dist.all_reduce(low, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, seg, low))
        else:
            work = self.dist.all_reduce(seg, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, None, None))

    def finish(self) -> None:
        """Join every outstanding bucket (call before clipping / optimizer.step)."""
        for work, seg, low in self._pending:
            if seg is not None and seg.is_cuda:
                with torch.cuda.stream(self._stream):     # the copy-back runs on the side stream: THAT stream must wait for the collective
                    work.wait()
                    seg.copy_(low)
            else:
                work.wait()
                if seg is not None:
                    seg.copy_(low)
        self._pending.clear()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def reduce_all_now(self) -> None:
        """Reduce every bucket (used at an accumulation boundary when earlier micro-steps ran with enabled=False)."""
        was = self.enabled
        self.enabled = True
        for name, st, en in self.eng.grad_ranges():
            self._on_ready(name, st, en)
        self.enabled = was
        self.finish()


def clip_grad_norm(parameters, max_norm: float, flat: Optional[torch.Tensor] = None, defer: bool = False):
    """util/clip_grad.py:59-210 for pure DP: fp32 global L2 norm, coef = max_norm/(norm+1e-6) clamped to 1, every gradient
    multiplied by coef.  Returns the norm.

    ``flat``: the training engine's flat gradient buffer (``TrainEngine.flat_grads()``); when every gradient is a view into
    it the norm is ONE reduction over the buffer instead of one per parameter (its padding is zero).  ``defer=True`` returns
    ``(norm, coef)`` and leaves the gradients untouched: the caller hands ``coef`` (a device scalar) to
    ``FusedAdamW.step(grad_scale=coef)``, which multiplies as it reads -- no separate read + write of every gradient."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        z = torch.zeros(())
        return (z, torch.ones(())) if defer else z
    in_flat = flat is not None and flat.dtype == torch.float32 and all(
        g.dtype == torch.float32 and g.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for g in grads)
    if in_flat:
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
    """data/alpaca.py:246-328: every GLOBAL batch (batch_size x replicas x acc_grad consecutive indices of one
    group) is homogeneous in data type; global batches are shuffled with default_rng(seed + epoch); rank r takes
    rows [r*bs + k*replicas*bs, +bs); resumable through set_epoch(epoch, start_iter)."""

    def __init__(self, dataset, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True,
                 seed: int = 0, batch_size=None, acc_grad: int = 1) -> None:
        if num_replicas is None or rank is None or rank >= num_replicas or rank < 0:
            raise ValueError(f"Invalid num_replicas ({num_replicas}) or rank ({rank})")
        assert batch_size is not None
        self.batch_size, self.dataset, self.num_replicas, self.rank, self.acc_grad = batch_size, dataset, num_replicas, rank, acc_grad
        self.epoch, self.start_iter = 0, 0
        group_indices = dataset.groups()
        global_bsz = batch_size * num_replicas * acc_grad
        group_indices = [ind[: len(ind) // global_bsz * global_bsz] for ind in group_indices]
        group_n_batch = [len(g) // batch_size for g in group_indices]
        assert all(n % num_replicas == 0 for n in group_n_batch)
        n_total_batch = sum(group_n_batch)
        self.group_indices = group_indices
        self.total_size = n_total_batch * batch_size
        self.num_samples = self.total_size // num_replicas
        self.shuffle, self.seed = shuffle, seed

    def __iter__(self) -> Iterator:
        gbs = self.batch_size * self.num_replicas * self.acc_grad
        groups = copy.deepcopy(self.group_indices)
        if self.shuffle:
            rng = np.random.default_rng(self.seed + self.epoch)
            batches = [g[i:i + gbs] for g in groups for i in range(0, len(g), gbs)]
            rng.shuffle(batches)
            indices = [i for b in batches for i in b]
        else:
            indices = [i for g in groups for i in g]
        assert len(indices) == self.total_size
        own: List[int] = []
        for start in range(self.rank * self.batch_size, len(indices), self.num_replicas * self.batch_size):
            own += indices[start:start + self.batch_size]
        assert len(own) == self.num_samples
        own = [] if self.start_iter * self.batch_size > len(own) else own[self.start_iter * self.batch_size:]
        return iter(own)

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int, start_iter: int = 0) -> None:
        self.epoch, self.start_iter = epoch, start_iter
