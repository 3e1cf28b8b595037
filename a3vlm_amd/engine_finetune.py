"""``train_one_epoch`` with the semantics of the reference's engine (model/accessory/engine_finetune.py:13-105)
on top of the HIP training engine and the DP reducer.

Kept: 3/4/5-tuple batches, LR set per accumulation boundary by FRACTIONAL epoch, ``loss / accum_iter``, clip (global L2,
reference coefficient) then ``optimizer.step()`` only on boundary micro-steps, ``zero_grad(set_to_none=True)``, gradient
all-reduce skipped on non-boundary micro-steps (the reference's ``no_sync``, util/misc.py:311-313), a non-finite loss ends
the run with ``sys.exit(1)`` and never reaches the weights.

Changed:
  * no per-iteration ``cuda.synchronize()`` and no per-meter all-reduce with host sync every iteration
    (engine_finetune.py:79,87-91): the loss is read back once per ``print_freq`` boundary steps;
  * the reference reads ``loss.item()`` every micro-step and exits BEFORE backward (engine_finetune.py:54-58).  Here the check
    stays on the device: a sticky flag ``bad |= !isfinite(loss) | !isfinite(grad_norm)`` is folded into the optimizer's
    gradient coefficient (negative coefficient = ``a3v_adamw_scaled`` does nothing), so a bad step is a no-op on masters,
    moments and bf16 weight images, and the host reads the flag at every logging boundary, before every checkpoint callback
    and at the end of the epoch -- exiting then, before anything poisoned can be written;
  * under DP the flag is made GLOBAL (``GradReducer.any_rank``: one tiny MAX all-reduce per optimizer step) before it is folded into
    the coefficient: a rank whose own loss was finite still skips the update its peers skip, so no rank ever applies the averaged
    NaN gradients or checkpoints past them;
  * with ``accum_iter > 1`` the DP all-reduce still overlaps the backward: the reducer is enabled on the LAST micro-step of a
    cycle, when each layer's bucket holds the cycle's accumulated sum as soon as that layer's backward is done.
"""
from __future__ import annotations

import inspect
import os
import math
import sys
from typing import Callable, Dict, Optional

import torch

from .dp import GradReducer, GradSquareSums, clip_grad_norm
from .util import adjust_learning_rate_epoch


def _unpack(batch):
    """(examples, labels, images, depth images) of a 3 / 4 / 5-tuple (engine_finetune.py:30-39)."""
    if len(batch) == 5:
        examples, labels, _mask, imgs, depth = batch
        return examples, labels, imgs, depth
    if len(batch) == 4:
        examples, labels, _mask, imgs = batch
        return examples, labels, imgs, None
    examples, labels, _mask = batch
    return examples, labels, None, None


def train_one_epoch(model, data_loader, optimizer, epoch: int, start_iter: int, args, reducer: Optional[GradReducer] = None,
                    log: Callable[[str], None] = print, on_save: Optional[Callable[[int], None]] = None) -> Dict[str, float]:
    model.train(True)
    accum_iter = args.accum_iter
    print_freq = getattr(args, "print_freq", 10)
    model.zero_grad(set_to_none=True)
    dev = next(model.parameters()).device
    n_iter = len(data_loader)                      # the FULL epoch, also when resuming at start_iter (sampler contract)
    stats = {"closs": 0.0, "n": 0, "grad_norm": 0.0, "lr": 0.0}
    params = [p for p in model.parameters() if p.requires_grad]
    clip = getattr(args, "clip_grad", -1) or -1
    bad = torch.zeros((), dtype=torch.bool, device=dev)          # sticky: some loss / gradient norm of this epoch was not finite
    zero1 = bool(getattr(optimizer, "zero1", False))            # zero1.Zero1Optimizer: also the gradient exchange (pass it as `reducer`)
    takes_scale = hasattr(optimizer, "engine") or zero1          # FusedAdamW / Zero1Optimizer: device-scalar coefficient, negative = skip

    def stop_if_bad(step: int) -> None:
        if bool(bad):                              # host read: only at logging / checkpoint / epoch boundaries
            log(f"Loss or gradient norm became non-finite at or before step {step}, stopping training")
            sys.exit(1)

    # our MetaModel: trim the batch on the host tensors (no device read) and tell the HIP engine the loss scale up front
    trim = getattr(type(model), "_trim", None) if "trimmed" in inspect.signature(model.forward).parameters else None
    engine = model.train_engine() if hasattr(model, "train_engine") and dev.type == "cuda" and params else None
    if engine is not None:
        engine.static_grad_scale = 1.0 / accum_iter
    # the clip's sums of squares ride the backward (side stream / behind each bucket's all-reduce) instead of one 27-GB pass after it
    sumsq = None
    if engine is not None and clip > 0 and takes_scale and not zero1 and optimizer.engine is engine:
        sumsq = getattr(engine, "_grad_square_sums", None)
        if sumsq is None or sumsq.reducer is not reducer:       # bound to ONE reducer: another (or none) this epoch = a new object
            if sumsq is not None:
                sumsq.detach()
            sumsq = engine._grad_square_sums = GradSquareSums(engine, reducer)
        sumsq.attach()
    overlap = takes_scale and engine is not None and getattr(optimizer, "engine", None) is engine \
        and os.environ.get("A3V_ADAMW_OVERLAP", "0") == "1"
    if zero1:
        # ZeRO-1: bucket i's all-gather under the AdamW of buckets i+1.. and the next forward's first layers (on by default; a no-op
        # at DP 1 and with stubbed collectives)
        overlap = engine is not None and getattr(optimizer, "eng", None) is engine and os.environ.get("A3V_ZERO1_OVERLAP", "1") == "1"
    try:
        for step, batch in enumerate(data_loader, start=start_iter):
            examples, labels, imgs, depth = _unpack(batch)
            if trim is not None and not examples.is_cuda:
                examples, labels = trim(examples, labels)
            if step % accum_iter == 0:
                stats["lr"] = adjust_learning_rate_epoch(optimizer, step / n_iter + epoch, lr=args.lr, min_lr=args.min_lr,
                                                         warmup_epochs=args.warmup_epochs, epochs=args.epochs)
            update_grad = (step + 1) % accum_iter == 0
            if reducer is not None:
                reducer.enabled = update_grad
            if sumsq is not None:
                sumsq.enabled = update_grad
            examples, labels = examples.to(dev, non_blocking=True), labels.to(dev, non_blocking=True)
            imgs = imgs.to(dev, non_blocking=True) if imgs is not None else None
            if trim is not None:
                c_loss, extra = model(examples, labels, images=imgs, depth_imgs=depth, trimmed=True)
            else:
                c_loss, extra = model(examples, labels, images=imgs, depth_imgs=depth)
            loss = c_loss
            for add_loss, weight in extra.values():
                loss = loss + add_loss * weight
            bad |= ~torch.isfinite(loss.detach())
            (loss / accum_iter).backward()
            if update_grad:
                if reducer is not None:
                    reducer.finish()
                    if hasattr(reducer, "any_rank"):
                        # DP: the flag must be the same on every rank BEFORE the update -- a NaN loss on one rank is in everybody's
                        # averaged gradients (one tiny MAX all-reduce on the stream, no host read)
                        bad = reducer.any_rank(bad)
                coef = None
                if clip > 0:
                    if zero1:
                        # the norm of the AVERAGED gradient from the ranks' slices (one tiny SUM all-reduce) + the replicated pieces
                        stats["grad_norm"], coef = optimizer.clip_coef(clip)
                    elif takes_scale:
                        # one norm over the engine's flat gradient buffer; the coefficient stays on the device and is applied by the
                        # optimizer kernel as it reads the gradients (a3v_adamw_scaled): no grad.mul_ pass, no host sync
                        eng = optimizer.engine
                        stats["grad_norm"], coef = clip_grad_norm(params, clip, flat=eng.flat_grads() if eng is not None else None, defer=True,
                                                                  sumsq=sumsq if eng is engine else None)
                    else:
                        stats["grad_norm"] = clip_grad_norm(params, clip)
                    bad |= ~torch.isfinite(torch.as_tensor(stats["grad_norm"]).to(dev))
                if takes_scale:
                    one = coef if coef is not None else torch.ones((), dtype=torch.float32, device=dev)
                    # overlap (A3V_ADAMW_OVERLAP=1, off by default): the update of layer i+1.. runs on the optimizer's stream under the
                    # next forward's layers ..i (optim.py)
                    optimizer.step(grad_scale=torch.where(bad, -one.new_ones(()), one).reshape(1), overlap=overlap)
                else:
                    stop_if_bad(step)                  # a stock optimizer cannot skip on a device flag: pay the host read
                    optimizer.step()
                model.zero_grad(set_to_none=True)
            boundary_idx = (step + 1) // accum_iter
            last = step + 1 == n_iter
            if update_grad and (boundary_idx % print_freq == 0 or last):
                stop_if_bad(step)
                lv = float(c_loss.detach())            # the loop's host sync: once per print_freq optimizer steps
                if not math.isfinite(lv):
                    log(f"Loss is {lv}, stopping training")
                    sys.exit(1)
                stats["closs"] += lv
                stats["n"] += 1
                gn = float(stats["grad_norm"]) if torch.is_tensor(stats["grad_norm"]) else stats["grad_norm"]
                log(f"Epoch: [{epoch}] [{step + 1}/{n_iter}] lr: {stats['lr']:.6f} closs: {lv:.4f} grad_norm: {gn:.4f}")
            if on_save is not None and update_grad and getattr(args, "save_iteration_interval", 0):
                if boundary_idx % max(args.save_iteration_interval // accum_iter, 1) == 0:
                    stop_if_bad(step)                  # never checkpoint past a bad step
                    if engine is not None:
                        engine.sync_optimizer()        # the checkpoint reads parameters and optimizer state on this stream
                    on_save(step)
    finally:
        # nothing of this epoch's wiring outlives it: the fused sums of squares and the static loss scale would otherwise leak into
        # bench / evaluation backward passes (or the next epoch's other reducer)
        if sumsq is not None:
            sumsq.detach()
        if engine is not None:
            engine.static_grad_scale = None
    if engine is not None:
        engine.sync_optimizer()                    # whatever runs next (checkpoint, evaluation) sees the last update
    stop_if_bad(n_iter - 1)                        # the caller writes the epoch-end checkpoint next
    return {"closs": stats["closs"] / max(stats["n"], 1), "lr": stats["lr"]}
