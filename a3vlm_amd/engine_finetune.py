"""``train_one_epoch`` with the semantics of the reference's engine (model/accessory/engine_finetune.py:13-105)
on top of the HIP training engine and the DP reducer.

Kept: 3/4/5-tuple batches, LR set per accumulation boundary by FRACTIONAL epoch, ``loss / accum_iter``,
non-finite loss -> ``sys.exit(1)``, clip (global L2, reference coefficient) then ``optimizer.step()`` only on
boundary micro-steps, ``zero_grad(set_to_none=True)``, gradient all-reduce skipped on non-boundary micro-steps
(the reference's ``no_sync``, util/misc.py:311-313).
Changed: no per-iteration ``cuda.synchronize()`` and no per-meter all-reduce with host sync every iteration
(engine_finetune.py:79,87-91) -- the loss is read back once per ``print_freq`` boundary steps.
"""
from __future__ import annotations

import math
import sys
from typing import Callable, Dict, Optional

import torch

from .dp import GradReducer, clip_grad_norm
from .util import adjust_learning_rate_epoch


def train_one_epoch(model, data_loader, optimizer, epoch: int, start_iter: int, args, reducer: Optional[GradReducer] = None,
                    log: Callable[[str], None] = print, on_save: Optional[Callable[[int], None]] = None) -> Dict[str, float]:
    model.train(True)
    accum_iter = args.accum_iter
    print_freq = getattr(args, "print_freq", 10)
    model.zero_grad(set_to_none=True)
    dev = next(model.parameters()).device
    n_iter = len(data_loader)
    stats = {"closs": 0.0, "n": 0, "grad_norm": 0.0, "lr": 0.0}
    params = [p for p in model.parameters() if p.requires_grad]
    for step, batch in enumerate(data_loader, start=start_iter):
        if len(batch) == 5:
            examples, labels, _mask, imgs, depth = batch
        elif len(batch) == 4:
            examples, labels, _mask, imgs = batch
            depth = None
        else:
            examples, labels, _mask = batch
            imgs = depth = None
        if step % accum_iter == 0:
            stats["lr"] = adjust_learning_rate_epoch(optimizer, step / n_iter + epoch, lr=args.lr, min_lr=args.min_lr,
                                                     warmup_epochs=args.warmup_epochs, epochs=args.epochs)
        update_grad = (step + 1) % accum_iter == 0
        if reducer is not None:
            reducer.enabled = update_grad and accum_iter == 1      # with accumulation the whole buffer is reduced at the boundary
        examples, labels = examples.to(dev, non_blocking=True), labels.to(dev, non_blocking=True)
        imgs = imgs.to(dev, non_blocking=True) if imgs is not None else None
        c_loss, extra = model(examples, labels, images=imgs, depth_imgs=depth)
        loss = c_loss
        for add_loss, weight in extra.values():
            loss = loss + add_loss * weight
        (loss / accum_iter).backward()
        if update_grad:
            if reducer is not None:
                if accum_iter > 1:
                    reducer.reduce_all_now()
                else:
                    reducer.finish()
            eng = getattr(optimizer, "engine", None)
            if getattr(args, "clip_grad", -1) and args.clip_grad > 0:
                if eng is not None:
                    # one norm over the engine's flat gradient buffer; the coefficient stays on the device and is applied by the
                    # optimizer kernel as it reads the gradients (a3v_adamw_scaled): no grad.mul_ pass, no host sync
                    stats["grad_norm"], coef = clip_grad_norm(params, args.clip_grad, flat=eng.flat_grads(), defer=True)
                    optimizer.step(grad_scale=coef)
                else:
                    stats["grad_norm"] = clip_grad_norm(params, args.clip_grad)
                    optimizer.step()
            else:
                optimizer.step()
            model.zero_grad(set_to_none=True)
        boundary_idx = (step + 1) // accum_iter
        if update_grad and (boundary_idx % print_freq == 0 or step + 1 == n_iter + start_iter):
            lv = float(c_loss.detach())            # the only host sync of the loop
            if not math.isfinite(lv):
                log(f"Loss is {lv}, stopping training")
                sys.exit(1)
            stats["closs"] += lv
            stats["n"] += 1
            gn = float(stats["grad_norm"]) if torch.is_tensor(stats["grad_norm"]) else stats["grad_norm"]
            log(f"Epoch: [{epoch}] [{step + 1}/{n_iter}] lr: {stats['lr']:.6f} closs: {lv:.4f} grad_norm: {gn:.4f}")
        if on_save is not None and update_grad and getattr(args, "save_iteration_interval", 0):
            if boundary_idx % max(args.save_iteration_interval // accum_iter, 1) == 0:
                on_save(step)
    return {"closs": stats["closs"] / max(stats["n"], 1), "lr": stats["lr"]}
