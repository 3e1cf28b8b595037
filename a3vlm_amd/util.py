"""Host-side helpers mirroring the reference's util/ pieces the path needs."""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def promote_trainable_params_to_fp32(model: nn.Module) -> None:
    """util/tensor_type.py:60-66: trainables become fp32 masters, frozen params keep their dtype."""
    for p in model.parameters():
        if p.requires_grad:
            p.data = p.data.float()


def adjust_learning_rate_epoch(optimizer, epoch: float, *, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """util/lr_sched.py:23-35: linear warm-up then half-cosine by FRACTIONAL epoch; honours per-group lr_scale."""
    if epoch < warmup_epochs:
        new_lr = lr * epoch / warmup_epochs
    else:
        new_lr = min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))
    for g in optimizer.param_groups:
        g["lr"] = new_lr * g["lr_scale"] if "lr_scale" in g else new_lr
    return new_lr


def add_weight_decay(model: nn.Module, weight_decay: float = 1e-5, skip_list=()):
    """util/misc.py:586-599: no decay for *.bias and *norm.weight."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if name.endswith(".bias") or name.endswith("norm.weight") or name in skip_list:
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]
