"""Host-side helpers mirroring the reference's util/ pieces the path needs."""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def promote_trainable_params_to_fp32(model: nn.Module, keep_matrices_sharded: bool = False) -> None:
    """util/tensor_type.py:60-66: trainables become fp32 masters, frozen params keep their dtype.
    ``keep_matrices_sharded`` (ZeRO-1, ``--zero1``): the decoder's linears, the embeddings and the LM head stay in the compute dtype --
    their fp32 masters exist only as 1/N slices inside ``zero1.Zero1Optimizer``, as FSDP(SHARD_GRAD_OP) shards them in the reference
    (main_finetune.py:241-263); everything else (norms, projector, tags, adapters) is promoted as usual."""
    big = (".wq.weight", ".wk.weight", ".wv.weight", ".wo.weight", ".w1.weight", ".w2.weight", ".w3.weight")
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if keep_matrices_sharded and "lora_" not in n and (n.endswith(big) and ".layers." in "." + n or n.endswith(("tok_embeddings.weight", "output.weight"))) \
                and "clip." not in n and "visual" not in n:
            continue
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


# ---------------------------------------------------------------------------------------------------------------------
# Parameter-update epochs.  The compute-dtype weight images (TrainEngine._Images, the LoRA step images, the plugin's packed
# weights) are caches keyed on the parameters' state.  ``Tensor._version`` alone is NOT a safe key: torch.optim.AdamW(fused=True)
# -- what the reference trainer's optimizer amounts to on recent torch -- updates parameters through a fused kernel that leaves
# ``_version`` untouched (checked on torch 2.10 / ROCm), so a version-keyed cache would keep serving the pre-update weights.
# A global optimizer post-step hook therefore counts, per parameter object, the optimizer steps that saw a gradient for it.
_PARAM_EPOCH: "dict[int, int]" = {}
_HOOKED = False


_OPT_STEPS = [0]


def optimizer_steps() -> int:
    """How many optimizer steps (any torch.optim.Optimizer subclass) this process has taken: a cheap "nothing can have changed" test
    in front of the per-parameter cache keys."""
    return _OPT_STEPS[0]


def _after_optimizer_step(optimizer, args, kwargs) -> None:
    _OPT_STEPS[0] += 1
    if getattr(optimizer, "bumps_version", False):      # e.g. a3vlm_amd.optim.FusedAdamW: Tensor._version already moved
        return
    for group in optimizer.param_groups:
        for p in group["params"]:
            if p.grad is not None:
                _PARAM_EPOCH[id(p)] = _PARAM_EPOCH.get(id(p), 0) + 1


def install_param_epoch_hook() -> None:
    """Idempotent: registers the global torch.optim post-step hook (any optimizer class, fused or not)."""
    global _HOOKED
    if not _HOOKED:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(_after_optimizer_step)
        _HOOKED = True


def param_state_key(p) -> tuple:
    """Cache key of a parameter's current value: (autograd version, optimizer-step epoch)."""
    return (p._version, _PARAM_EPOCH.get(id(p), 0))

