"""Checkpoint layout of the reference (drop-in contract).

Writer  : ``save_checkpoint``  -- util/misc.py:324-438 at MP = 1: ``epoch{E}[-iter{I}]/consolidated.00-of-01.model.pth``
          = ``{"model": {"llma.<key>": tensor in the save dtype}}``, ``tokenizer.model``, ``config.json``
          (= dataclasses.asdict(model.llma.args)), ``meta.json`` ({"llama_type": ...}), ``...optimizer.pth``,
          ``...other.pth`` and ``rank-specific-{rank:05d}-of-{ws:05d}.pth``.
Reader  : ``load_tensor_parallel_model_list`` -- util/tensor_parallel.py:425-485 incl. the add-to-existing semantics of
          ``consolidated_diff`` folders (:387-423) and the MP split (:133-161, ``split_tensor_parallel_state_dict``): format / MP-size inference from the
          file names (:40-45, :333-384), ``meta_ori`` key prefixing (:223-225), merge of MP > 1 shards along the
          tensor-parallel dim of each weight (Column -> 0, Row -> 1, Embedding -> 1, :34-38; key table
          tools/convert_weights_to_hf.py:101-115), replicated params taken from shard 0 with a consistency warning
          (:116-123), ``load_state_dict(strict=False)`` and the {'missing_keys','unexpected_keys'} summary.
"""
from __future__ import annotations

import dataclasses
import json
import os
import re
import warnings
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

FORMAT_FILENAME_PATTERNS = {
    "meta_ori": re.compile(r"^consolidated.(\d{2}).pth$"),
    "consolidated": re.compile(r"^consolidated.(\d{2})-of-(\d{2}).model.pth$"),
    "consolidated_diff": re.compile(r"^consolidated.(\d{2})-of-(\d{2}).model-diff.pth$"),
}

# tensor-parallel merge dim by key (everything else is replicated)
_MERGE_DIM = (
    (re.compile(r"^llma\.tok_embeddings\.weight$"), 1),
    (re.compile(r"^llma\.layers\.\d+\.attention\.w[qkv]\.weight$"), 0),
    (re.compile(r"^llma\.layers\.\d+\.attention\.wo\.weight$"), 1),
    (re.compile(r"^llma\.layers\.\d+\.feed_forward\.w[13]\.weight$"), 0),
    (re.compile(r"^llma\.layers\.\d+\.feed_forward\.w2\.weight$"), 1),
    (re.compile(r"^llma\.output\.weight$"), 0),
)


def merge_dim(key: str) -> int:
    for pat, d in _MERGE_DIM:
        if pat.match(key):
            return d
    return -1


def shard_file_names(fmt: str, num_shards: int) -> List[str]:
    if fmt == "meta_ori":
        return [f"consolidated.{i:02d}.pth" for i in range(num_shards)]
    suffix = "model.pth" if fmt == "consolidated" else "model-diff.pth"
    return [f"consolidated.{i:02d}-of-{num_shards:02d}.{suffix}" for i in range(num_shards)]


def infer_checkpoint_format_and_mp_size(path: str) -> Tuple[str, int]:
    if not os.path.isdir(path):
        raise NotImplementedError("The given path does not point to a valid folder.")
    files = [f for f in os.listdir(path) if os.path.isfile(os.path.join(path, f))]
    fmt, mp = None, None
    for name, pat in FORMAT_FILENAME_PATTERNS.items():
        matched = [f for f in files if pat.match(f)]
        if matched:
            if fmt is not None:
                raise NotImplementedError(f"Multiple matched format detected: {fmt} and {name}.")
            fmt, mp = name, len(matched)
    if fmt is None:
        raise NotImplementedError(f"Files in the given folder do not match any format. Contents: {sorted(os.listdir(path))}")
    for fn in shard_file_names(fmt, mp):
        if fn not in files:
            raise NotImplementedError("An expected file is not found in the target folder: " + fn)
    return fmt, mp


def load_shard(path: str, fmt: str, shard_id: int, num_shards: int) -> Dict[str, torch.Tensor]:
    shard = torch.load(os.path.join(path, shard_file_names(fmt, num_shards)[shard_id]), map_location="cpu", weights_only=False)
    if fmt.startswith("consolidated"):
        if "model" in shard and isinstance(shard["model"], dict):
            shard = shard["model"]
    elif fmt == "meta_ori":
        shard = {"llma." + k: v for k, v in shard.items()}
    return shard


def load_merged_state_dict(path: str, known_keys=None) -> "OrderedDict[str, torch.Tensor]":
    """All MP shards of one checkpoint folder merged to MP = 1."""
    fmt, mp = infer_checkpoint_format_and_mp_size(path)
    shards = [load_shard(path, fmt, i, mp) for i in range(mp)]
    merged: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key in sorted({k for s in shards for k in s}):
        if known_keys is not None and key not in known_keys:
            continue
        parts = [s[key] for s in shards if key in s]
        d = merge_dim(key)
        if d >= 0 and len(parts) > 1:
            merged[key] = torch.cat(parts, dim=d)
        else:
            if len(parts) > 1:
                diff = max(float((p.float() - parts[0].float()).abs().max()) for p in parts[1:])
                if diff > 0.0:
                    warnings.warn(f"Found unequal replicas of non-tensor-parallel params: name={key}, max_diff={diff}.")
            merged[key] = parts[0]
    return merged


def split_tensor_parallel_state_dict(state_dict: Dict[str, torch.Tensor], mp_size: int) -> List[Dict[str, torch.Tensor]]:
    """The inverse of the shard merge (util/tensor_parallel.py:133-161): MP = 1 tensors cut into ``mp_size`` shards along each
    weight's tensor-parallel dim, replicated parameters copied to every shard -- what a reference MP = ``mp_size`` job loads."""
    shards: List[Dict[str, torch.Tensor]] = [OrderedDict() for _ in range(mp_size)]
    for key, value in state_dict.items():
        d = merge_dim(key)
        if d < 0 or mp_size == 1:
            for sh in shards:
                sh[key] = value
            continue
        if value.shape[d] % mp_size:
            raise ValueError(f"{key}: dim {d} of size {value.shape[d]} does not split into {mp_size} shards")
        for sh, piece in zip(shards, torch.chunk(value, mp_size, dim=d)):
            sh[key] = piece.contiguous()
    return shards


def _load_filtered(model, sd: Dict[str, torch.Tensor], own: Dict[str, torch.Tensor], mismatched: List[str]):
    """load_state_dict(strict=False) that REPORTS shape-mismatched keys instead of raising from inside torch (a checkpoint of the
    full 4-encoder ensemble has e.g. visual_proj.0.weight [dim, 5632] against a CLIP-only build's [dim, 1024])."""
    ok = {}
    for k, v in sd.items():
        if k not in own:
            continue
        if tuple(v.shape) != tuple(own[k].shape):
            mismatched.append(f"{k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
            continue
        ok[k] = v
    return model.load_state_dict(ok, strict=False)


def load_tensor_parallel_model_list(model, paths: List[str], verbose: bool = False) -> Dict[str, List[str]]:
    """Loads one or more checkpoint folders sequentially into a MetaModel-like module whose parameters carry the ``llma.``
    prefix (util/tensor_parallel.py:425-485).  A base folder (``meta_ori`` / ``consolidated``) overrides what earlier folders
    set; a ``consolidated_diff`` folder ADDS its tensors to the values already loaded from earlier folders and plainly sets keys
    no earlier folder provided (load_diff_checkpoint, :387-423 -- implemented as documented there; the reference's own call site
    :470-472 omits the ``existing_keys`` argument and cannot run).  Returns the keys still missing after the last folder and the
    union of unexpected keys; shape-mismatched keys are skipped and listed under ``mismatched_keys`` when there are any."""
    if isinstance(paths, str):
        paths = [paths]
    own = model.state_dict()
    existing = set(own.keys())
    seen: set = set()
    missing, unexpected, mismatched = set(existing), set(), []
    for i, path in enumerate(paths):
        fmt, _ = infer_checkpoint_format_and_mp_size(path)
        assert i != 0 or not fmt.endswith("_diff"), "The first checkpoint in the list cannot be a *_diff checkpoint."
        sd = load_merged_state_dict(path)
        if fmt.endswith("_diff"):
            cur = model.state_dict()
            for k in list(sd.keys()):
                if k in seen and k in cur and tuple(cur[k].shape) == tuple(sd[k].shape):
                    sd[k] = cur[k].detach().to("cpu") + sd[k].to(cur[k].dtype)
        res = _load_filtered(model, sd, own, mismatched)
        unexpected |= {k for k in sd if k not in existing}
        step_missing = set(res.missing_keys)
        missing = {k for k in missing if k in step_missing}
        seen |= set(sd.keys())
        if verbose:
            print(f"loaded {path} ({fmt}): {len(sd)} tensors")
    if hasattr(getattr(model, "llma", None), "invalidate_packed_weights"):
        model.llma.invalidate_packed_weights()
    out = {"missing_keys": sorted(missing), "unexpected_keys": sorted(unexpected)}
    if mismatched:
        warnings.warn("checkpoint tensors skipped because their shapes differ from the model's:\n  " + "\n  ".join(mismatched))
        out["mismatched_keys"] = mismatched
    return out


def save_checkpoint(output_dir: str, args, model, optimizer=None, loss_scaler=None, dataset_state=None, epoch=None,
                    iteration=None, rank: int = 0, world_size: int = 1) -> str:
    name = f"epoch{epoch}" + (f"-iter{iteration}" if iteration is not None else "")
    save_dir = os.path.join(output_dir, name)
    os.makedirs(save_dir, exist_ok=True)
    # ZeRO-1: the slices of the fp32 masters and moments are gathered bucket by bucket (COLLECTIVE: every rank calls this) and rank 0
    # writes ONE world-size-independent ``consolidated.00-of-01.optimizer.pth``, as the reference does with
    # ``FSDP.full_optim_state_dict`` on DP rank 0 (util/misc.py:395-403); any DP size -- and a resume at another one -- can load it
    zero1_full = optimizer.full_state_dict() if optimizer is not None and getattr(optimizer, "zero1", False) else None
    if rank == 0:
        save_dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "tf32": torch.float}[getattr(args, "precision", "bf16")]
        sd = model.state_dict()
        if getattr(args, "only_save_trainable", False):
            keep = set(model.get_trainable_params().keys())
            sd = {k: v for k, v in sd.items() if k in keep}
        torch.save({"model": {k: v.to(save_dtype) for k, v in sd.items()}},
                   os.path.join(save_dir, "consolidated.00-of-01.model.pth"))
        model.tokenizer.save(save_dir)
        with open(os.path.join(save_dir, "config.json"), "w") as f:
            json.dump(dataclasses.asdict(model.llma.args), f, indent=2)
        with open(os.path.join(save_dir, "meta.json"), "w") as f:
            json.dump({"llama_type": model.llama_type}, f, indent=2)
        if optimizer is not None:
            torch.save({"optimizer": zero1_full if zero1_full is not None else optimizer.state_dict()},
                       os.path.join(save_dir, "consolidated.00-of-01.optimizer.pth"))
        torch.save({"epoch": epoch, "iter": iteration,
                    "scaler": loss_scaler.state_dict() if hasattr(loss_scaler, "state_dict") else loss_scaler,
                    "args": vars(args) if hasattr(args, "__dict__") else args},
                   os.path.join(save_dir, "consolidated.00-of-01.other.pth"))
    torch.save({"dataset_state": dataset_state}, os.path.join(save_dir, f"rank-specific-{rank:05d}-of-{world_size:05d}.pth"))
    return save_dir


def latest_checkpoint_dir(ckpt_path: str):
    """util/misc.py:440-464: the ``epoch{E}[-iter{I}]`` sub-folder with the largest (epoch, iter)."""
    if not os.path.isdir(ckpt_path):
        return None
    best, best_key = None, None
    for name in os.listdir(ckpt_path):
        m = re.match(r"^epoch(\d+)(?:-iter(\d+))?$", name)
        if not m or not os.path.isdir(os.path.join(ckpt_path, name)):
            continue
        key = (int(m.group(1)), float("inf") if m.group(2) is None else int(m.group(2)))
        if best_key is None or key > best_key:
            best, best_key = os.path.join(ckpt_path, name), key
    return best
