"""Import the reference (``/root/reference/model/accessory``) on CPU at MP=1.

TEST INFRASTRUCTURE ONLY -- used by ``oracle/gen_golden.py`` and by the
in-container oracle-vs-reference tests; never by the product, never on the GPU
box (``/root/reference`` does not exist there: ``available()`` is False).

The reference snapshot cannot be imported as shipped: it
needs ``fairscale``, ``open_clip``, ``dacite`` (absent), its own missing
``accessory/model/LLM/llama.py`` and ``accessory/configs/global_configs.py``,
and hard-codes ``.cuda()``.  This module injects *stand-ins written here* into
``sys.modules`` -- MP=1 forms of the fairscale layers (plain nn.Linear /
nn.Embedding honouring ``init_method``), an empty ``open_clip``, the restated
RoPE helpers from ``oracle/ref_cpu.py`` as ``accessory.model.LLM.llama`` -- and
makes ``Tensor.cuda`` the identity.  No reference source is copied.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = "/root/reference/model"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "accessory", "model"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _ColumnParallelLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, gather_output=True,
                 init_method=None, **kw):
        super().__init__(in_features, out_features, bias=bias)
        if init_method is not None:
            init_method(self.weight)


class _RowParallelLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, input_is_parallel=False,
                 init_method=None, **kw):
        super().__init__(in_features, out_features, bias=bias)
        if init_method is not None:
            init_method(self.weight)


class _ParallelEmbedding(nn.Embedding):
    def __init__(self, num_embeddings, embedding_dim, init_method=None, **kw):
        super().__init__(num_embeddings, embedding_dim)
        if init_method is not None:
            init_method(self.weight)


_installed = False


def _install_data_stubs() -> None:
    for name in ("h5py", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "torch.utils.tensorboard"):
        if name not in sys.modules:
            _mod(name)
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object


def install(data_stubs: bool = False) -> None:
    """Idempotently install the stand-ins and put the reference on sys.path.

    ``data_stubs`` additionally stubs h5py / torchvision / tensorboard so that the
    host-side modules (datasets, sampler, trainer) import; it must come after
    ``transformers`` has resolved its own optional torchvision probe."""
    global _installed
    if _installed:
        if data_stubs:
            _install_data_stubs()
        return
    assert available(), "reference tree not present"
    from oracle import ref_cpu
    from transformers import Blip2Processor, Blip2Model, Blip2Config  # noqa: F401  (llama_ens5.py:20)

    fs = _mod("fairscale")
    fs_nn = _mod("fairscale.nn")
    fs_mp = _mod("fairscale.nn.model_parallel")
    init = _mod(
        "fairscale.nn.model_parallel.initialize",
        get_model_parallel_world_size=lambda: 1,
        get_model_parallel_rank=lambda: 0,
        get_model_parallel_group=lambda: None,
        get_data_parallel_world_size=lambda: 1,
        get_data_parallel_rank=lambda: 0,
        get_data_parallel_group=lambda: None,
        get_model_parallel_src_rank=lambda: 0,
        model_parallel_is_initialized=lambda: True,
        initialize_model_parallel=lambda *a, **k: None,
        _MODEL_PARALLEL_GROUP=None,
    )
    layers = _mod(
        "fairscale.nn.model_parallel.layers",
        ColumnParallelLinear=_ColumnParallelLinear,
        RowParallelLinear=_RowParallelLinear,
        ParallelEmbedding=_ParallelEmbedding,
        _initialize_affine_weight=lambda *a, **k: None,
    )
    ident = lambda x: x  # noqa: E731
    mappings = _mod(
        "fairscale.nn.model_parallel.mappings",
        copy_to_model_parallel_region=ident, gather_from_model_parallel_region=ident,
        reduce_from_model_parallel_region=ident, scatter_to_model_parallel_region=ident,
    )
    utils = _mod("fairscale.nn.model_parallel.utils",
                 divide_and_check_no_remainder=lambda a, b: a // b, VocabUtility=object)
    fs.nn = fs_nn
    fs_nn.model_parallel = fs_mp
    fs_mp.initialize, fs_mp.layers, fs_mp.mappings, fs_mp.utils = init, layers, mappings, utils

    _mod("dacite", from_dict=lambda cls, d: cls(**d))
    _mod("open_clip")
    import re as _re
    sys.modules.setdefault("regex", _re)
    if data_stubs:
        _install_data_stubs()

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import accessory  # namespace package (the reference ships no __init__.py)
    _mod("accessory.configs.global_configs", USE_FLASH_ATTENTION=False)
    import accessory.configs  # noqa: F401
    accessory.configs.global_configs = sys.modules["accessory.configs.global_configs"]
    _mod("accessory.model.LLM.llama",
         precompute_freqs_cis=ref_cpu.precompute_freqs_cis,
         apply_rotary_emb=ref_cpu.apply_rotary_emb,
         repeat_kv=ref_cpu.repeat_kv,
         reshape_for_broadcast=ref_cpu.reshape_for_broadcast)

    torch.Tensor.cuda = lambda self, *a, **k: self
    _installed = True


def init_dist_ws1() -> None:
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29611")
        dist.init_process_group("gloo", rank=0, world_size=1)
