"""ctypes loader for liba3vlm_hip.so (the C-ABI declared in include/a3vlm_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent the first
use raises, so a GPU run can never silently route around the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# A3V_LIB_PATH: another build of the SAME library (tools/: cycle-stamp and ablation builds); never a different implementation
LIB_PATH = os.environ.get("A3V_LIB_PATH") or os.path.join(_HERE, "liba3vlm_hip.so")

BF16, F32 = 0, 1
EPI_NONE, EPI_BIAS, EPI_GELU, EPI_QUICKGELU, EPI_RESIDUAL, EPI_SWIGLU, EPI_OUT_F32, EPI_RES_F32 = 0, 1, 2, 4, 8, 16, 32, 64
EPI_SWIGLU_BWD = 128

EPI_TILE_128, EPI_TILE_256, EPI_TILE_256PP, EPI_TILE_256PP32 = 1 << 16, 1 << 17, 1 << 18, 1 << 19
EPI_TILE_192PP = 1 << 23

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> (restype, argtypes); mirrors include/a3vlm_hip.h one to one
SIGNATURES = {
    "a3v_version": (I, []),
    "a3v_reload_env": (I, []),
    "a3v_build_flags": (I, []),
    "a3v_gemm_nt": (I, [P, L, P, L, P, L, I, I, I, P, P, L, I, I, P]),
    "a3v_gemm_qkv_rope": (I, [P, L, P, L, I, P, L, P, P, P, L, P, L, P, I, I, I, I, I, I, I, I, P]),
    "a3v_gemm_nt_fp8": (I, [P, L, P, P, L, P, P, L, I, I, I, P, P, L, I, P]),
    "a3v_gemm_qkv_rope_fp8": (I, [P, L, P, P, L, P, I, P, L, P, P, P, I, I, I, I, I, I, I, I, P]),
    "a3v_quantize_rows_fp8": (I, [P, L, P, F, P, L, P, I, I, I, P]),
    "a3v_gemm_tn_splitk": (I, [P, L, P, L, P, I, I, I, I, P]),
    "a3v_attention_bwd_packed": (I, [P, P, L, L, P, L, L, L, P, L, P, P, P, P, L, P, I, I, I, I, I, I, I, I, P]),
    "a3v_lora_refresh": (I, [P, P, I, I, I, P, L, P, L, P, L, P, L, I, I, P]),
    "a3v_adamw": (I, [P, P, P, P, L, F, F, F, F, F, L, P, P]),
    "a3v_adamw_scaled": (I, [P, P, P, P, L, F, F, F, F, F, L, P, P, P]),
    "a3v_adamw_scaled_t": (I, [P, P, P, P, I, I, F, F, F, F, F, L, P, P, L, P, P]),
    "a3v_gemm_tn_strip": (I, [P, L, P, L, P, I, I, I, I, P]),
    "a3v_lora_gb_scatter": (I, [P, L, I, I, P, P, P, P]),
    "a3v_adamw_multi": (I, [P, I, L, F, F, F, F, F, L, P, P]),
    "a3v_scale_cast": (I, [P, I, P, I, L, F, P]),
    "a3v_sumsq_partials": (I, [P, L, P, P]),
    "a3v_grad_bucket_allreduce": (I, [P, P, L, P, I, P]),
    "a3v_grad_bucket_reduce_scatter": (I, [P, P, L, P, P, P, I, P]),
    "a3v_param_shard_all_gather": (I, [P, P, P, L, P]),
    "a3v_rccl_comm_count": (I, [P]),
    "a3v_rccl_available": (I, []),
    "a3v_gemm_nn": (I, [P, L, P, L, P, L, I, I, I, P, L, I, P]),
    "a3v_gemm_tn": (I, [P, L, P, L, P, L, I, I, I, P, L, I, P]),
    "a3v_gemm_tn_sumsq": (I, [P, L, P, L, P, L, I, I, I, P, L, I, P, L, P]),
    "a3v_gemm_tn_sumsq_slots": (L, [I, I]),
    "a3v_gemm_set_workspace": (I, [P, L]),
    "a3v_gemm_set_workspace_for": (I, [P, P, L]),
    "a3v_gemm_nt_splitk": (I, [P, L, P, L, P, I, I, I, I, P]),
    "a3v_splitk_reduce": (I, [P, I, I, I, P, L, I, I, P]),
    "a3v_gemm_skinny_split": (I, [I, I, I]),
    "a3v_gemm_skinny_ws_bytes": (L, [I, I, I]),
    "a3v_gemm_skinny_fp8": (I, [P, L, P, L, P, P, L, I, I, I, P, L, I, P, P]),
    "a3v_gemm_skinny": (I, [P, L, P, L, P, L, I, I, I, P, L, I, P, P]),
    "a3v_rmsnorm": (I, [P, L, P, P, L, I, I, F, I, I, I, P]),
    "a3v_layernorm": (I, [P, L, P, P, P, L, P, I, I, F, I, I, I, P]),
    "a3v_rope_kvcache": (I, [P, L, P, L, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "a3v_vt_pack": (I, [P, L, P, I, I, I, I, I, I, P]),
    "a3v_attention_scratch_floats": (L, [I, I, I, I]),
    "a3v_attention": (I, [P, P, P, P, I, I, I, I, I, I, ctypes.POINTER(c_int64), I, P, I, P]),
    "a3v_embed_assemble": (I, [P, L, P, P, I, I, I, I, I, I, I, P]),
    "a3v_fill_rows": (I, [P, P, L, P, I, I, I, I, P]),
    "a3v_patch_im2col": (I, [P, P, I, I, I, I, I, I, I, P]),
    "a3v_split_views": (I, [P, P, I, I, I, I, P]),
    "a3v_vit_embed": (I, [P, P, P, P, I, I, I, I, P]),
    "a3v_argmax": (I, [P, L, P, I, I, P]),
    "a3v_sample_top_p": (I, [P, L, I, I, F, F, P, P, P]),
    "a3v_preprocess_image": (I, [P, I, I, I, I, I, P, P, P, I, P, P, I, I, P, P, I, P, P, P]),
    "a3v_preprocess_batch": (I, [P, I, P, I, P, L, P, L, I, P, P, P]),
    "a3v_generate_step": (I, [P, L, P, I, I, P, L, P, L, I, P, P, I, P, P, P, P]),
    "a3v_count_valid": (I, [P, I, P, P]),
    "a3v_cross_entropy": (I, [P, L, P, P, P, L, P, F, I, I, I, P]),
    "a3v_attention_lse": (I, [P, P, P, P, P, I, I, I, I, I, I, ctypes.POINTER(c_int64), I, I, P]),
    "a3v_transpose": (I, [P, L, L, P, L, L, I, I, I, I, I, P]),
    "a3v_rmsnorm_bwd_scratch_floats": (L, [I, I]),
    "a3v_rmsnorm_bwd": (I, [P, L, P, P, L, P, L, P, P, I, I, F, I, P]),
    "a3v_rmsnorm_bwd_cast": (I, [P, L, P, P, L, P, L, P, P, I, I, F, I, P, L, P]),
    "a3v_rmsnorm_bwd_bf16": (I, [P, L, P, P, L, P, L, P, P, I, I, F, P]),
    "a3v_layernorm_bwd_bf16": (I, [P, L, P, P, L, P, P, L, P, P, I, I, F, P]),
    "a3v_embed_bwd_bf16": (I, [P, L, P, P, I, I, I, I, I, P]),
    "a3v_layernorm_bwd": (I, [P, L, P, P, L, P, P, L, P, P, I, I, F, I, P]),
    "a3v_swiglu_fwd": (I, [P, L, P, L, I, I, I, I, P]),
    "a3v_swiglu_bwd": (I, [P, L, P, L, P, L, I, I, I, I, P]),
    "a3v_cast": (I, [P, L, I, P, L, I, I, I, P]),
    "a3v_add2d": (I, [P, L, P, L, I, I, I, P]),
    "a3v_rope_bwd_pack": (I, [P, P, P, P, L, P, I, I, I, I, I, I, I, P]),
    "a3v_attention_bwd": (I, [P, P, L, L, P, L, L, L, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "a3v_attention_bwd_workspace_bytes": (L, [I, I, I, I, I]),
    "a3v_attention_bwd_mfma": (I, [P, P, L, L, P, L, L, L, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "a3v_embed_bwd": (I, [P, L, P, P, I, I, I, I, I, P]),
    "a3v_rows_sum": (I, [P, L, P, I, I, P, I, P]),
}

class LlamaLayer(ctypes.Structure):
    """a3v_llama_layer of include/a3vlm_hip.h"""
    _fields_ = [(n, c_void_p) for n in ("attn_norm_w", "wqkv", "wo", "ffn_norm_w", "w13", "w2", "k_cache", "vt_cache",
                                        "wqkv_q", "wqkv_s", "wo_q", "wo_s", "w13_q", "w13_s", "w2_q", "w2_s")]


class ImageDesc(ctypes.Structure):
    """a3v_image_desc of include/a3vlm_hip.h"""
    _fields_ = [("src", c_void_p), ("coeffs", c_void_p), ("bounds", c_void_p), ("H", c_int), ("W", c_int), ("side", c_int), ("pad_x", c_int),
                ("pad_y", c_int), ("ksize", c_int)]


SIGNATURES["a3v_llama_decode_step"] = (I, [ctypes.POINTER(LlamaLayer), I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, P])
SIGNATURES["a3v_llama_decode_step_form"] = (I, [I, I, I, I, I, I, I])

_lib = None


def load() -> ctypes.CDLL:
    """Load the library once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C a3vlm_amd/csrc`). "
            "a3vlm_amd has no non-HIP fallback.")
    # PyTorch-ROCm ships its own libamdhip64; it must be the HIP runtime this library binds to (device memory and
    # streams come from torch).  dlopen-ing liba3vlm_hip.so BEFORE torch would pull in /opt/rocm's copy as a second
    # runtime in the process, and every launch from here would then fail with hipErrorNoDevice.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def has_experiments() -> bool:
    """True when the library was built with `make EXPERIMENTS=1` (the measured-and-not-dispatched GEMM kernels exist)."""
    return bool(load().a3v_build_flags() & 1)


class env:
    """``with lib.env(A3V_GEMM_FAST_EPI="0"): ...`` -- set A3V_* switches for a block and have the library re-read them (the
    library caches every switch after its first read), restoring both on exit.  A/B runs and equality tests only."""

    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}
        self.old = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v
        load().a3v_reload_env()
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        load().a3v_reload_env()
        return False


class A3VError(RuntimeError):
    pass


_ERR = {-1: "A3V_ERR_SHAPE (unsupported size/alignment)", -2: "A3V_ERR_DTYPE", -3: "A3V_ERR_ARG"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise A3VError(f"{what} failed: {_ERR.get(rc, f'hipError_t {rc}')}")
