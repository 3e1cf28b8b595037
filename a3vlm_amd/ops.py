"""Thin tensor wrappers over the C-ABI (include/a3vlm_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every op below is one
call into liba3vlm_hip.so with raw device pointers on torch's CURRENT stream.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import lib as _l
from .lib import BF16, F32, EPI_BIAS, EPI_GELU, EPI_QUICKGELU, EPI_RESIDUAL, EPI_SWIGLU, EPI_OUT_F32, EPI_RES_F32, EPI_SWIGLU_BWD  # noqa: F401

_DT = {torch.bfloat16: BF16, torch.float32: F32}


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"a3vlm_amd supports bf16/fp32 tensors, got {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("a3vlm_amd ops need device tensors (no CPU fallback exists)")


_gemm_ws = {}                 # (device, stream) -> scratch tensor, in order of last use (dicts keep insertion order)
GEMM_WS_MAX_STREAMS = 8       # per process: a stream that has not run a GEMM while eight others did loses its 96-MiB scratch (it gets a new one
                              # on its next GEMM); before round 5 every stream that ever ran a GEMM kept one for the life of the process


def _ensure_gemm_workspace(device) -> None:
    """Register (once per device and stream) the scratch the GEMM entry points may use for the split-K planes of their hybrid dispatch
    (a3v_gemm_set_workspace_for: keyed by stream, so concurrent streams never share planes).  The scratch is allocated while its stream is
    current and only ever used by launches on that stream, so dropping it (least recently used first, beyond GEMM_WS_MAX_STREAMS) is
    ordered like any other free on that stream by the caching allocator; the registration is removed first."""
    st = _stream()
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), st)
    ws = _gemm_ws.pop(key, None)
    if ws is not None:
        _gemm_ws[key] = ws            # most recently used last
        return
    while len(_gemm_ws) >= GEMM_WS_MAX_STREAMS:
        (odev, ost), old = next(iter(_gemm_ws.items()))
        with torch.cuda.device(odev):
            _l.check(_l.load().a3v_gemm_set_workspace_for(ost, None, 0), "a3v_gemm_set_workspace_for(unregister)")
        del _gemm_ws[(odev, ost)], old
    ws = torch.empty(96 << 20, dtype=torch.uint8, device=device)
    _gemm_ws[key] = ws
    with torch.cuda.device(key[0]):
        _l.check(_l.load().a3v_gemm_set_workspace_for(st, ws.data_ptr(), ws.numel()), "a3v_gemm_set_workspace_for")


def gemm_nt(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias=None, residual=None,
            epilogue: int = 0) -> torch.Tensor:
    """out[M, N or N/2] = epilogue(a[M,K] @ w[N,K]^T); a/w/out 2-D with unit inner stride."""
    _dev(a, w, out, bias, residual)
    _ensure_gemm_workspace(a.device)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    ep = epilogue
    if bias is not None:
        ep |= EPI_BIAS
    if residual is not None and not (ep & (EPI_RES_F32 | EPI_SWIGLU_BWD)):      # (EPI_SWIGLU_BWD: `residual` carries the forward's gate | up rows)
        ep |= EPI_RESIDUAL
    rc = _l.load().a3v_gemm_nt(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K,
                               _p(bias), _p(residual), residual.stride(0) if residual is not None else 0,
                               ep, dt(a), _stream())
    _l.check(rc, f"a3v_gemm_nt(M={M},N={N},K={K},epi={ep})")
    return out


def gemm_tn_sumsq_slots(M: int, N: int) -> int:
    return int(_l.load().a3v_gemm_tn_sumsq_slots(M, N))


def gemm_tn(at, wt, out, *, residual=None, epilogue: int = 0, sumsq=None):
    """out[M, N] = epilogue(at[K, M]^T @ wt[K, N]) -- both operands with the contracted index as rows (no transposes).
    ``sumsq`` (fp32 outputs only): a zeroed float32 buffer of >= gemm_tn_sumsq_slots(M, N) elements that receives partial sums of
    squares of the stored values (the clip's norm without a pass over the gradient)."""
    _dev(at, wt, out, residual, sumsq)
    K, M = at.shape
    N = wt.shape[1]
    assert wt.shape[0] == K and at.stride(1) == 1 and wt.stride(1) == 1 and out.stride(1) == 1
    _ensure_gemm_workspace(at.device)
    ep = epilogue
    if residual is not None and not (ep & EPI_RES_F32):
        ep |= EPI_RESIDUAL
    if sumsq is not None:
        assert sumsq.dtype == torch.float32 and sumsq.is_contiguous()
        rc = _l.load().a3v_gemm_tn_sumsq(_p(at), at.stride(0), _p(wt), wt.stride(0), _p(out), out.stride(0), M, N, K,
                                         _p(residual), residual.stride(0) if residual is not None else 0, ep, _p(sumsq), sumsq.numel(), _stream())
        _l.check(rc, f"a3v_gemm_tn_sumsq(M={M},N={N},K={K},epi={ep})")
        return out
    rc = _l.load().a3v_gemm_tn(_p(at), at.stride(0), _p(wt), wt.stride(0), _p(out), out.stride(0), M, N, K,
                               _p(residual), residual.stride(0) if residual is not None else 0, ep, _stream())
    _l.check(rc, f"a3v_gemm_tn(M={M},N={N},K={K},epi={ep})")
    return out


def gemm_nt_splitk(a, w, out, scratch, S: int, accumulate: bool = False):
    """out (+)= a @ w.T through S split-K planes (fp32 scratch of >= S*M*N floats); rounds once to out.dtype."""
    _dev(a, w, out, scratch)
    M, K = a.shape
    N = w.shape[0]
    assert scratch.dtype == torch.float32 and scratch.numel() >= S * M * N
    lib = _l.load()
    _l.check(lib.a3v_gemm_nt_splitk(_p(a), a.stride(0), _p(w), w.stride(0), _p(scratch), M, N, K, S, _stream()), f"a3v_gemm_nt_splitk(M={M},N={N},K={K},S={S})")
    _l.check(lib.a3v_splitk_reduce(_p(scratch), S, M, N, _p(out), out.stride(0), dt(out), 1 if accumulate else 0, _stream()), "a3v_splitk_reduce")
    return out


def gemm_nn(a, wt, out, *, residual=None, epilogue: int = 0):
    """out[M, N] = epilogue(a[M, K] @ wt[K, N]) -- wt row-indexed by the contracted index (dX = dY @ W on the forward image)."""
    _dev(a, wt, out, residual)
    M, K = a.shape
    N = wt.shape[1]
    assert wt.shape[0] == K and a.stride(1) == 1 and wt.stride(1) == 1 and out.stride(1) == 1
    _ensure_gemm_workspace(a.device)
    ep = epilogue
    if residual is not None and not (ep & (EPI_RES_F32 | EPI_SWIGLU_BWD)):
        ep |= EPI_RESIDUAL
    rc = _l.load().a3v_gemm_nn(_p(a), a.stride(0), _p(wt), wt.stride(0), _p(out), out.stride(0), M, N, K,
                               _p(residual), residual.stride(0) if residual is not None else 0, ep, _stream())
    _l.check(rc, f"a3v_gemm_nn(M={M},N={N},K={K},epi={ep})")
    return out


def gemm_tn_strip(t, x, out, scratch, S: int, accumulate: bool = False):
    """out[R, N] (+)= t[Kt, R]^T @ x[Kt, N] for R <= 64 (adapter weight gradients) through the small-block streaming kernel
    a3v_gemm_tn_strip + the split-K reduce; t's rows must hold 64 readable elements (the kernel loads 64 columns)."""
    _dev(t, x, out, scratch)
    Kt, R = t.shape
    N = x.shape[1]
    assert x.shape[0] == Kt and t.stride(1) == 1 and x.stride(1) == 1 and t.stride(0) >= 64
    assert scratch.dtype == torch.float32 and scratch.numel() >= S * R * N
    lib = _l.load()
    _l.check(lib.a3v_gemm_tn_strip(_p(t), t.stride(0), _p(x), x.stride(0), _p(scratch), R, N, Kt, S, _stream()), f"a3v_gemm_tn_strip(R={R},N={N},Kt={Kt},S={S})")
    _l.check(lib.a3v_splitk_reduce(_p(scratch), S, R, N, _p(out), out.stride(0), dt(out), 1 if accumulate else 0, _stream()), "a3v_splitk_reduce")
    return out


def gemm_tn_splitk(at, wt, out, scratch, S: int, accumulate: bool = False):
    """out[M, N] (+)= at[K, M]^T @ wt[K, N] through S split-K planes of the TN kernel (adapter-sized M or N, long K)."""
    _dev(at, wt, out, scratch)
    K, M = at.shape
    N = wt.shape[1]
    assert wt.shape[0] == K and at.stride(1) == 1 and wt.stride(1) == 1
    assert scratch.dtype == torch.float32 and scratch.numel() >= S * M * N
    lib = _l.load()
    _l.check(lib.a3v_gemm_tn_splitk(_p(at), at.stride(0), _p(wt), wt.stride(0), _p(scratch), M, N, K, S, _stream()), f"a3v_gemm_tn_splitk(M={M},N={N},K={K},S={S})")
    _l.check(lib.a3v_splitk_reduce(_p(scratch), S, M, N, _p(out), out.stride(0), dt(out), 1 if accumulate else 0, _stream()), "a3v_splitk_reduce")
    return out


def gemm_skinny_split(M: int, N: int, K: int) -> int:
    return _l.load().a3v_gemm_skinny_split(M, N, K)


def gemm_skinny_ws_bytes(M: int, N: int, K: int) -> int:
    return int(_l.load().a3v_gemm_skinny_ws_bytes(M, N, K))


def gemm_skinny_workspace(M: int, N: int, K: int, device) -> "torch.Tensor":
    """Zero-filled workspace for gemm_skinny (arrival counters first; every call leaves them zero)."""
    import torch
    return torch.zeros((gemm_skinny_ws_bytes(M, N, K) + 3) // 4, dtype=torch.float32, device=device)


def gemm_skinny(a, w, out, partial, *, residual=None, epilogue: int = 0):
    _dev(a, w, out, partial, residual)
    M, K = a.shape
    N = w.shape[0]
    if partial.numel() * partial.element_size() < gemm_skinny_ws_bytes(M, N, K):
        raise ValueError("gemm_skinny workspace too small (a3v_gemm_skinny_ws_bytes)")
    ep = epilogue | (EPI_RESIDUAL if residual is not None else 0)
    rc = _l.load().a3v_gemm_skinny(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K,
                                   _p(residual), residual.stride(0) if residual is not None else 0, ep,
                                   _p(partial), _stream())
    _l.check(rc, f"a3v_gemm_skinny(M={M},N={N},K={K},epi={ep})")
    return out


def quantize_rows_fp8(x, q, scales, norm_w=None, eps: float = 0.0):
    """q[r] = fp8(y[r] / scales[r]), scales[r] = max|y[r]| / 448; y = x, or the bf16 RMSNorm of x when norm_w is given."""
    _dev(x, q, scales, norm_w)
    rows, dim = x.shape
    assert q.dtype == torch.uint8 and scales.dtype == torch.float32 and scales.numel() >= rows
    rc = _l.load().a3v_quantize_rows_fp8(_p(x), x.stride(0), _p(norm_w), eps, _p(q), q.stride(0), _p(scales), rows, dim, dt(x), _stream())
    _l.check(rc, "a3v_quantize_rows_fp8")
    return q, scales


def gemm_nt_fp8(aq, sa, wq, sw, out, *, bias=None, residual=None, epilogue: int = 0):
    """out = epilogue((aq @ wq.T) * sa[:, None] * sw[None, :]) with fp8 operands (uint8 views of e4m3fn) on the MX-scaled MFMA."""
    _dev(aq, sa, wq, sw, out, bias, residual)
    M, K = aq.shape
    N = wq.shape[0]
    assert aq.dtype == torch.uint8 and wq.dtype == torch.uint8 and wq.shape[1] == K
    _ensure_gemm_workspace(aq.device)
    ep = epilogue
    if bias is not None:
        ep |= EPI_BIAS
    if residual is not None and not (ep & EPI_RES_F32):
        ep |= EPI_RESIDUAL
    rc = _l.load().a3v_gemm_nt_fp8(_p(aq), aq.stride(0), _p(sa), _p(wq), wq.stride(0), _p(sw), _p(out), out.stride(0), M, N, K,
                                   _p(bias), _p(residual), residual.stride(0) if residual is not None else 0, ep, _stream())
    _l.check(rc, f"a3v_gemm_nt_fp8(M={M},N={N},K={K},epi={ep})")
    return out


def gemm_qkv_rope_fp8(xq, sx, wq, sw, qkv, k_cache, vt_cache, cos_sin, B, S, H, Hkv, hd, start_pos, rope_pos0):
    _dev(xq, sx, wq, sw, qkv, k_cache, vt_cache, cos_sin)
    assert xq.shape[0] == B * S and wq.shape[0] == (H + 2 * Hkv) * hd
    rc = _l.load().a3v_gemm_qkv_rope_fp8(_p(xq), xq.stride(0), _p(sx), _p(wq), wq.stride(0), _p(sw), xq.shape[1], _p(qkv),
                                         qkv.stride(0), _p(k_cache), _p(vt_cache), _p(cos_sin), B, S, H, Hkv, hd,
                                         k_cache.shape[2], start_pos, rope_pos0, _stream())
    _l.check(rc, "a3v_gemm_qkv_rope_fp8")


def gemm_skinny_fp8(a, wq, wscale, out, workspace, *, residual=None, epilogue: int = 0):
    """out = epilogue((a . float(wq)^T) * wscale): weight-only fp8 (torch.float8_e4m3fn / uint8 bytes) decode GEMV."""
    _dev(a, wq, wscale, out, workspace, residual)
    M, K = a.shape
    N = wq.shape[0]
    assert wq.element_size() == 1 and wscale.dtype == torch.float32 and wscale.numel() == N
    if workspace.numel() * workspace.element_size() < gemm_skinny_ws_bytes(M, N, K):
        raise ValueError("gemm_skinny workspace too small (a3v_gemm_skinny_ws_bytes)")
    ep = epilogue | (EPI_RESIDUAL if residual is not None else 0)
    rc = _l.load().a3v_gemm_skinny_fp8(_p(a), a.stride(0), _p(wq), wq.stride(0), _p(wscale), _p(out), out.stride(0), M, N, K,
                                       _p(residual), residual.stride(0) if residual is not None else 0, ep, _p(workspace), _stream())
    _l.check(rc, f"a3v_gemm_skinny_fp8(M={M},N={N},K={K},epi={ep})")
    return out


def rmsnorm(x, w, out, eps: float):
    _dev(x, w, out)
    rows, dim = x.shape
    rc = _l.load().a3v_rmsnorm(_p(x), x.stride(0), _p(w), _p(out), out.stride(0), rows, dim, eps,
                               dt(x), dt(w), dt(out), _stream())
    _l.check(rc, "a3v_rmsnorm")
    return out


def layernorm(x, w, b, out, eps: float = 1e-5, row_map=None, rows: Optional[int] = None):
    _dev(x, w, b, out, row_map)
    n, dim = x.shape
    rc = _l.load().a3v_layernorm(_p(x), x.stride(0), _p(w), _p(b), _p(out), out.stride(0), _p(row_map),
                                 n if rows is None else rows, dim, eps, dt(x), dt(w), dt(out), _stream())
    _l.check(rc, "a3v_layernorm")
    return out


def rope_kvcache(qkv, q_out, k_cache, vt_cache, cos_sin, B, S, H, Hkv, hd, start_pos, rope_pos0):
    _dev(qkv, q_out, k_cache, vt_cache, cos_sin)
    Smax = k_cache.shape[2]
    rc = _l.load().a3v_rope_kvcache(_p(qkv), qkv.stride(0), _p(q_out), q_out.stride(0), _p(k_cache), _p(vt_cache),
                                    _p(cos_sin), B, S, H, Hkv, hd, Smax, start_pos, rope_pos0, dt(qkv), _stream())
    _l.check(rc, "a3v_rope_kvcache")


def gemm_qkv_rope(x, wqkv, qkv, k_cache, vt_cache, cos_sin, B, S, H, Hkv, hd, start_pos, rope_pos0, v_rows=None, delta=None):
    """qkv GEMM with RoPE + KV-cache write in the epilogue: rotated q lands in qkv[:, :H*hd] (further columns are not
    written), rotated k in k_cache, v transposed in vt_cache (and token-major in v_rows if given) -- the values of gemm_nt
    followed by rope_kvcache.  ``delta`` [B*S, N] (bf16): added to the projection before the rotation (the LoRA branch)."""
    _dev(x, wqkv, qkv, k_cache, vt_cache, cos_sin, v_rows, delta)
    assert x.shape[0] == B * S and wqkv.shape[0] == (H + 2 * Hkv) * hd and x.dtype == torch.bfloat16
    rc = _l.load().a3v_gemm_qkv_rope(_p(x), x.stride(0), _p(wqkv), wqkv.stride(0), x.shape[1], _p(qkv), qkv.stride(0),
                                     _p(k_cache), _p(vt_cache), _p(v_rows), v_rows.stride(0) if v_rows is not None else 0,
                                     _p(delta), delta.stride(0) if delta is not None else 0,
                                     _p(cos_sin), B, S, H, Hkv, hd, k_cache.shape[2], start_pos, rope_pos0, _stream())
    _l.check(rc, "a3v_gemm_qkv_rope")


def vt_pack(v, ldv, vt, N, L, H, hd, Lpad):
    _dev(v, vt)
    rc = _l.load().a3v_vt_pack(_p(v), ldv, _p(vt), N, L, H, hd, Lpad, dt(v), _stream())
    _l.check(rc, "a3v_vt_pack")


def attention_scratch_floats(B, H, hd, Sk) -> int:
    return _l.load().a3v_attention_scratch_floats(B, H, hd, Sk)


_Strides = ctypes.c_int64 * 12


def attention(q, k, vt, out, B, Sq, Sk, H, Hkv, hd, strides, causal: bool, scratch=None):
    _dev(q, k, vt, out, scratch)
    rc = _l.load().a3v_attention(_p(q), _p(k), _p(vt), _p(out), B, Sq, Sk, H, Hkv, hd, _Strides(*strides),
                                 1 if causal else 0, _p(scratch), dt(q), _stream())
    _l.check(rc, f"a3v_attention(B={B},Sq={Sq},Sk={Sk},H={H},hd={hd})")
    return out


def embed_assemble(tokens, table, h, B, T, W, dim):
    _dev(tokens, table, h)
    assert tokens.dtype == torch.int64 and tokens.stride(1) == 1
    rc = _l.load().a3v_embed_assemble(_p(tokens), tokens.stride(0), _p(table), _p(h), B, T, W, dim, table.shape[0],
                                      dt(table), dt(h), _stream())
    _l.check(rc, "a3v_embed_assemble")


def fill_rows(src, dst, row_idx):
    _dev(src, dst, row_idx)
    assert row_idx.dtype == torch.int32
    rc = _l.load().a3v_fill_rows(_p(src), _p(dst), dst.stride(0), _p(row_idx), row_idx.numel(), dst.shape[1],
                                 dt(src), dt(dst), _stream())
    _l.check(rc, "a3v_fill_rows")


def patch_im2col(img, cols, P):
    _dev(img, cols)
    N, C, Hi, Wi = img.shape
    assert C == 3 and img.is_contiguous()
    rc = _l.load().a3v_patch_im2col(_p(img), _p(cols), N, Hi, Wi, P, cols.shape[1], dt(img), dt(cols), _stream())
    _l.check(rc, "a3v_patch_im2col")


def split_views(img, out):
    _dev(img, out)
    B, C, S2, _ = img.shape
    assert C == 3 and img.is_contiguous() and out.is_contiguous()
    rc = _l.load().a3v_split_views(_p(img), _p(out), B, S2 // 2, dt(img), dt(out), _stream())
    _l.check(rc, "a3v_split_views")


def vit_embed(patch, cls, pos, x, N, T, width):
    _dev(patch, cls, pos, x)
    rc = _l.load().a3v_vit_embed(_p(patch), _p(cls), _p(pos), _p(x), N, T, width, dt(patch), _stream())
    _l.check(rc, "a3v_vit_embed")


def generate_step(logits, sampled, tokens, text_mask, cur_pos: int, stop_seq, stop_off, n_stop: int, stopped, stop_pos, live):
    """One launch = one step of MetaModel.generate's bookkeeping (a3v_generate_step): argmax (or the sampled ids), teacher
    forcing, tokens[:, cur_pos] write, stop-sequence match, stop_pos / stopped / live update."""
    _dev(tokens, text_mask, stopped, stop_pos, live)
    B = tokens.shape[0]
    assert tokens.dtype == torch.int64 and text_mask.dtype == torch.bool and stopped.dtype == torch.bool and stop_pos.dtype == torch.int64
    assert tokens.stride(1) == 1 and text_mask.stride(1) == 1 and live.dtype == torch.int32
    if sampled is None:
        assert logits.dtype == torch.float32 and logits.stride(1) == 1 and logits.shape[0] == B
        lp, ld, V = _p(logits), logits.stride(0), logits.shape[1]
    else:
        assert sampled.dtype == torch.int64 and sampled.is_contiguous() and sampled.numel() == B
        lp, ld, V = None, 0, 1
    rc = _l.load().a3v_generate_step(lp, ld, _p(sampled), B, V, _p(tokens), tokens.stride(0), _p(text_mask), text_mask.stride(0), int(cur_pos),
                                     _p(stop_seq), _p(stop_off), int(n_stop), _p(stopped), _p(stop_pos), _p(live), _stream())
    _l.check(rc, "a3v_generate_step")


def sample_top_p(logits, temperature: float, top_p: float, u, out):
    """out[b] = one draw from the top-p nucleus of softmax(logits[b] / temperature) at the uniform number u[b] (a3v_sample_top_p;
    model/meta.py:456-459, 568-583 without the full-vocabulary sort)."""
    _dev(logits, u, out)
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and u.dtype == torch.float32 and out.dtype == torch.int64
    B, V = logits.shape
    assert u.numel() >= B and out.numel() >= B and u.is_contiguous() and out.is_contiguous()
    rc = _l.load().a3v_sample_top_p(_p(logits), logits.stride(0), B, V, float(temperature), float(top_p), _p(u), _p(out), _stream())
    _l.check(rc, "a3v_sample_top_p")
    return out


def argmax(logits, out):
    _dev(logits, out)
    assert logits.dtype == torch.float32 and out.dtype == torch.int64
    B, V = logits.shape
    rc = _l.load().a3v_argmax(_p(logits), logits.stride(0), _p(out), B, V, _stream())
    _l.check(rc, "a3v_argmax")
    return out


def count_valid(labels, n_valid):
    _dev(labels, n_valid)
    rc = _l.load().a3v_count_valid(_p(labels), labels.numel(), _p(n_valid), _stream())
    _l.check(rc, "a3v_count_valid")


def cross_entropy(logits, labels, row_loss, dlogits=None, n_valid=None, grad_scale: float = 1.0):
    _dev(logits, labels, row_loss, dlogits, n_valid)
    rows, V = logits.shape
    rc = _l.load().a3v_cross_entropy(_p(logits), logits.stride(0), _p(labels), _p(row_loss), _p(dlogits),
                                     dlogits.stride(0) if dlogits is not None else 0, _p(n_valid), grad_scale,
                                     rows, V, dt(logits), _stream())
    _l.check(rc, "a3v_cross_entropy")


# ------------------------------------------------------------------ training (backward) ops
def attention_lse(q, k, vt, out, lse, B, Sq, Sk, H, Hkv, hd, strides, causal: bool):
    _dev(q, k, vt, out, lse)
    rc = _l.load().a3v_attention_lse(_p(q), _p(k), _p(vt), _p(out), _p(lse), B, Sq, Sk, H, Hkv, hd, _Strides(*strides),
                                     1 if causal else 0, dt(q), _stream())
    _l.check(rc, "a3v_attention_lse")


def transpose(src, dst, R, C, Rpad, batch=1, bs_src=0, bs_dst=0):
    """dst[b, c, r] = src[b, r, c]; src/dst 2-D (or batched through element strides)."""
    _dev(src, dst)
    rc = _l.load().a3v_transpose(_p(src), src.stride(-2), bs_src, _p(dst), dst.stride(-2), bs_dst, R, C, Rpad, batch,
                                 dt(src), _stream())
    _l.check(rc, "a3v_transpose")
    return dst


def cast(src, dst):
    _dev(src, dst)
    rows, cols = src.shape
    rc = _l.load().a3v_cast(_p(src), src.stride(0), dt(src), _p(dst), dst.stride(0), dt(dst), rows, cols, _stream())
    _l.check(rc, "a3v_cast")
    return dst


def scale_cast(src, dst, scale: float = 1.0):
    """dst = (dst.dtype)(src * scale) over contiguous 1-D buffers of equal length (DP gradient wire conversions)."""
    _dev(src, dst)
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    rc = _l.load().a3v_scale_cast(_p(src), dt(src), _p(dst), dt(dst), src.numel(), float(scale), _stream())
    _l.check(rc, "a3v_scale_cast")
    return dst


SUMSQ_SLOTS = 1024


def sumsq_partials(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[0 .. SUMSQ_SLOTS) = partial sums of squares of the fp32 vector x (deterministic; the clip's norm = sqrt of their sum)."""
    _dev(x, out)
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and out.numel() >= SUMSQ_SLOTS and x.is_contiguous() and out.is_contiguous()
    rc = _l.load().a3v_sumsq_partials(_p(x), x.numel(), _p(out), _stream())
    _l.check(rc, "a3v_sumsq_partials")
    return out


def add2d(dst, src):
    """dst += src (2-D blocks of one dtype, arbitrary row strides)."""
    _dev(dst, src)
    assert dst.dtype == src.dtype and dst.shape == src.shape
    rows, cols = dst.shape
    rc = _l.load().a3v_add2d(_p(dst), dst.stride(0), _p(src), src.stride(0), rows, cols, dt(dst), _stream())
    _l.check(rc, "a3v_add2d")
    return dst


_rb_scratch = {}


def rmsnorm_bwd(x, w, dy, dh, dw, eps, dh_lowp=None):
    """dh += d(rmsnorm)/dx . dy (fp32 stream), dw += ...; ``dh_lowp``: the updated dh also as bf16 (the next GEMMs' operand)."""
    _dev(x, w, dy, dh, dw, dh_lowp)
    assert w.dtype == torch.float32 and x.dtype == dh.dtype and x.dtype in (torch.float32, torch.bfloat16)
    rows, dim = x.shape
    scratch = None
    if dw is not None:                       # per-device scratch for the weight-gradient partial rows (grown on demand)
        need = int(_l.load().a3v_rmsnorm_bwd_scratch_floats(rows, dim))
        scratch = _rb_scratch.get(x.device)
        if scratch is None or scratch.numel() < need:
            scratch = torch.empty(need, dtype=torch.float32, device=x.device)
            _rb_scratch[x.device] = scratch
    if x.dtype == torch.bfloat16:            # bf16 residual stream: x, dy and the accumulated dh are bf16 (fp32 arithmetic inside)
        assert dh_lowp is None and dy.dtype == torch.bfloat16
        rc = _l.load().a3v_rmsnorm_bwd_bf16(_p(x), x.stride(0), _p(w), _p(dy), dy.stride(0), _p(dh), dh.stride(0), _p(dw), _p(scratch), rows,
                                            dim, eps, _stream())
        _l.check(rc, "a3v_rmsnorm_bwd_bf16")
        return
    if dh_lowp is not None:
        assert dh_lowp.dtype == torch.bfloat16 and dh_lowp.shape == dh.shape
        rc = _l.load().a3v_rmsnorm_bwd_cast(_p(x), x.stride(0), _p(w), _p(dy), dy.stride(0), _p(dh), dh.stride(0), _p(dw), _p(scratch), rows,
                                            dim, eps, dt(dy), _p(dh_lowp), dh_lowp.stride(0), _stream())
        _l.check(rc, "a3v_rmsnorm_bwd_cast")
        return
    rc = _l.load().a3v_rmsnorm_bwd(_p(x), x.stride(0), _p(w), _p(dy), dy.stride(0), _p(dh), dh.stride(0), _p(dw), _p(scratch), rows, dim,
                                   eps, dt(dy), _stream())
    _l.check(rc, "a3v_rmsnorm_bwd")


def layernorm_bwd(x, w, dy, row_map, dx, dw, db, eps=1e-5):
    _dev(x, w, dy, row_map, dx, dw, db)
    rows, dim = x.shape
    if dy.dtype == torch.bfloat16:           # gradient rows gathered from a bf16 residual stream
        rc = _l.load().a3v_layernorm_bwd_bf16(_p(x), x.stride(0), _p(w), _p(dy), dy.stride(0), _p(row_map), _p(dx), dx.stride(0),
                                              _p(dw), _p(db), rows, dim, eps, _stream())
        _l.check(rc, "a3v_layernorm_bwd_bf16")
        return
    rc = _l.load().a3v_layernorm_bwd(_p(x), x.stride(0), _p(w), _p(dy), dy.stride(0), _p(row_map), _p(dx), dx.stride(0),
                                     _p(dw), _p(db), rows, dim, eps, dt(x), _stream())
    _l.check(rc, "a3v_layernorm_bwd")


def swiglu_fwd(gu, act, F, interleaved: bool):
    _dev(gu, act)
    rc = _l.load().a3v_swiglu_fwd(_p(gu), gu.stride(0), _p(act), act.stride(0), gu.shape[0], F, 1 if interleaved else 0,
                                  dt(gu), _stream())
    _l.check(rc, "a3v_swiglu_fwd")


def swiglu_bwd(gu, dact, dgu, F, interleaved: bool):
    _dev(gu, dact, dgu)
    rc = _l.load().a3v_swiglu_bwd(_p(gu), gu.stride(0), _p(dact), dact.stride(0), _p(dgu), dgu.stride(0), gu.shape[0], F,
                                  1 if interleaved else 0, dt(gu), _stream())
    _l.check(rc, "a3v_swiglu_bwd")


def rope_bwd_pack(dq, dk, dv, dqkv, cos_sin, B, S, H, Hkv, hd, rope_pos0=0):
    _dev(dq, dk, dv, dqkv, cos_sin)
    rc = _l.load().a3v_rope_bwd_pack(_p(dq), _p(dk), _p(dv), _p(dqkv), dqkv.stride(0), _p(cos_sin), B, S, H, Hkv, hd,
                                     rope_pos0, dt(dq), _stream())
    _l.check(rc, "a3v_rope_bwd_pack")


def attention_bwd_workspace_bytes(B, S, H, Hkv, hd) -> int:
    return _l.load().a3v_attention_bwd_workspace_bytes(B, S, H, Hkv, hd)


def attention_bwd(q, k, k_sb, k_sh, v, v_sb, v_ss, v_sh, out, dout, lse, D, dq, dk, dv, B, S, H, Hkv, hd, causal: bool,
                  workspace=None):
    _dev(q, k, v, out, dout, lse, D, dq, dk, dv, workspace)
    rc = _l.load().a3v_attention_bwd(_p(q), _p(k), k_sb, k_sh, _p(v), v_sb, v_ss, v_sh, _p(out), _p(dout), _p(lse), _p(D),
                                     _p(dq), _p(dk), _p(dv), _p(workspace), B, S, H, Hkv, hd, 1 if causal else 0, dt(q), _stream())
    _l.check(rc, "a3v_attention_bwd")


def attention_bwd_packed(q, k, k_sb, k_sh, v, v_sb, v_ss, v_sh, out, dout, lse, D, dqkv, cos_sin, B, S, H, Hkv, hd, causal: bool,
                         rope_pos0: int = 0):
    """attention backward + inverse RoPE + packing into the fused-qkv gradient in one pass (bf16, hd 64 / 128)."""
    _dev(q, k, v, out, dout, lse, D, dqkv, cos_sin)
    ld_out = out.stride(-2) if out.dim() == 2 else out.stride(1)      # token stride of out ([rows, H*hd (+pad)] or [B, S, H, hd])
    rc = _l.load().a3v_attention_bwd_packed(_p(q), _p(k), k_sb, k_sh, _p(v), v_sb, v_ss, v_sh, _p(out), ld_out, _p(dout), _p(lse), _p(D),
                                            _p(dqkv), dqkv.stride(0), _p(cos_sin), rope_pos0, B, S, H, Hkv, hd, 1 if causal else 0,
                                            dt(q), _stream())
    _l.check(rc, "a3v_attention_bwd_packed")


def embed_bwd(tokens, dh, dtable, B, T, W, dim):
    _dev(tokens, dh, dtable)
    assert dh.is_contiguous()
    fn = _l.load().a3v_embed_bwd_bf16 if dh.dtype == torch.bfloat16 else _l.load().a3v_embed_bwd
    rc = fn(_p(tokens), tokens.stride(0), _p(dh), _p(dtable), B, T, W, dim, dtable.shape[0], _stream())
    _l.check(rc, "a3v_embed_bwd")


def rows_sum(src, row_idx, n_rows, out):
    _dev(src, row_idx, out)
    rc = _l.load().a3v_rows_sum(_p(src), src.stride(0), _p(row_idx), n_rows, src.shape[1], _p(out), dt(src), _stream())
    _l.check(rc, "a3v_rows_sum")


def lora_gb_scatter(gbt, r: int, views, row0s):
    """views[j] ([n_j, r] fp32 contiguous) += gbt[j*r:(j+1)*r, row0s[j]:row0s[j]+n_j].T for the modules of a fused adapter group."""
    import ctypes
    _dev(gbt, *views)
    n = len(views)
    assert gbt.dtype == torch.float32 and gbt.stride(1) == 1 and all(v.dtype == torch.float32 and v.is_contiguous() and v.shape[1] == r for v in views)
    dst = (ctypes.c_void_p * n)(*[v.data_ptr() for v in views])
    r0 = (ctypes.c_int * n)(*[int(x) for x in row0s])
    nj = (ctypes.c_int * n)(*[int(v.shape[0]) for v in views])
    rc = _l.load().a3v_lora_gb_scatter(_p(gbt), gbt.stride(0), int(r), n, ctypes.cast(dst, ctypes.c_void_p), ctypes.cast(r0, ctypes.c_void_p),
                                       ctypes.cast(nj, ctypes.c_void_p), _stream())
    _l.check(rc, "a3v_lora_gb_scatter")


def lora_refresh(wa, wb, A, At, B, Bt, col0: int, row0: int):
    """One adapter's rows / columns of the fused group images (A, At, B, Bt; bf16) from its fp32 lora_a [r, in], lora_b [nj, r]."""
    _dev(wa, wb, A, At, B, Bt)
    assert wa.dtype == torch.float32 and wb.dtype == torch.float32 and wa.is_contiguous() and wb.is_contiguous()
    assert all(t.dtype == torch.bfloat16 and t.stride(1) == 1 for t in (A, At, B, Bt))
    r, in_f = wa.shape
    nj = wb.shape[0]
    assert wb.shape[1] == r and col0 + r <= A.shape[0] and in_f <= A.shape[1] and row0 + nj <= B.shape[0]
    rc = _l.load().a3v_lora_refresh(_p(wa), _p(wb), r, in_f, nj, _p(A), A.stride(0), _p(At), At.stride(0), _p(B), B.stride(0), _p(Bt),
                                    Bt.stride(0), col0, row0, _stream())
    _l.check(rc, "a3v_lora_refresh")
