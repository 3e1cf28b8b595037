// One greedy-decode step of the Llama decoder stack (seqlen == 1, llama_ens5.py:490-531 with
// mask == None) issued from ONE host call: every kernel of every layer is enqueued back to back on
// the caller's stream, so the per-kernel host cost is a C++ launch (~2-3 us) instead of a Python
// ctypes round trip -- at batch 8 the step is HBM-bound on the GPU only if the host keeps up.
#include "a3v_common.h"

extern "C" int a3v_llama_decode_step(const a3v_llama_layer* layers, int n_layers, void* h, void* xn, void* qkv, void* att,
                                     void* act, float* attn_scratch, void* skinny_ws, const float* cos_sin, int B, int dim, int H, int Hkv,
                                     int hd, int ffn, int Smax, int pos, float eps, void* stream) {
  if (!layers || n_layers <= 0 || !h || !xn || !qkv || !att || !act || !attn_scratch || !skinny_ws || !cos_sin) return A3V_ERR_ARG;
  if (B <= 0 || B > 16 || pos < 0 || pos >= Smax) return A3V_ERR_SHAPE;
  const int64_t ldq = (int64_t)(H + 2 * Hkv) * hd;
  const int64_t strides[12] = {ldq, ldq, hd,
                               (int64_t)Hkv * Smax * hd, (int64_t)Smax * hd, hd,
                               (int64_t)Hkv * hd * Smax, (int64_t)hd * Smax, Smax,
                               (int64_t)H * hd, (int64_t)H * hd, hd};
  int rc;
  for (int i = 0; i < n_layers; ++i) {
    const a3v_llama_layer& L = layers[i];
    if ((rc = a3v_rmsnorm(h, dim, L.attn_norm_w, xn, dim, B, dim, eps, A3V_BF16, A3V_BF16, A3V_BF16, stream))) return rc;
    if ((rc = a3v_gemm_skinny(xn, dim, L.wqkv, dim, qkv, ldq, B, (int)ldq, dim, nullptr, 0, 0, skinny_ws, stream))) return rc;
    if ((rc = a3v_rope_kvcache(qkv, ldq, qkv, ldq, L.k_cache, L.vt_cache, cos_sin, B, 1, H, Hkv, hd, Smax, pos, pos, A3V_BF16, stream))) return rc;
    if ((rc = a3v_attention(qkv, L.k_cache, L.vt_cache, att, B, 1, pos + 1, H, Hkv, hd, strides, 0, attn_scratch, A3V_BF16, stream))) return rc;
    if ((rc = a3v_gemm_skinny(att, (int64_t)H * hd, L.wo, (int64_t)H * hd, h, dim, B, dim, H * hd, h, dim, A3V_EPI_RESIDUAL, skinny_ws, stream))) return rc;
    if ((rc = a3v_rmsnorm(h, dim, L.ffn_norm_w, xn, dim, B, dim, eps, A3V_BF16, A3V_BF16, A3V_BF16, stream))) return rc;
    if ((rc = a3v_gemm_skinny(xn, dim, L.w13, dim, act, ffn, B, 2 * ffn, dim, nullptr, 0, A3V_EPI_SWIGLU, skinny_ws, stream))) return rc;
    if ((rc = a3v_gemm_skinny(act, ffn, L.w2, ffn, h, dim, B, dim, ffn, h, dim, A3V_EPI_RESIDUAL, skinny_ws, stream))) return rc;
  }
  return A3V_OK;
}
