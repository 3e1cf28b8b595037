// One greedy-decode step of the Llama decoder stack (seqlen == 1, llama_ens5.py:490-531 with
// mask == None) issued from ONE host call: every kernel of every layer is enqueued back to back on
// the caller's stream, so the per-kernel host cost is a C++ launch (~2-3 us) instead of a Python
// ctypes round trip -- at batch 8 the step is HBM-bound on the GPU only if the host keeps up.
//
// Fused form (dim, ffn, H*hd multiples of 128; hd in {64, 128}): FIVE launches per layer instead of ten --
//   qkv GEMV  [RMSNorm prologue | RoPE + KV-cache epilogue]      (was rmsnorm, GEMV, rope)
//   attention [split-KV, combine folded in by the last-arriving block]   (was attention, combine)
//   wo GEMV   [+residual, emits per-tile sums of squares of the new h]
//   w1|w3 GEMV [RMSNorm prologue from those sums | SwiGLU]       (was rmsnorm, GEMV)
//   w2 GEMV   [+residual, sums of squares for the next layer's prologue]
// The small kernels were ~5-7 us each of pure launch + latency: 24 of ~146 us per 7B layer.
#include "a3v_common.h"

namespace {
// sums of squares of the rows of h [B, dim] in the producer layout ([dim/16][16]) for the first layer's prologue
__global__ __launch_bounds__(256) void rows_ssq_kernel(const bf16_t* __restrict__ h, int64_t ld, int B, int dim, float* __restrict__ ssq) {
  const int t = blockIdx.x * 256 + threadIdx.x;         // (tile, m)
  const int tile = t >> 4, m = t & 15;
  if (tile * 16 >= dim) return;
  float s = 0.f;
  if (m < B) {
    const bf16_t* r = h + (int64_t)m * ld + tile * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float x = (float)r[e]; s = fmaf(x, x, s); }
  }
  ssq[t] = s;
}
}  // namespace

// Which form a3v_llama_decode_step takes for a geometry, decided BEFORE anything is launched: 2 = the fused five-launch form, 1 = the
// per-kernel form (<= 16 rows, bf16 weights), 0 = not taken (A3V_ERR_SHAPE up front; the host runs its general path on untouched state).
static int decode_step_form(int B, int dim, int H, int Hkv, int hd, int ffn, int w8) {
  if (B <= 0 || B > 32) return 0;
  const int64_t ldq = (int64_t)(H + 2 * Hkv) * hd;
  const int Bc = B > 16 ? (B + 1) / 2 : B;
  const bool fused = (hd == 64 || hd == 128) && dim % 16 == 0 && dim / 16 * 16 * 4 <= A3V_WS_PARTIALS - A3V_WS_SSQ &&
                     a3v_gemv_supported(Bc, (int)ldq, dim, 0, w8) && a3v_gemv_supported(Bc, dim, H * hd, A3V_EPI_RESIDUAL, w8) &&
                     a3v_gemv_supported(Bc, 2 * ffn, dim, A3V_EPI_SWIGLU, w8) && a3v_gemv_supported(Bc, dim, ffn, A3V_EPI_RESIDUAL, w8) &&
                     ldq % 16 == 0 && (2 * ffn) % 32 == 0 && Bc * H * (int)sizeof(int) <= A3V_WS_SSQ - A3V_WS_ATTN_COUNTERS;
  if (fused) return 2;
  return (w8 || B > 16) ? 0 : 1;
}
extern "C" int a3v_llama_decode_step_form(int B, int dim, int H, int Hkv, int hd, int ffn, int w8) { return decode_step_form(B, dim, H, Hkv, hd, ffn, w8); }

extern "C" int a3v_llama_decode_step(const a3v_llama_layer* layers, int n_layers, void* h, void* xn, void* qkv, void* att,
                                     void* act, float* attn_scratch, void* skinny_ws, const float* cos_sin, int B, int dim, int H, int Hkv,
                                     int hd, int ffn, int Smax, int pos, float eps, void* stream) {
  if (!layers || n_layers <= 0 || !h || !xn || !qkv || !att || !act || !attn_scratch || !skinny_ws || !cos_sin) return A3V_ERR_ARG;
  if (B <= 0 || B > 32 || pos < 0 || pos >= Smax) return A3V_ERR_SHAPE;   // the plugin contract's max_batch_size = 32 (meta.py:34-52)
  const int64_t ldq = (int64_t)(H + 2 * Hkv) * hd;
  const int64_t strides[12] = {ldq, ldq, hd,
                               (int64_t)Hkv * Smax * hd, (int64_t)Smax * hd, hd,
                               (int64_t)Hkv * hd * Smax, (int64_t)hd * Smax, Smax,
                               (int64_t)H * hd, (int64_t)H * hd, hd};
  int rc;
  const int w8 = layers[0].wqkv_q != nullptr;
  for (int i = 0; i < n_layers; ++i) {
    const a3v_llama_layer& L = layers[i];
    const int q8 = (L.wqkv_q != nullptr) + (L.wo_q != nullptr) + (L.w13_q != nullptr) + (L.w2_q != nullptr);
    if (q8 != (w8 ? 4 : 0) || (w8 && (!L.wqkv_s || !L.wo_s || !L.w13_s || !L.w2_s))) return A3V_ERR_ARG;   // all four or none
  }
  // The GEMV kernels take up to 16 activation rows.  Batch rows are independent through the whole stack (each has its own KV
  // cache rows), so a batch of 17..32 runs as two row chunks through the same fused launches: the weights are streamed once per
  // chunk (a 32-row batch costs two 16-row steps, i.e. the 16-row tok/s), every buffer is addressed at its row offset.
  const int Bc = B > 16 ? (B + 1) / 2 : B;
  const int form = decode_step_form(B, dim, H, Hkv, hd, ffn, w8);
  if (!form) return A3V_ERR_SHAPE;                 // fp8 images / 17..32 rows exist only in the fused form (the host takes its general path)
  const bool fused = form == 2;
  if (fused) {
    float* ssq = (float*)((char*)skinny_ws + A3V_WS_SSQ);
    int* actr = (int*)((char*)skinny_ws + A3V_WS_ATTN_COUNTERS);
    const int64_t kv_b = (int64_t)Hkv * Smax * hd;       // elements per batch row of a K / V^T cache
    for (int b0 = 0; b0 < B; b0 += Bc) {
      const int nb = B - b0 < Bc ? B - b0 : Bc;
      bf16_t* hc = (bf16_t*)h + (int64_t)b0 * dim;
      bf16_t* qc = (bf16_t*)qkv + (int64_t)b0 * ldq;
      bf16_t* ac = (bf16_t*)att + (int64_t)b0 * H * hd;
      bf16_t* fc = (bf16_t*)act + (int64_t)b0 * ffn;
      hipLaunchKernelGGL(rows_ssq_kernel, dim3((dim / 16 * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)hc, (int64_t)dim, nb, dim, ssq);
      A3V_LAUNCH_CHECK();
      for (int i = 0; i < n_layers; ++i) {
        const a3v_llama_layer& L = layers[i];
        bf16_t* kc = (bf16_t*)L.k_cache + b0 * kv_b;
        bf16_t* vc = (bf16_t*)L.vt_cache + b0 * kv_b;
        if ((rc = a3v_gemv_fused(hc, dim, w8 ? L.wqkv_q : L.wqkv, dim, L.wqkv_s, qc, ldq, nb, (int)ldq, dim, nullptr, 0, 0, L.attn_norm_w, ssq, eps,
                                 nullptr, 1, cos_sin, kc, vc, H, Hkv, hd, Smax, pos, skinny_ws, stream))) return rc;
        if ((rc = a3v_attention_decode_fused(qc, kc, vc, ac, nb, pos + 1, H, Hkv, hd, strides, attn_scratch, actr, stream))) return rc;
        if ((rc = a3v_gemv_fused(ac, (int64_t)H * hd, w8 ? L.wo_q : L.wo, (int64_t)H * hd, L.wo_s, hc, dim, nb, dim, H * hd, hc, dim, A3V_EPI_RESIDUAL,
                                 nullptr, nullptr, eps, ssq, 0, nullptr, nullptr, nullptr, H, Hkv, hd, Smax, pos, skinny_ws, stream))) return rc;
        if ((rc = a3v_gemv_fused(hc, dim, w8 ? L.w13_q : L.w13, dim, L.w13_s, fc, ffn, nb, 2 * ffn, dim, nullptr, 0, A3V_EPI_SWIGLU, L.ffn_norm_w, ssq,
                                 eps, nullptr, 0, nullptr, nullptr, nullptr, H, Hkv, hd, Smax, pos, skinny_ws, stream))) return rc;
        if ((rc = a3v_gemv_fused(fc, ffn, w8 ? L.w2_q : L.w2, ffn, L.w2_s, hc, dim, nb, dim, ffn, hc, dim, A3V_EPI_RESIDUAL, nullptr, nullptr, eps, ssq,
                                 0, nullptr, nullptr, nullptr, H, Hkv, hd, Smax, pos, skinny_ws, stream))) return rc;
      }
    }
    return A3V_OK;
  }
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
