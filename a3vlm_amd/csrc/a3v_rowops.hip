// HBM-bound row kernels of the path (gfx950): norms, RoPE + KV-cache write, embedding /
// sequence assembly, patch im2col, view split, argmax, cross-entropy.
// All are one-pass streaming kernels with 16-byte per-lane accesses; each row is read once
// and written once (algorithmic bytes = in + out), reductions via wave shuffles.
#include "a3v_common.h"

namespace {

// ---------------------------------------------------------------- RMSNorm
// model/components.py:39,52-53: (x.float() * rsqrt(mean(x^2)+eps)).type_as(x) * weight
template <typename TX, typename TW, typename TY>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const TX* __restrict__ x, int64_t ldx, const TW* __restrict__ w,
                                                      TY* __restrict__ y, int64_t ldy, int dim, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const TX* xr = x + (int64_t)row * ldx;
  TY* yr = y + (int64_t)row * ldy;
  constexpr int MAXV = 4;  // dim <= 256*8*4 = 8192
  float v[MAXV][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c < dim) {
      load8(xr + c, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf(v[i][e], v[i][e], ss);
    }
  }
  ss = block_sum<256>(ss, red);
  const float inv = rsqrtf(ss / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c < dim) {
      float wv[8], o[8];
      load8(w + c, wv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = Cvt<TX>::rnd(v[i][e] * inv) * wv[e];
      store8(yr + c, o);
    }
  }
}

// ---------------------------------------------------------------- per-row fp8 (OCP e4m3fn) quantisation of activations
// q[row][k] = fp8(y[row][k] / scale[row]), scale[row] = max_k |y[row][k]| / 448, with y = the bf16 rows x themselves (NORM = false)
// or the bf16 RMSNorm output of x (NORM = true: the values rmsnorm_kernel<.., bf16_t> stores, never written to memory).
// Feeds a3v_gemm_nt_fp8 (a3v_gemm.hip), which multiplies the accumulator of row m by scale[m].
template <typename TX, typename TW, bool NORM>
__global__ __launch_bounds__(256) void rows_quant_fp8_kernel(const TX* __restrict__ x, int64_t ldx, const TW* __restrict__ w,
                                                             uint8_t* __restrict__ q, int64_t ldq, float* __restrict__ scales, int dim,
                                                             float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const TX* xr = x + (int64_t)row * ldx;
  constexpr int MAXV = 8;  // dim <= 256*8*8 = 16384
  float v[MAXV][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c < dim) {
      load8(xr + c, v[i]);
      if (NORM) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(v[i][e], v[i][e], ss);
      }
    }
  }
  float amax = 0.f;
  if (NORM) {
    ss = block_sum<256>(ss, red);
    const float inv = rsqrtf(ss / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (tid + i * 256) * 8;
      if (c < dim) {
        float wv[8];
        load8(w + c, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = rbf(Cvt<TX>::rnd(v[i][e] * inv) * wv[e]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c < dim) {
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float scale = fmaxf(amax, 1e-12f) * (1.f / 448.f);
  const float rs = 1.f / scale;
  if (tid == 0) scales[row] = scale;
  uint8_t* qr = q + (int64_t)row * ldq;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c < dim) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = fminf(fmaxf(v[i][e] * rs, -448.f), 448.f);
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], hi, true);
      *reinterpret_cast<int2*>(qr + c) = make_int2(lo, hi);
    }
  }
}

// ---------------------------------------------------------------- LayerNorm (torch.nn.LayerNorm)
template <typename T, typename TP, typename TY>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int64_t ldx, const TP* __restrict__ w,
                                                        const TP* __restrict__ b, TY* __restrict__ y, int64_t ldy,
                                                        const int32_t* __restrict__ row_map, int dim, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const T* xr = x + (int64_t)row * ldx;
  TY* yr = y + (int64_t)(row_map ? row_map[row] : row) * ldy;
  constexpr int MAXV = 4;
  float v[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c < dim) {
      load8(xr + c, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = block_sum<256>(s, red) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c < dim) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
    }
  }
  const float rstd = rsqrtf(block_sum<256>(q, red) / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c < dim) {
      float wv[8], bv[8], o[8];
      load8(w + c, wv);
      load8(b + c, bv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
      store8(yr + c, o);
    }
  }
}

// ---------------------------------------------------------------- RoPE + KV-cache write
// grid (ceil(S/64), H + 2*Hkv, B); block 256 handles 64 tokens x one head (hd <= 128).
// q heads: rotate -> q_out; k heads: rotate -> k_cache[b,hk,pos,:]; v heads: transpose
// through LDS -> vt_cache[b,hk,:,pos] (contiguous along pos).
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_kernel(const T* __restrict__ qkv, int64_t ldqkv, T* __restrict__ q_out,
                                                      int64_t ldq, T* __restrict__ k_cache, T* __restrict__ vt_cache,
                                                      const float* __restrict__ cos_sin, int S, int H, int Hkv, int hd,
                                                      int Smax, int start_pos, int rope_pos0) {
  __shared__ T tile[64][128 + 8];
  const int s0 = blockIdx.x * 64, slot = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int cpr = hd / 8;                 // 8-element chunks per row
  const int n_chunks = 64 * cpr;
  const T* src = qkv + (int64_t)b * S * ldqkv + (int64_t)slot * hd;
  if (slot < H + Hkv) {
    const bool is_q = slot < H;
    for (int id = tid; id < n_chunks; id += 256) {
      const int r = id / cpr, c = (id % cpr) * 8;
      const int s = s0 + r;
      if (s >= S) continue;
      float v[8], o[8];
      load8(src + (int64_t)s * ldqkv + c, v);
      const float* cs = cos_sin + ((int64_t)(rope_pos0 + s) * (hd / 2) + c / 2) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float co = cs[2 * e], si = cs[2 * e + 1];
        const float a = v[2 * e], bb = v[2 * e + 1];
        o[2 * e] = a * co - bb * si;
        o[2 * e + 1] = a * si + bb * co;
      }
      if (is_q) {
        store8(q_out + ((int64_t)b * S + s) * ldq + (int64_t)slot * hd + c, o);
      } else {
        const int hk = slot - H;
        store8(k_cache + (((int64_t)b * Hkv + hk) * Smax + start_pos + s) * hd + c, o);
      }
    }
    return;
  }
  // V: [64 tok][hd] -> vt[hd][64 tok]
  const int hk = slot - H - Hkv;
  for (int id = tid; id < n_chunks; id += 256) {
    const int r = id / cpr, c = (id % cpr) * 8;
    const int s = s0 + r;
    float v[8];
    if (s < S) load8(src + (int64_t)s * ldqkv + c, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) Cvt<T>::st(&tile[r][c + e], v[e]);
  }
  __syncthreads();
  T* dst = vt_cache + ((int64_t)b * Hkv + hk) * (int64_t)hd * Smax + start_pos + s0;
  const int nvalid = min(64, S - s0);
  // thread -> (d, 8 consecutive tokens); 16-B stores when the destination is aligned
  const bool aligned = ((start_pos + s0) % 8 == 0) && (Smax % 8 == 0);
  for (int id = tid; id < hd * 8; id += 256) {
    const int d = id >> 3, t0 = (id & 7) * 8;
    if (t0 >= nvalid) continue;
    T* dp = dst + (int64_t)d * Smax + t0;
    if (aligned && t0 + 8 <= nvalid) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = Cvt<T>::ld(&tile[t0 + e][d]);
      store8(dp, o);
    } else {
      for (int e = 0; e < 8 && t0 + e < nvalid; ++e) dp[e] = tile[t0 + e][d];
    }
  }
}

// Decode form (S <= 4 tokens per sequence): one 64-thread block per (slot, token, batch); no LDS tile.
template <typename T>
__global__ __launch_bounds__(64) void rope_kv_small_kernel(const T* __restrict__ qkv, int64_t ldqkv, T* __restrict__ q_out,
                                                           int64_t ldq, T* __restrict__ k_cache, T* __restrict__ vt_cache,
                                                           const float* __restrict__ cos_sin, int S, int H, int Hkv, int hd,
                                                           int Smax, int start_pos, int rope_pos0) {
  const int slot = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const T* src = qkv + ((int64_t)b * S + s) * ldqkv + (int64_t)slot * hd;
  const int half = hd / 2;
  for (int pr = threadIdx.x; pr < half; pr += 64) {
    const float a = Cvt<T>::ld(src + 2 * pr), bb = Cvt<T>::ld(src + 2 * pr + 1);
    if (slot < H + Hkv) {
      const float co = cos_sin[((int64_t)(rope_pos0 + s) * half + pr) * 2], si = cos_sin[((int64_t)(rope_pos0 + s) * half + pr) * 2 + 1];
      const float o0 = a * co - bb * si, o1 = a * si + bb * co;
      T* dst = slot < H ? q_out + ((int64_t)b * S + s) * ldq + (int64_t)slot * hd + 2 * pr
                        : k_cache + (((int64_t)b * Hkv + (slot - H)) * Smax + start_pos + s) * hd + 2 * pr;
      Cvt<T>::st(dst, o0);
      Cvt<T>::st(dst + 1, o1);
    } else {
      T* dst = vt_cache + (((int64_t)b * Hkv + (slot - H - Hkv)) * hd + 2 * pr) * (int64_t)Smax + start_pos + s;
      Cvt<T>::st(dst, a);
      Cvt<T>::st(dst + Smax, bb);
    }
  }
}

// v [N, L, H*hd] (row stride ldv) -> vt [N, H, hd, Lpad]   (ViT: transposed V for attention)
template <typename T>
__global__ __launch_bounds__(256) void vt_pack_kernel(const T* __restrict__ v, int64_t ldv, T* __restrict__ vt,
                                                      int L, int H, int hd, int Lpad) {
  __shared__ T tile[64][128 + 8];
  const int s0 = blockIdx.x * 64, h = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, cpr = hd / 8;
  const T* src = v + (int64_t)n * L * ldv + (int64_t)h * hd;
  for (int id = tid; id < 64 * cpr; id += 256) {
    const int r = id / cpr, c = (id % cpr) * 8;
    float x[8];
    if (s0 + r < L) load8(src + (int64_t)(s0 + r) * ldv + c, x);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) Cvt<T>::st(&tile[r][c + e], x[e]);
  }
  __syncthreads();
  T* dst = vt + ((int64_t)n * H + h) * (int64_t)hd * Lpad + s0;
  for (int id = tid; id < hd * 8; id += 256) {
    const int d = id >> 3, t0 = (id & 7) * 8;
    if (s0 + t0 >= Lpad) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = Cvt<T>::ld(&tile[t0 + e][d]);   // rows >= L are zero
    store8(dst + (int64_t)d * Lpad + t0, o);
  }
}

// ---------------------------------------------------------------- embedding / assembly
template <typename TT, typename TH>
__global__ __launch_bounds__(256) void embed_assemble_kernel(const int64_t* __restrict__ tokens, int64_t ld_tok,
                                                             const TT* __restrict__ table, TH* __restrict__ h, int T,
                                                             int W, int dim, int vocab) {
  const int S = T + W;
  const int row = blockIdx.x;           // b*S + s
  const int b = row / S, s = row % S;
  if (s >= 1 && s <= W) return;         // image words: written by the projector epilogue
  const int t = s == 0 ? 0 : s - W;
  int64_t tok = tokens[(int64_t)b * ld_tok + t];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  const TT* src = table + tok * dim;
  TH* dst = h + (int64_t)row * dim;
  for (int c = threadIdx.x * 8; c < dim; c += 256 * 8) {
    float v[8];
    load8(src + c, v);
    store8(dst + c, v);
  }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void fill_rows_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t ldd,
                                                        const int32_t* __restrict__ row_idx, int dim) {
  TD* d = dst + (int64_t)row_idx[blockIdx.x] * ldd;
  for (int c = threadIdx.x * 8; c < dim; c += 256 * 8) {
    float v[8];
    load8(src + c, v);
    store8(d + c, v);
  }
}

// ---------------------------------------------------------------- patch embed helpers
template <typename TI, typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const TI* __restrict__ img, T* __restrict__ cols, int Hi, int Wi,
                                                     int P, int Kpad) {
  const int g = Wi / P, gh = Hi / P;
  const int row = blockIdx.x;                 // n*gh*g + gy*g + gx
  const int n = row / (gh * g), gy = (row / g) % gh, gx = row % g;
  const int K = 3 * P * P;
  T* out = cols + (int64_t)row * Kpad;
  for (int k = threadIdx.x; k < Kpad; k += 256) {
    float v = 0.f;
    if (k < K) {
      const int c = k / (P * P), py = (k / P) % P, px = k % P;
      v = Cvt<TI>::ld(img + (((int64_t)n * 3 + c) * Hi + gy * P + py) * Wi + gx * P + px);
    }
    Cvt<T>::st(out + k, v);
  }
}

// x[n,0,:] = cls + pos[0];  x[n,1+t,:] = patch[n*T+t,:] + pos[1+t]   (bf16: cls.to(dtype)
// + zeros, then (x + pos) rounded once per element -- llama_ens5.py:358-362)
template <typename T>
__global__ __launch_bounds__(256) void vit_embed_kernel(const T* __restrict__ patch, const T* __restrict__ cls,
                                                        const T* __restrict__ pos, T* __restrict__ x, int Ttok, int width) {
  const int row = blockIdx.x;                 // n*(T+1) + t
  const int n = row / (Ttok + 1), t = row % (Ttok + 1);
  const T* src = t == 0 ? cls : patch + ((int64_t)n * Ttok + (t - 1)) * width;
  const T* pr = pos + (int64_t)t * width;
  T* dst = x + (int64_t)row * width;
  for (int c = threadIdx.x * 8; c < width; c += 256 * 8) {
    float a[8], p[8], o[8];
    load8(src + c, a);
    load8(pr + c, p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] + p[e];
    store8(dst + c, o);
  }
}

// LLM/llama_ens5.py:383-385: out[0:B] = bicubic(fp16) 2x downsample, out[(1+i)B:(2+i)B] = quadrant i.
// align_corners=False, A=-0.75, scale 2 -> source x = 2*dst + 0.5: taps {-1,0,1,2} with
// weights {-3/32, 19/32, 19/32, -3/32}, indices clamped at the border.
template <typename TI, typename T>
__global__ __launch_bounds__(256) void split_views_kernel(const TI* __restrict__ img, T* __restrict__ out, int B, int c) {
  const int64_t total = (int64_t)5 * B * 3 * c * c;
  const int S2 = 2 * c;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = i % c, y = (i / c) % c, ch = (i / ((int64_t)c * c)) % 3;
    const int nb = i / ((int64_t)3 * c * c);
    const int v = nb / B, b = nb % B;
    const TI* src = img + ((int64_t)b * 3 + ch) * S2 * S2;
    float r;
    if (v == 0) {
      const float wt[4] = {-0.09375f, 0.59375f, 0.59375f, -0.09375f};
      float acc = 0.f;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        const int yy = min(max(2 * y - 1 + dy, 0), S2 - 1);
        float rowacc = 0.f;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
          const int xx = min(max(2 * x - 1 + dx, 0), S2 - 1);
          const float px = (float)(_Float16)Cvt<TI>::ld(src + (int64_t)yy * S2 + xx);   // image.half()
          rowacc += wt[dx] * px;
        }
        acc += wt[dy] * rowacc;
      }
      r = (float)(_Float16)acc;
    } else {
      const int oy = (v - 1) / 2 * c, ox = (v - 1) % 2 * c;
      r = Cvt<TI>::ld(src + (int64_t)(oy + y) * S2 + ox + x);
    }
    Cvt<T>::st(out + i, r);
  }
}

// ---------------------------------------------------------------- argmax / CE
// one 1024-thread block per row; 4 independent 16-B loads in flight per thread (the decode loop calls this every step:
// a 256-thread scalar loop took 39 us on a [8, 32000] fp32 row set, latency-bound)
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int64_t ld, int64_t* __restrict__ out, int V) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const float* r = logits + (int64_t)blockIdx.x * ld;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  auto take = [&](float v, int i) { if (v > best || (v == best && i < idx)) { best = v; idx = i; } };
  const bool vec = ((reinterpret_cast<uintptr_t>(r) & 15) == 0);
  const int V4 = vec ? V / 4 : 0;
  for (int i0 = threadIdx.x; i0 < V4; i0 += 4 * 1024) {
    f32x4 x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * 1024;
      x[q] = i < V4 ? reinterpret_cast<const f32x4*>(r)[i] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) take(x[q][e], (i0 + q * 1024) * 4 + e);
  }
  for (int i = V4 * 4 + threadIdx.x; i < V; i += 1024) take(r[i], i);
  // NaN handling follows torch only for finite rows (the path never produces NaN logits)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;
  }
}

// ---------------------------------------------------------------- nucleus (top-p) sampling on the device
// model/meta.py:456-459 + sample_top_p (:568-583) for one batch row per 1024-thread block, WITHOUT the full-vocabulary sort:
//   probs = softmax(logits / temperature)                     e_i = exp(l_i / T - max_j l_j / T),  Z = sum e_i
//   sort descending, cumsum, drop where cumsum - p > top_p    token kept  <=>  mass of the tokens ranked before it <= top_p
//   renormalise, draw one                                     inverse CDF over the kept tokens in sorted order at u * M,
//                                                             u = the caller's uniform [0, 1) number for this row
// Both questions ("which token is the last one kept", "which token does u * M fall on") are the same selection: find the element
// whose predecessors in (value descending, index ascending) order weigh <= target < predecessors + itself.  It is answered by a
// radix descent over the 32 bits of e_i (non-negative floats order like their bit patterns): four passes, each a 256-bin histogram
// of MASS restricted to the prefix chosen so far, then a scan from the heaviest bin down.  Ties (equal e_i; common in the bf16-
// rounded tail, rare at the boundary) are ranked by token index.  Histograms are per wave and merged in wave order, so the sums
// -- and the chosen token -- are reproducible run to run.
// Round 5 (the eval recipe samples EVERY token: 203 us per step = 5 % of a decode step at bs 8): (i) e_i is computed ONCE and kept in LDS
// (V <= 32768: 128 KiB beside the 30-KiB scratch; larger vocabularies recompute it per pass as before); (ii) the radix digits are taken
// from key - min_key shifted so that the FIRST digit already spans the data's range: sharpened softmax values share their exponent,
// so the plain bit pattern's top byte put every element into one or two bins -- 64 lanes of a wave on one LDS atomic, serialised,
// 25 us per pass -- and spent a whole level on constant bits; now at most ceil(bits(max_key - min_key) / 8) levels run, each on
// spread-out bins.
struct TopPSel { unsigned key; float above; int rank; int id; };

template <bool CACHE>
__device__ __forceinline__ float topp_e(const float* __restrict__ r, const float* __restrict__ ec, int i, float T, float xm) {
  if constexpr (CACHE) return ec[i];
  else return expf(r[i] / T - xm);
}

// shared scratch of the selection (one block = one row)
struct TopPShared {
  float hist[16][256];        // per-wave mass histograms of the current digit
  float bin[256];
  float bin_top[256];         // the first digit's merged histogram: the same for both selections of a row (taken once)
  unsigned long long ball[64][16];   // tie ballots per (iteration, wave) of the last pass
  int wcnt[64][16];
  float red[16];
  unsigned ured[2][16];
  unsigned sel_key; float sel_above; int sel_bin; int found;
  int tie_n;
};

// target in units of Z.  want_id: also locate the token (selection 2); else only (key, above, rank) are needed (selection 1).
// kmin / levels / lsh: keys are ranked as (bits(e) - kmin) << lsh -- the left shift puts the range's top bit on the top bit of the
// first of `levels` 8-bit digits (0 levels: every element has the same value), so the first histogram already spreads over 128+ bins.
template <bool CACHE>
__device__ void topp_select(const float* __restrict__ r, const float* __restrict__ ec, int V, float T, float xm, unsigned kmin, int levels, int lsh,
                            float target, bool want_id, bool reuse_top, TopPShared& sh, TopPSel& out) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned prefix = 0;
  float above = 0.f;
  bool all_kept = false;
  for (int level = levels - 1; level >= 0; --level) {
    const int shift = level * 8;
    if (level == levels - 1 && reuse_top) {               // selection 2: the first digit's histogram of selection 1
      if (tid < 256) sh.bin[tid] = sh.bin_top[tid];
      if (tid == 0) sh.found = -1;
      __syncthreads();
    } else {
    for (int b = tid; b < 16 * 256; b += 1024) (&sh.hist[0][0])[b] = 0.f;
    __syncthreads();
    // four elements per trip: the reads (LDS or expf) of a trip are in flight together; the adds keep the element order of the plain loop
    for (int i0 = tid; i0 < V; i0 += 4096) {
      float e4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q * 1024;
        e4[q] = i < V ? topp_e<CACHE>(r, ec, i, T, xm) : -1.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned k = (__float_as_uint(e4[q]) - kmin) << lsh;
        if (e4[q] >= 0.f && (level == levels - 1 || (k >> (shift + 8)) == (prefix >> (shift + 8)))) atomicAdd(&sh.hist[wave][(k >> shift) & 255], e4[q]);
      }
    }
    __syncthreads();
    if (tid < 256) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) m += sh.hist[w][tid];
      sh.bin[tid] = m;
      if (level == levels - 1) sh.bin_top[tid] = m;
    }
    if (tid == 0) sh.found = -1;
    __syncthreads();
    }
    if (wave == 0) {
      // lane l owns bins 4 l .. 4 l + 3; suffix sums from bin 255 down
      float b4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b4[q] = sh.bin[lane * 4 + q];
      const float own = (b4[0] + b4[1]) + (b4[2] + b4[3]);
      float suf = own;                                   // inclusive suffix over lanes >= l
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float v = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += v;
      }
      const float nxt = __shfl_down(suf, 1, 64);
      float excl = above + (lane < 63 ? nxt : 0.f);      // mass of everything ranked before bin 4 l + 3
      int hi = -1, lo = 0x7fffffff;                      // highest bin whose inclusive mass passes the target / lowest non-empty bin
      float hi_excl = 0.f, lo_excl = 0.f;
#pragma unroll
      for (int q = 3; q >= 0; --q) {
        const float incl = excl + b4[q];
        if (b4[q] > 0.f) {
          if (hi < 0 && incl > target) { hi = lane * 4 + q; hi_excl = excl; }
          lo = lane * 4 + q; lo_excl = excl;
        }
        excl = incl;
      }
      int best_hi = hi, best_lo = lo;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        best_hi = max(best_hi, __shfl_xor(best_hi, o, 64));
        best_lo = min(best_lo, __shfl_xor(best_lo, o, 64));
      }
      // no bin passes: at the top level the target is beyond the total mass (cut: nothing is cut; draw: clamp onto the last
      // token); below it, the sub-bins' sum fell an ulp short of their parent's -- take the last non-empty one
      const bool clamp_lo = best_hi < 0 && (level < levels - 1 || want_id) && best_lo != 0x7fffffff;
      if (best_hi >= 0 && best_hi == hi) { sh.found = hi; sh.sel_above = hi_excl; }
      if (clamp_lo && best_lo == lo) { sh.found = lo; sh.sel_above = lo_excl; }
    }
    __syncthreads();
    if (sh.found < 0) { all_kept = true; break; }        // target >= total mass: nothing is cut (top_p >= 1)
    prefix |= (unsigned)sh.found << shift;
    above = sh.sel_above;
    __syncthreads();
  }
  if (all_kept) { out.key = 0u; out.above = -1.f; out.rank = 0; out.id = -1; return; }
  const unsigned key = (prefix >> lsh) + kmin;           // (levels == 0: every element equals the smallest key)
  const float ek = __uint_as_float(key);
  // ties on the selected value: rank inside them by token index
  int rank = ek > 0.f ? (int)floorf((target - above) / ek) : 0;
  if (rank < 0) rank = 0;
  out.key = key; out.above = above; out.rank = rank; out.id = -1;
  const int iters = (V + 1023) / 1024;
  for (int k = 0; k < iters; ++k) {
    const int i = k * 1024 + tid;
    const bool tie = i < V && __float_as_uint(topp_e<CACHE>(r, ec, i, T, xm)) == key;
    const unsigned long long bl = __ballot(tie);
    if (lane == 0) { sh.ball[k][wave] = bl; sh.wcnt[k][wave] = __popcll(bl); }
  }
  __syncthreads();
  // (k, w) slots in token order = flat index t = k * 16 + w; thread t holds slot t's count: total and the slot the rank falls into by a
  // block-wide scan over the <= 1024 slots (was: one thread walking 512 LDS words twice, ~10 us per selection)
  {
    const int c = tid < iters * 16 ? (&sh.wcnt[0][0])[tid] : 0;
    int incl = c;                                        // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) sh.ured[0][wave] = (unsigned)incl;
    __syncthreads();
    int base = 0, n = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int t = (int)sh.ured[0][w];
      if (w < wave) base += t;
      n += t;
    }
    if (rank >= n) rank = n - 1;                         // float division landed past the last tied element
    if (tid == 0) { sh.tie_n = rank; sh.sel_bin = -1; }
    __syncthreads();
    const int before = base + incl - c;                  // ties in earlier slots
    if (want_id && c > 0 && rank >= before && rank < before + c) {
      unsigned long long bl = (&sh.ball[0][0])[tid];
      for (int q = 0; q < rank - before; ++q) bl &= bl - 1;   // drop the lowest set bits
      sh.sel_bin = (tid >> 4) * 1024 + (tid & 15) * 64 + __ffsll((long long)bl) - 1;
    }
  }
  __syncthreads();
  out.rank = sh.tie_n;
  out.id = sh.sel_bin;
  __syncthreads();
}

template <bool CACHE>
__global__ __launch_bounds__(1024) void sample_top_p_kernel(const float* __restrict__ logits, int64_t ld, int V, float T, float top_p,
                                                            const float* __restrict__ u, int64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char topp_lds[];
  TopPShared& sh = *reinterpret_cast<TopPShared*>(topp_lds);
  float* ec = reinterpret_cast<float*>(topp_lds + ((sizeof(TopPShared) + 15) & ~(size_t)15));     // [V] when CACHE
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* r = logits + (int64_t)blockIdx.x * ld;
  // x_max and Z, fixed summation order (per thread, then lanes, then waves)
  float xm = -INFINITY;
  for (int i = tid; i < V; i += 1024) xm = fmaxf(xm, r[i] / T);
  xm = wave_max(xm);
  if (lane == 0) sh.red[wave] = xm;
  __syncthreads();
  xm = sh.red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) xm = fmaxf(xm, sh.red[w]);
  __syncthreads();
  float z = 0.f;
  unsigned kmin = 0xffffffffu, kmax = 0u;
  for (int i = tid; i < V; i += 1024) {
    const float e = expf(r[i] / T - xm);
    if constexpr (CACHE) ec[i] = e;
    z += e;
    const unsigned k = __float_as_uint(e);
    kmin = min(kmin, k);
    kmax = max(kmax, k);
  }
  z = wave_sum(z);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o, 64));
    kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
  }
  if (lane == 0) { sh.red[wave] = z; sh.ured[0][wave] = kmin; sh.ured[1][wave] = kmax; }
  __syncthreads();
  z = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    z += sh.red[w];
    kmin = min(kmin, sh.ured[0][w]);
    kmax = max(kmax, sh.ured[1][w]);
  }
  __syncthreads();
  const unsigned range = kmax - kmin;
  const int nbits = range == 0 ? 0 : 32 - __clz((int)range);
  const int levels = (nbits + 7) / 8, lsh = levels * 8 - nbits;
  // 1. the nucleus: the last kept token is the one the cumulative mass passes top_p on (meta.py:571-573)
  TopPSel cut, pick;
  topp_select<CACHE>(r, ec, V, T, xm, kmin, levels, lsh, top_p * z, false, false, sh, cut);
  const float M = cut.above < 0.f ? z : cut.above + (float)(cut.rank + 1) * __uint_as_float(cut.key);
  // 2. one draw from the renormalised nucleus (meta.py:574-577): inverse CDF at u * M
  float uu = u[blockIdx.x];
  uu = uu < 0.f ? 0.f : (uu >= 1.f ? 0.99999994f : uu);
  topp_select<CACHE>(r, ec, V, T, xm, kmin, levels, lsh, uu * M, true, true, sh, pick);
  if (tid == 0) {
    out[blockIdx.x] = pick.id < 0 ? 0 : pick.id;        // (id < 0 only for a row without any finite logit)
  }
}

// ---------------------------------------------------------------- one step of MetaModel.generate's token bookkeeping
// model/meta.py:456-477 for one batch row per block, after the row's argmax (temperature 0, :460) or with an externally
// sampled id (top-p branch, :457-459):
//   next = text_mask[row, cur] ? tokens[row, cur] : next          (prompts longer than the shortest one are teacher-forced, :463-465)
//   tokens[row, cur] = next
//   stop_pos[row] = stopped[row] ? stop_pos[row] : cur + 1
//   for every stop sequence st (in order), n = len(st), if cur + 1 - n >= 0:
//       hit = tokens[row, cur+1-n : cur+1] == st  &&  !text_mask[row, cur]  &&  !stopped[row]
//       if hit: stop_pos[row] = cur + 1 - n; stopped[row] = 1
// `live` counts the rows still running (the host polls that ONE word every few steps instead of `stopped.all()` every step).
__global__ __launch_bounds__(1024) void generate_step_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ sampled,
                                                             int V, int64_t* __restrict__ tokens, int64_t ld_tok,
                                                             const uint8_t* __restrict__ text_mask, int64_t ld_mask, int cur,
                                                             const int64_t* __restrict__ stop_seq, const int32_t* __restrict__ stop_off,
                                                             int n_stop, uint8_t* __restrict__ stopped, int64_t* __restrict__ stop_pos,
                                                             int32_t* __restrict__ live) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int row = blockIdx.x;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  if (!sampled) {
    const float* r = logits + (int64_t)row * ld;
    auto take = [&](float v, int i) { if (v > best || (v == best && i < idx)) { best = v; idx = i; } };
    const bool vec = ((reinterpret_cast<uintptr_t>(r) & 15) == 0);
    const int V4 = vec ? V / 4 : 0;
    for (int i0 = threadIdx.x; i0 < V4; i0 += 4 * 1024) {
      f32x4 x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q * 1024;
        x[q] = i < V4 ? reinterpret_cast<const f32x4*>(r)[i] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) take(x[q][e], (i0 + q * 1024) * 4 + e);
    }
    for (int i = V4 * 4 + threadIdx.x; i < V; i += 1024) take(r[i], i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  int64_t next;
  if (sampled) {
    next = sampled[row];
  } else {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    next = idx == 0x7fffffff ? 0 : idx;
  }
  int64_t* trow = tokens + (int64_t)row * ld_tok;
  const bool forced = text_mask[(int64_t)row * ld_mask + cur] != 0;
  if (forced) next = trow[cur];
  trow[cur] = next;
  bool st = stopped[row] != 0;
  int64_t sp = st ? stop_pos[row] : (int64_t)cur + 1;
  for (int s = 0; s < n_stop; ++s) {
    const int o = stop_off[s], n = stop_off[s + 1] - o;
    if (cur + 1 - n < 0 || forced || st) continue;
    bool hit = true;
    for (int k = 0; k < n && hit; ++k) hit = trow[cur + 1 - n + k] == stop_seq[o + k];
    if (hit) { sp = cur + 1 - n; st = true; }
  }
  if (st && !stopped[row] && live) atomicSub(live, 1);
  stopped[row] = st ? 1 : 0;
  stop_pos[row] = sp;
}

__global__ __launch_bounds__(256) void count_valid_kernel(const int64_t* __restrict__ labels, int rows, int32_t* __restrict__ out) {
  __shared__ float red[4];
  float c = 0.f;
  for (int i = threadIdx.x; i < rows; i += 256) c += labels[i] != 0 ? 1.f : 0.f;
  c = block_sum<256>(c, red);
  if (threadIdx.x == 0) *out = (int32_t)(c + 0.5f);
}

// row_loss[r] = logsumexp(logits[r]) - logits[r][label] (0 if label == 0: ignore_index)
// dlogits[r][v] = (softmax - onehot) * grad_scale / n_valid
template <typename T>
__global__ __launch_bounds__(256) void cross_entropy_kernel(const T* __restrict__ logits, int64_t ld,
                                                            const int64_t* __restrict__ labels, float* __restrict__ row_loss,
                                                            T* __restrict__ dlogits, int64_t ldd,
                                                            const int32_t* __restrict__ n_valid, float grad_scale, int V) {
  __shared__ float red[4];
  __shared__ float bmax[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const T* lr = logits + (int64_t)row * ld;
  const int64_t lab = labels[row];
  // 0 = pad = CrossEntropyLoss(ignore_index=0) (model/meta.py:67).  A label outside [0, V) would index past the logits row
  // (torch raises on it): it is treated like an ignored position -- zero loss, zero gradient -- never read.
  if (lab <= 0 || lab >= V) {
    if (tid == 0) row_loss[row] = 0.f;
    if (dlogits)
      for (int i = tid; i < V; i += 256) Cvt<T>::st(dlogits + (int64_t)row * ldd + i, 0.f);
    return;
  }
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 256) mx = fmaxf(mx, Cvt<T>::ld(lr + i));
  mx = wave_max(mx);
  if ((tid & 63) == 0) bmax[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(bmax[0], bmax[1]), fmaxf(bmax[2], bmax[3]));
  float s = 0.f;
  for (int i = tid; i < V; i += 256) s += __expf(Cvt<T>::ld(lr + i) - mx);
  s = block_sum<256>(s, red);
  const float lse = mx + __logf(s);
  if (tid == 0) row_loss[row] = lse - Cvt<T>::ld(lr + lab);
  if (dlogits) {
    const float g = grad_scale / (float)(*n_valid);
    for (int i = tid; i < V; i += 256) {
      float pr = __expf(Cvt<T>::ld(lr + i) - lse);
      if (i == lab) pr -= 1.f;
      Cvt<T>::st(dlogits + (int64_t)row * ldd + i, pr * g);
    }
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int a3v_rmsnorm(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int rows, int dim,
                           float eps, int x_dtype, int w_dtype, int y_dtype, void* stream) {
  if (!x || !w || !y || rows <= 0) return A3V_ERR_ARG;
  if (dim % 8 || dim > 8192 || ldx % 8 || ldy % 8) return A3V_ERR_SHAPE;
  const int key = x_dtype * 4 + w_dtype * 2 + y_dtype;
  dim3 g(rows), b(256);
  switch (key) {
    case 0: hipLaunchKernelGGL((rmsnorm_kernel<bf16_t, bf16_t, bf16_t>), g, b, 0, ST, (const bf16_t*)x, ldx, (const bf16_t*)w, (bf16_t*)y, ldy, dim, eps); break;
    case 7: hipLaunchKernelGGL((rmsnorm_kernel<float, float, float>), g, b, 0, ST, (const float*)x, ldx, (const float*)w, (float*)y, ldy, dim, eps); break;
    case 6: hipLaunchKernelGGL((rmsnorm_kernel<float, float, bf16_t>), g, b, 0, ST, (const float*)x, ldx, (const float*)w, (bf16_t*)y, ldy, dim, eps); break;
    case 4: hipLaunchKernelGGL((rmsnorm_kernel<float, bf16_t, bf16_t>), g, b, 0, ST, (const float*)x, ldx, (const bf16_t*)w, (bf16_t*)y, ldy, dim, eps); break;
    case 2: hipLaunchKernelGGL((rmsnorm_kernel<bf16_t, float, bf16_t>), g, b, 0, ST, (const bf16_t*)x, ldx, (const float*)w, (bf16_t*)y, ldy, dim, eps); break;
    default: return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_layernorm(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy,
                             const int32_t* row_map, int rows, int dim, float eps, int x_dtype, int p_dtype, int y_dtype,
                             void* stream) {
  if (!x || !w || !b || !y || rows <= 0) return A3V_ERR_ARG;
  if (dim % 8 || dim > 8192 || ldx % 8 || ldy % 8) return A3V_ERR_SHAPE;
  dim3 g(rows), t(256);
  switch (x_dtype * 4 + p_dtype * 2 + y_dtype) {
    case 0: hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t, bf16_t>), g, t, 0, ST, (const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, ldy, row_map, dim, eps); break;
    case 7: hipLaunchKernelGGL((layernorm_kernel<float, float, float>), g, t, 0, ST, (const float*)x, ldx, (const float*)w, (const float*)b, (float*)y, ldy, row_map, dim, eps); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<bf16_t, float, float>), g, t, 0, ST, (const bf16_t*)x, ldx, (const float*)w, (const float*)b, (float*)y, ldy, row_map, dim, eps); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<bf16_t, float, bf16_t>), g, t, 0, ST, (const bf16_t*)x, ldx, (const float*)w, (const float*)b, (bf16_t*)y, ldy, row_map, dim, eps); break;   // fp32 masters, bf16 residual stream
    default: return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_rope_kvcache(const void* qkv, int64_t ldqkv, void* q_out, int64_t ldq, void* k_cache, void* vt_cache,
                                const float* cos_sin, int B, int S, int H, int Hkv, int hd, int Smax,
                                int start_pos, int rope_pos0, int dtype, void* stream) {
  if (!qkv || !q_out || !k_cache || !vt_cache || !cos_sin || B <= 0 || S <= 0) return A3V_ERR_ARG;
  if (hd % 8 || hd > 128 || ldqkv % 8 || ldq % 8 || start_pos + S > Smax) return A3V_ERR_SHAPE;
  if (S <= 4) {
    dim3 gs(H + 2 * Hkv, S, B);
    if (dtype == A3V_BF16)
      hipLaunchKernelGGL(rope_kv_small_kernel<bf16_t>, gs, dim3(64), 0, ST, (const bf16_t*)qkv, ldqkv, (bf16_t*)q_out, ldq, (bf16_t*)k_cache, (bf16_t*)vt_cache, cos_sin, S, H, Hkv, hd, Smax, start_pos, rope_pos0);
    else if (dtype == A3V_F32)
      hipLaunchKernelGGL(rope_kv_small_kernel<float>, gs, dim3(64), 0, ST, (const float*)qkv, ldqkv, (float*)q_out, ldq, (float*)k_cache, (float*)vt_cache, cos_sin, S, H, Hkv, hd, Smax, start_pos, rope_pos0);
    else return A3V_ERR_DTYPE;
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  dim3 g((S + 63) / 64, H + 2 * Hkv, B);
  if (dtype == A3V_BF16)
    hipLaunchKernelGGL(rope_kv_kernel<bf16_t>, g, dim3(256), 0, ST, (const bf16_t*)qkv, ldqkv, (bf16_t*)q_out, ldq, (bf16_t*)k_cache, (bf16_t*)vt_cache, cos_sin, S, H, Hkv, hd, Smax, start_pos, rope_pos0);
  else if (dtype == A3V_F32)
    hipLaunchKernelGGL(rope_kv_kernel<float>, g, dim3(256), 0, ST, (const float*)qkv, ldqkv, (float*)q_out, ldq, (float*)k_cache, (float*)vt_cache, cos_sin, S, H, Hkv, hd, Smax, start_pos, rope_pos0);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_vt_pack(const void* v, int64_t ldv, void* vt, int N, int L, int H, int hd, int Lpad, int dtype, void* stream) {
  if (!v || !vt || N <= 0 || L <= 0) return A3V_ERR_ARG;
  if (hd % 8 || hd > 128 || ldv % 8 || Lpad % 8 || Lpad < L) return A3V_ERR_SHAPE;
  dim3 g((Lpad + 63) / 64, H, N);
  if (dtype == A3V_BF16)
    hipLaunchKernelGGL(vt_pack_kernel<bf16_t>, g, dim3(256), 0, ST, (const bf16_t*)v, ldv, (bf16_t*)vt, L, H, hd, Lpad);
  else if (dtype == A3V_F32)
    hipLaunchKernelGGL(vt_pack_kernel<float>, g, dim3(256), 0, ST, (const float*)v, ldv, (float*)vt, L, H, hd, Lpad);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_embed_assemble(const int64_t* tokens, int64_t ld_tok, const void* table, void* h, int B, int T, int W,
                                  int dim, int vocab, int table_dtype, int h_dtype, void* stream) {
  if (!tokens || !table || !h || B <= 0 || T <= 0 || W < 0) return A3V_ERR_ARG;
  if (dim % 8) return A3V_ERR_SHAPE;
  dim3 g(B * (T + W)), b(256);
  const int key = table_dtype * 2 + h_dtype;
  switch (key) {
    case 0: hipLaunchKernelGGL((embed_assemble_kernel<bf16_t, bf16_t>), g, b, 0, ST, tokens, ld_tok, (const bf16_t*)table, (bf16_t*)h, T, W, dim, vocab); break;
    case 3: hipLaunchKernelGGL((embed_assemble_kernel<float, float>), g, b, 0, ST, tokens, ld_tok, (const float*)table, (float*)h, T, W, dim, vocab); break;
    case 2: hipLaunchKernelGGL((embed_assemble_kernel<float, bf16_t>), g, b, 0, ST, tokens, ld_tok, (const float*)table, (bf16_t*)h, T, W, dim, vocab); break;
    case 1: hipLaunchKernelGGL((embed_assemble_kernel<bf16_t, float>), g, b, 0, ST, tokens, ld_tok, (const bf16_t*)table, (float*)h, T, W, dim, vocab); break;
    default: return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_fill_rows(const void* src, void* dst, int64_t ldd, const int32_t* row_idx, int n_rows, int dim,
                             int src_dtype, int dst_dtype, void* stream) {
  if (!src || !dst || !row_idx || n_rows <= 0) return A3V_ERR_ARG;
  if (dim % 8 || ldd % 8) return A3V_ERR_SHAPE;
  dim3 g(n_rows), b(256);
  const int key = src_dtype * 2 + dst_dtype;
  switch (key) {
    case 0: hipLaunchKernelGGL((fill_rows_kernel<bf16_t, bf16_t>), g, b, 0, ST, (const bf16_t*)src, (bf16_t*)dst, ldd, row_idx, dim); break;
    case 3: hipLaunchKernelGGL((fill_rows_kernel<float, float>), g, b, 0, ST, (const float*)src, (float*)dst, ldd, row_idx, dim); break;
    case 2: hipLaunchKernelGGL((fill_rows_kernel<float, bf16_t>), g, b, 0, ST, (const float*)src, (bf16_t*)dst, ldd, row_idx, dim); break;
    case 1: hipLaunchKernelGGL((fill_rows_kernel<bf16_t, float>), g, b, 0, ST, (const bf16_t*)src, (float*)dst, ldd, row_idx, dim); break;
    default: return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_patch_im2col(const void* img, void* cols, int N, int Hi, int Wi, int P, int Kpad, int in_dtype,
                                int out_dtype, void* stream) {
  if (!img || !cols || N <= 0 || P <= 0) return A3V_ERR_ARG;
  if (Hi % P || Wi % P || Kpad < 3 * P * P) return A3V_ERR_SHAPE;
  dim3 g(N * (Hi / P) * (Wi / P)), b(256);
  switch (in_dtype * 2 + out_dtype) {
    case 0: hipLaunchKernelGGL((im2col_kernel<bf16_t, bf16_t>), g, b, 0, ST, (const bf16_t*)img, (bf16_t*)cols, Hi, Wi, P, Kpad); break;
    case 2: hipLaunchKernelGGL((im2col_kernel<float, bf16_t>), g, b, 0, ST, (const float*)img, (bf16_t*)cols, Hi, Wi, P, Kpad); break;
    case 3: hipLaunchKernelGGL((im2col_kernel<float, float>), g, b, 0, ST, (const float*)img, (float*)cols, Hi, Wi, P, Kpad); break;
    default: return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_vit_embed(const void* patch, const void* cls, const void* pos, void* x, int N, int T, int width, int dtype, void* stream) {
  if (!patch || !cls || !pos || !x || N <= 0 || T <= 0) return A3V_ERR_ARG;
  if (width % 8) return A3V_ERR_SHAPE;
  dim3 g(N * (T + 1));
  if (dtype == A3V_BF16) hipLaunchKernelGGL(vit_embed_kernel<bf16_t>, g, dim3(256), 0, ST, (const bf16_t*)patch, (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)x, T, width);
  else if (dtype == A3V_F32) hipLaunchKernelGGL(vit_embed_kernel<float>, g, dim3(256), 0, ST, (const float*)patch, (const float*)cls, (const float*)pos, (float*)x, T, width);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_split_views(const void* img, void* out, int B, int crop, int in_dtype, int out_dtype, void* stream) {
  if (!img || !out || B <= 0 || crop <= 0) return A3V_ERR_ARG;
  const int64_t total = (int64_t)5 * B * 3 * crop * crop;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  dim3 g(blocks), b(256);
  switch (in_dtype * 2 + out_dtype) {
    case 0: hipLaunchKernelGGL((split_views_kernel<bf16_t, bf16_t>), g, b, 0, ST, (const bf16_t*)img, (bf16_t*)out, B, crop); break;
    case 2: hipLaunchKernelGGL((split_views_kernel<float, bf16_t>), g, b, 0, ST, (const float*)img, (bf16_t*)out, B, crop); break;
    case 3: hipLaunchKernelGGL((split_views_kernel<float, float>), g, b, 0, ST, (const float*)img, (float*)out, B, crop); break;
    default: return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// ---------------------------------------------------------------- image preprocessing (data/transform.py:13-68 on the device)
// PadToSquare(fill) -> bicubic resize -> ToTensor -> Normalize, with Pillow's 8-bit resampling arithmetic (the reference transform
// is torchvision on PIL images = Pillow's ImagingResample): two separable passes, each output sample
//   clip8((2^21 + sum_k pixel[lo + k] * coef[k]) >> 22)      coef = round(weight * 2^22) (host-computed table, Resample.c)
// horizontal pass first (uint8 intermediate), then vertical; the padded square is never materialised (pixels outside the image
// rectangle read as the fill colour).  The vertical pass finishes with ((v / 255) - mean) / std in fp32 and writes CHW.
namespace {
__device__ __forceinline__ int clip8(int v) { v >>= 22; return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int H, int W, int side, int px, int py, int fill_r,
                                                         int fill_g, int fill_b, const int32_t* __restrict__ kx, const int32_t* __restrict__ bx,
                                                         int ksize, int out, uint8_t* __restrict__ tmp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= side * out) return;
  const int y = i / out, x = i - y * out;
  const int lo = bx[2 * x], n = bx[2 * x + 1];
  const int32_t* k = kx + (int64_t)x * ksize;
  int sr = 1 << 21, sg = 1 << 21, sb = 1 << 21;
  const int sy = y - py;
  const bool row_in = sy >= 0 && sy < H;
  for (int t = 0; t < n; ++t) {
    const int sx = lo + t - px;
    int r = fill_r, g = fill_g, b = fill_b;
    if (row_in && sx >= 0 && sx < W) {
      const uint8_t* q = src + ((int64_t)sy * W + sx) * 3;
      r = q[0]; g = q[1]; b = q[2];
    }
    sr += r * k[t]; sg += g * k[t]; sb += b * k[t];
  }
  uint8_t* o = tmp + (int64_t)i * 3;
  o[0] = (uint8_t)clip8(sr); o[1] = (uint8_t)clip8(sg); o[2] = (uint8_t)clip8(sb);
}

template <typename TO>
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int side, int out, const int32_t* __restrict__ ky,
                                                              const int32_t* __restrict__ by, int ksize, float m0, float m1, float m2,
                                                              float s0, float s1, float s2, TO* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= out * out) return;
  const int y = i / out, x = i - y * out;
  const int lo = by[2 * y], n = by[2 * y + 1];
  const int32_t* k = ky + (int64_t)y * ksize;
  int sr = 1 << 21, sg = 1 << 21, sb = 1 << 21;
  for (int t = 0; t < n; ++t) {
    const uint8_t* q = tmp + ((int64_t)(lo + t) * out + x) * 3;
    sr += q[0] * k[t]; sg += q[1] * k[t]; sb += q[2] * k[t];
  }
  const int64_t plane = (int64_t)out * out;
  // ToTensor: uint8 -> float / 255 (a correctly rounded fp32 division); Normalize: (x - mean) / std, both fp32 (torch ops)
  Cvt<TO>::st(dst + i, __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(sr), 255.f), m0), s0));
  Cvt<TO>::st(dst + plane + i, __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(sg), 255.f), m1), s1));
  Cvt<TO>::st(dst + 2 * plane + i, __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(sb), 255.f), m2), s2));
}
}  // namespace

// ---- the same two passes for a BATCH of decoded images in two launches per 16 images (blockIdx.y = image): the descriptors travel
// by value in the kernel arguments, so a loader hands over one list per batch and nothing is staged per image
namespace {
constexpr int PRE_CHUNK = 16;
struct PreBatch { a3v_image_desc d[PRE_CHUNK]; };

__global__ __launch_bounds__(256) void resample_h_batch_kernel(PreBatch pb, int fill_r, int fill_g, int fill_b, int out, uint8_t* __restrict__ tmp,
                                                               int64_t tmp_stride) {
  const a3v_image_desc& im = pb.d[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= im.side * out) return;
  const int y = i / out, x = i - y * out;
  const int lo = im.bounds[2 * x], n = im.bounds[2 * x + 1];
  const int32_t* k = im.coeffs + (int64_t)x * im.ksize;
  int sr = 1 << 21, sg = 1 << 21, sb = 1 << 21;
  const int sy = y - im.pad_y;
  const bool row_in = sy >= 0 && sy < im.H;
  for (int t = 0; t < n; ++t) {
    const int sx = lo + t - im.pad_x;
    int r = fill_r, g = fill_g, b = fill_b;
    if (row_in && sx >= 0 && sx < im.W) {
      const uint8_t* q = im.src + ((int64_t)sy * im.W + sx) * 3;
      r = q[0]; g = q[1]; b = q[2];
    }
    sr += r * k[t]; sg += g * k[t]; sb += b * k[t];
  }
  uint8_t* o = tmp + blockIdx.y * tmp_stride + (int64_t)i * 3;
  o[0] = (uint8_t)clip8(sr); o[1] = (uint8_t)clip8(sg); o[2] = (uint8_t)clip8(sb);
}

template <typename TO>
__global__ __launch_bounds__(256) void resample_v_norm_batch_kernel(PreBatch pb, const uint8_t* __restrict__ tmp, int64_t tmp_stride, int out,
                                                                    float m0, float m1, float m2, float s0, float s1, float s2,
                                                                    TO* __restrict__ dst, int64_t dst_stride) {
  const a3v_image_desc& im = pb.d[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= out * out) return;
  const int y = i / out, x = i - y * out;
  const int lo = im.bounds[2 * y], n = im.bounds[2 * y + 1];
  const int32_t* k = im.coeffs + (int64_t)y * im.ksize;
  const uint8_t* tm = tmp + blockIdx.y * tmp_stride;
  int sr = 1 << 21, sg = 1 << 21, sb = 1 << 21;
  for (int t = 0; t < n; ++t) {
    const uint8_t* q = tm + ((int64_t)(lo + t) * out + x) * 3;
    sr += q[0] * k[t]; sg += q[1] * k[t]; sb += q[2] * k[t];
  }
  const int64_t plane = (int64_t)out * out;
  TO* d = dst + blockIdx.y * dst_stride;
  Cvt<TO>::st(d + i, __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(sr), 255.f), m0), s0));
  Cvt<TO>::st(d + plane + i, __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(sg), 255.f), m1), s1));
  Cvt<TO>::st(d + 2 * plane + i, __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(sb), 255.f), m2), s2));
}
}  // namespace

extern "C" int a3v_preprocess_batch(const a3v_image_desc* images, int n, const int* fill_rgb, int out_size, uint8_t* tmp, int64_t tmp_stride,
                                    void* dst, int64_t dst_stride, int dst_dtype, const float* mean, const float* std_, void* stream) {
  if (!images || n <= 0 || !fill_rgb || !tmp || !dst || !mean || !std_ || out_size <= 0) return A3V_ERR_ARG;
  if (dst_dtype != A3V_F32 && dst_dtype != A3V_BF16) return A3V_ERR_DTYPE;
  if (dst_stride < (int64_t)3 * out_size * out_size) return A3V_ERR_SHAPE;
  for (int i0 = 0; i0 < n; i0 += PRE_CHUNK) {
    const int nc = n - i0 < PRE_CHUNK ? n - i0 : PRE_CHUNK;
    PreBatch pb{};
    int max_side = 0;
    for (int j = 0; j < nc; ++j) {
      const a3v_image_desc& im = images[i0 + j];
      if (!im.src || !im.coeffs || !im.bounds) return A3V_ERR_ARG;
      if (im.H <= 0 || im.W <= 0 || im.side < im.H || im.side < im.W || im.pad_x < 0 || im.pad_y < 0 || im.pad_x + im.W > im.side ||
          im.pad_y + im.H > im.side || im.ksize <= 0 || (int64_t)im.side * out_size * 3 > tmp_stride) return A3V_ERR_SHAPE;
      pb.d[j] = im;
      max_side = im.side > max_side ? im.side : max_side;
    }
    uint8_t* tm = tmp + (int64_t)i0 * tmp_stride;
    const int n1 = max_side * out_size, n2 = out_size * out_size;
    hipLaunchKernelGGL(resample_h_batch_kernel, dim3((n1 + 255) / 256, nc), dim3(256), 0, ST, pb, fill_rgb[0], fill_rgb[1], fill_rgb[2], out_size,
                       tm, tmp_stride);
    if (dst_dtype == A3V_F32)
      hipLaunchKernelGGL(resample_v_norm_batch_kernel<float>, dim3((n2 + 255) / 256, nc), dim3(256), 0, ST, pb, tm, tmp_stride, out_size, mean[0],
                         mean[1], mean[2], std_[0], std_[1], std_[2], (float*)dst + (int64_t)i0 * dst_stride, dst_stride);
    else
      hipLaunchKernelGGL(resample_v_norm_batch_kernel<bf16_t>, dim3((n2 + 255) / 256, nc), dim3(256), 0, ST, pb, tm, tmp_stride, out_size, mean[0],
                         mean[1], mean[2], std_[0], std_[1], std_[2], (bf16_t*)dst + (int64_t)i0 * dst_stride, dst_stride);
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_preprocess_image(const uint8_t* src, int H, int W, int side, int pad_x, int pad_y, const int* fill_rgb,
                                    const int32_t* kx, const int32_t* bx, int ksize_x, const int32_t* ky, const int32_t* by, int ksize_y,
                                    int out_size, uint8_t* tmp, void* dst, int dst_dtype, const float* mean, const float* std_, void* stream) {
  if (!src || !kx || !bx || !ky || !by || !tmp || !dst || !fill_rgb || !mean || !std_) return A3V_ERR_ARG;
  if (H <= 0 || W <= 0 || side < H || side < W || out_size <= 0 || pad_x < 0 || pad_y < 0 || pad_x + W > side || pad_y + H > side) return A3V_ERR_SHAPE;
  const int n1 = side * out_size, n2 = out_size * out_size;
  hipLaunchKernelGGL(resample_h_kernel, dim3((n1 + 255) / 256), dim3(256), 0, ST, src, H, W, side, pad_x, pad_y, fill_rgb[0], fill_rgb[1],
                     fill_rgb[2], kx, bx, ksize_x, out_size, tmp);
  if (dst_dtype == A3V_F32)
    hipLaunchKernelGGL(resample_v_norm_kernel<float>, dim3((n2 + 255) / 256), dim3(256), 0, ST, tmp, side, out_size, ky, by, ksize_y, mean[0],
                       mean[1], mean[2], std_[0], std_[1], std_[2], (float*)dst);
  else if (dst_dtype == A3V_BF16)
    hipLaunchKernelGGL(resample_v_norm_kernel<bf16_t>, dim3((n2 + 255) / 256), dim3(256), 0, ST, tmp, side, out_size, ky, by, ksize_y, mean[0],
                       mean[1], mean[2], std_[0], std_[1], std_[2], (bf16_t*)dst);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_generate_step(const float* logits, int64_t ld, const int64_t* sampled, int B, int V, int64_t* tokens, int64_t ld_tok,
                                 const uint8_t* text_mask, int64_t ld_mask, int cur_pos, const int64_t* stop_seq, const int32_t* stop_off,
                                 int n_stop, uint8_t* stopped, int64_t* stop_pos, int32_t* live, void* stream) {
  if ((!logits && !sampled) || !tokens || !text_mask || !stopped || !stop_pos || B <= 0 || V <= 0 || cur_pos < 0 || n_stop < 0) return A3V_ERR_ARG;
  if (n_stop > 0 && (!stop_seq || !stop_off)) return A3V_ERR_ARG;
  hipLaunchKernelGGL(generate_step_kernel, dim3(B), dim3(1024), 0, ST, logits, ld, sampled, V, tokens, ld_tok, text_mask, ld_mask, cur_pos,
                     stop_seq, stop_off, n_stop, stopped, stop_pos, live);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_sample_top_p(const float* logits, int64_t ld, int B, int V, float temperature, float top_p, const float* u,
                                int64_t* out, void* stream) {
  if (!logits || !u || !out || B <= 0 || V <= 0) return A3V_ERR_ARG;
  if (V > 65536 || !(temperature > 0.f) || !(top_p > 0.f)) return A3V_ERR_SHAPE;
  const size_t base = (sizeof(TopPShared) + 15) & ~(size_t)15;
  if (V <= 32768 && A3V_ENV_INT("A3V_SAMPLER_CACHE", 1)) {     // e_i kept in LDS (A3V_SAMPLER_CACHE=0: recomputed per pass, A/B and equality tests)
    static bool attr[A3V_MAX_DEV][1] = {};
    const int bytes = (int)(base + (size_t)V * 4);
    const int rc = a3v_dyn_lds_once(attr, 0, (const void*)sample_top_p_kernel<true>, 160 * 1024);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(sample_top_p_kernel<true>, dim3(B), dim3(1024), (size_t)bytes, ST, logits, ld, V, temperature, top_p, u, out);
  } else {
    hipLaunchKernelGGL(sample_top_p_kernel<false>, dim3(B), dim3(1024), base, ST, logits, ld, V, temperature, top_p, u, out);
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_argmax(const float* logits, int64_t ld, int64_t* out, int B, int V, void* stream) {
  if (!logits || !out || B <= 0 || V <= 0) return A3V_ERR_ARG;
  hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(1024), 0, ST, logits, ld, out, V);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_count_valid(const int64_t* labels, int rows, int32_t* n_valid_dev, void* stream) {
  if (!labels || !n_valid_dev || rows <= 0) return A3V_ERR_ARG;
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(256), 0, ST, labels, rows, n_valid_dev);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_cross_entropy(const void* logits, int64_t ld, const int64_t* labels, float* row_loss, void* dlogits,
                                 int64_t ldd, const int32_t* n_valid_dev, float grad_scale, int rows, int V, int dtype,
                                 void* stream) {
  if (!logits || !labels || !row_loss || rows <= 0 || V <= 0) return A3V_ERR_ARG;
  if (dlogits && !n_valid_dev) return A3V_ERR_ARG;
  if (dtype == A3V_BF16)
    hipLaunchKernelGGL(cross_entropy_kernel<bf16_t>, dim3(rows), dim3(256), 0, ST, (const bf16_t*)logits, ld, labels, row_loss, (bf16_t*)dlogits, ldd, n_valid_dev, grad_scale, V);
  else if (dtype == A3V_F32)
    hipLaunchKernelGGL(cross_entropy_kernel<float>, dim3(rows), dim3(256), 0, ST, (const float*)logits, ld, labels, row_loss, (float*)dlogits, ldd, n_valid_dev, grad_scale, V);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_quantize_rows_fp8(const void* x, int64_t ldx, const void* norm_w, float eps, void* q, int64_t ldq, float* scales,
                                     int rows, int dim, int x_dtype, void* stream) {
  if (!x || !q || !scales || rows <= 0 || dim <= 0) return A3V_ERR_ARG;
  if (dim % 8 || dim > 16384 || ldx % 8 || ldq % 8) return A3V_ERR_SHAPE;
  dim3 g(rows), b(256);
  if (x_dtype == A3V_BF16) {
    if (norm_w) hipLaunchKernelGGL((rows_quant_fp8_kernel<bf16_t, bf16_t, true>), g, b, 0, ST, (const bf16_t*)x, ldx, (const bf16_t*)norm_w, (uint8_t*)q, ldq, scales, dim, eps);
    else hipLaunchKernelGGL((rows_quant_fp8_kernel<bf16_t, bf16_t, false>), g, b, 0, ST, (const bf16_t*)x, ldx, (const bf16_t*)nullptr, (uint8_t*)q, ldq, scales, dim, eps);
  } else {
    return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

