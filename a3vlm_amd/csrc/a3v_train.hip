// Backward-path kernels (training): transposes for the dgrad/wgrad GEMMs, RMSNorm / LayerNorm /
// SwiGLU / RoPE backward, attention backward, embedding scatter, reductions.
// GEMM-shaped backward work reuses a3v_gemm_nt on transposed images:
//   dX[M,K] = dY[M,N] W[N,K]        = gemm_nt(dY, W^T[K,N])
//   dW[N,K] += dY^T[N,M] X[M,K]     = gemm_nt(dY^T[N,Mp], X^T[K,Mp])  (RES_F32 accumulate into the fp32 grad)
// Activations and their grads are in the activation dtype TA (bf16, or f32 on the parity path); the
// residual stream, master weights and weight grads are fp32 (reference: autocast(bf16) + fp32 masters,
// main_finetune.py:212-217, engine_finetune.py:44-68).
#include "a3v_common.h"

namespace {

// ------------------------------------------------------------------ transpose  src[R,C] -> dst[C,Rpad]
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, int64_t lds_, int64_t bs_s,
                                                        T* __restrict__ dst, int64_t ldd, int64_t bs_d, int R, int C, int Rpad) {
  __shared__ T tile[64][64 + 2];
  const T* s = src + blockIdx.z * bs_s;
  T* d = dst + blockIdx.z * bs_d;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? s[(int64_t)r * lds_ + c] : (T)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) d[(int64_t)c * ldd + r] = tile[tx][i];
  }
}

// bf16 fast path: 16-B global loads and stores, the 64 x 64 tile is transposed through LDS (row pitch 66 elements: the
// eight 2-B column reads of a lane and the lanes of a wave fall on distinct banks).  C % 8 == 0, Rpad % 8 == 0, 16-B aligned.
__global__ __launch_bounds__(256) void transpose_bf16_vec_kernel(const bf16_t* __restrict__ src, int64_t lds_, int64_t bs_s,
                                                                 bf16_t* __restrict__ dst, int64_t ldd, int64_t bs_d, int R, int C, int Rpad,
                                                                 int n_in = 0x7fffffff, int64_t bs_s2 = 0, int64_t bs_d2 = 0) {
  __shared__ uint32_t tile[64][33];                  // [row][column pair]: 66 bf16 per row
  // two-level batch: z = outer * n_in + inner (one launch for all (batch, head) matrices of a [B, S, H, hd] activation)
  const int zo = blockIdx.z / n_in, zi = blockIdx.z - zo * n_in;
  const bf16_t* s = src + zi * bs_s + zo * bs_s2;
  bf16_t* d = dst + zi * bs_d + zo * bs_d2;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = threadIdx.x + j * 256;
    const int row = idx >> 3, c8 = (idx & 7) * 8;
    const int r = r0 + row, c = c0 + c8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < R && c < C) v = *reinterpret_cast<const u32x4*>(s + (int64_t)r * lds_ + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[row][(c8 >> 1) + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = threadIdx.x + j * 256;
    const int c = idx >> 3, r8 = (idx & 7) * 8;
    if (c0 + c < C && r0 + r8 < Rpad) {
      const int sh = (c & 1) * 16;
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t lo = (tile[r8 + 2 * e][c >> 1] >> sh) & 0xffffu;
        const uint32_t hi = (tile[r8 + 2 * e + 1][c >> 1] >> sh) & 0xffffu;
        o[e] = lo | (hi << 16);
      }
      *reinterpret_cast<u32x4*>(d + (int64_t)(c0 + c) * ldd + r0 + r8) = o;
    }
  }
}

// ------------------------------------------------------------------ RMSNorm backward
// y = w * x * r, r = rsqrt(mean(x^2)+eps).  dx = r*dy*w - x * r^3 * sum(dy*w*x)/dim  (added into dh);
// dw[j] += sum_rows dy*x*r  (thread-private partial over the block's rows, one atomic per column per block)
template <typename TA, int RPB, typename TS = float>     // TS: dtype of the residual stream (x and the accumulated dh)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const TS* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                          const TA* __restrict__ dy, int64_t lddy, TS* __restrict__ dh,
                                                          int64_t lddh, float* __restrict__ dw, int rows, int dim, float eps) {
  __shared__ float red[8];
  const int tid = threadIdx.x;
  constexpr int MAXC = 32;                // dim <= 8192
  float dwp[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) dwp[i] = 0.f;
  const int row0 = blockIdx.x * RPB;
  for (int rr = 0; rr < RPB; ++rr) {
    const int row = row0 + rr;
    if (row >= rows) break;
    const TS* xr = x + (int64_t)row * ldx;
    const TA* dyr = dy + (int64_t)row * lddy;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * 256;
      if (c < dim) {
        const float xv = Cvt<TS>::ld(xr + c), g = Cvt<TA>::ld(dyr + c) * w[c];
        s1 = fmaf(xv, xv, s1);
        s2 = fmaf(g, xv, s2);
      }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = s1; red[4 + (tid >> 6)] = s2; }
    __syncthreads();
    s1 = red[0] + red[1] + red[2] + red[3];
    s2 = red[4] + red[5] + red[6] + red[7];
    const float r = rsqrtf(s1 / (float)dim + eps);
    const float k2 = r * r * r * s2 / (float)dim;
    TS* dhr = dh + (int64_t)row * lddh;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * 256;
      if (c < dim) {
        const float xv = Cvt<TS>::ld(xr + c), dyv = Cvt<TA>::ld(dyr + c);
        Cvt<TS>::st(dhr + c, Cvt<TS>::ld(dhr + c) + (r * dyv * w[c] - xv * k2));
        dwp[i] += dyv * xv * r;
      }
    }
  }
  if (dw) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * 256;
      if (c < dim) atomicAdd(dw + c, dwp[i]);
    }
  }
}


// Vector form (dim % 4 == 0, 16-B aligned rows): a thread keeps its float4 slices of x and dy in registers across the two
// passes of a row (one HBM read each), RPB rows per block for the weight-gradient partials.
template <typename TA, int RPB, int MAXV, bool PART, typename TS = float>     // TS: dtype of the residual stream (x, dh)
__global__ __launch_bounds__(256) void rmsnorm_bwd_vec_kernel(const TS* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                              const TA* __restrict__ dy, int64_t lddy, TS* __restrict__ dh,
                                                              int64_t lddh, float* __restrict__ dw, int rows, int dim, float eps,
                                                              bf16_t* __restrict__ dh_bf = nullptr, int64_t ld_bf = 0) {
  __shared__ float red[2][8];
  const int tid = threadIdx.x;
  const int nv = dim >> 2;
  f32x4 wv[MAXV], dwp[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = tid + i * 256;
    wv[i] = c < nv ? reinterpret_cast<const f32x4*>(w)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    dwp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int row0 = blockIdx.x * RPB;
  // bf16 stream: the NEXT row's x, dy and dh vectors (2 registers each) are requested before the current row's reductions and barrier,
  // so the row loop no longer pays a memory round trip per row behind its barrier (round 4, rows wider than 4096; A/B macro RMSB_NO_PREFETCH)
#ifdef RMSB_NO_PREFETCH
  constexpr bool PF = false;
#else
  constexpr bool PF = sizeof(TS) == 2 && sizeof(TA) == 2 && MAXV > 4;     // dim 5120 (13B): 115 -> 82 us; dim 4096: 53 -> 55 us, kept off (tools/rmsnorm_bwd_bf16_bench.py)
#endif
  bf16x4 nx[MAXV], ng[MAXV], nh[MAXV];
  auto fetch = [&](int row) {
    if constexpr (PF) {
      const int rc = row < rows ? row : rows - 1;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = tid + i * 256;
        const int cc = c < nv ? c : 0;
        nx[i] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(x) + (int64_t)rc * ldx + 4 * cc);
        ng[i] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(dy) + (int64_t)rc * lddy + 4 * cc);
        nh[i] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(dh) + (int64_t)rc * lddh + 4 * cc);
      }
    }
  };
  fetch(row0);
  for (int rr = 0; rr < RPB; ++rr) {
    const int row = row0 + rr;
    if (row >= rows) break;
    const TS* xr = x + (int64_t)row * ldx;
    const TA* dyr = dy + (int64_t)row * lddy;
    f32x4 xv[MAXV], gv[MAXV], hv[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = tid + i * 256;
      if (c < nv) {
        if constexpr (PF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { xv[i][e] = (float)nx[i][e]; gv[i][e] = (float)ng[i][e]; hv[i][e] = (float)nh[i][e]; }
        } else {
        if constexpr (sizeof(TS) == 2) {
          const bf16x4 xb = *reinterpret_cast<const bf16x4*>(xr + 4 * c);
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[i][e] = (float)xb[e];
        } else {
          xv[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        }
        if constexpr (sizeof(TA) == 2) {
          const bf16x4 d = *reinterpret_cast<const bf16x4*>(dyr + 4 * c);
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[i][e] = (float)d[e];
        } else {
          gv[i] = *reinterpret_cast<const f32x4*>(dyr + 4 * c);
        }
        }
      } else {
        xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        gv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1 = fmaf(xv[i][e], xv[i][e], s1);
        s2 = fmaf(gv[i][e] * wv[i][e], xv[i][e], s2);
      }
    }
    if (rr + 1 < RPB) fetch(row + 1);
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const int pb = rr & 1;                    // double-buffered: one barrier per row
    if ((tid & 63) == 0) { red[pb][tid >> 6] = s1; red[pb][4 + (tid >> 6)] = s2; }
    __syncthreads();
    s1 = red[pb][0] + red[pb][1] + red[pb][2] + red[pb][3];
    s2 = red[pb][4] + red[pb][5] + red[pb][6] + red[pb][7];
    const float r = rsqrtf(s1 / (float)dim + eps);
    const float k2 = r * r * r * s2 / (float)dim;
    TS* dhr = dh + (int64_t)row * lddh;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = tid + i * 256;
      if (c < nv) {
        f32x4 o;
        if constexpr (PF) {
          o = hv[i];
        } else if constexpr (sizeof(TS) == 2) {
          const bf16x4 ob0 = *reinterpret_cast<const bf16x4*>(dhr + 4 * c);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (float)ob0[e];
        } else {
          o = *reinterpret_cast<const f32x4*>(dhr + 4 * c);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] += r * gv[i][e] * wv[i][e] - xv[i][e] * k2;
          dwp[i][e] += gv[i][e] * xv[i][e] * r;
        }
        if constexpr (sizeof(TS) == 2) {       // bf16 stream: the accumulated gradient is rounded as autograd rounds it at the residual add
          bf16x4 ob1;
#pragma unroll
          for (int e = 0; e < 4; ++e) ob1[e] = f2bf(o[e]);
          *reinterpret_cast<bf16x4*>(dhr + 4 * c) = ob1;
        } else {
          *reinterpret_cast<f32x4*>(dhr + 4 * c) = o;
        }
        if (dh_bf) {            // the bf16 copy the next weight / input gradient GEMMs read (was a cast pass of its own)
          bf16x4 ob;
#pragma unroll
          for (int e = 0; e < 4; ++e) ob[e] = f2bf(o[e]);
          *reinterpret_cast<bf16x4*>(dh_bf + (int64_t)row * ld_bf + 4 * c) = ob;
        }
      }
    }
  }
  if (dw) {
    if (PART) {         // dw = per-block partial rows [gridDim.x][dim]: summed by colsum_add_kernel (1091 blocks x 4096 atomics on
                        // the same 16 KB cost 110 of this kernel's 195 us at 8728 x 4096)
      f32x4* out = reinterpret_cast<f32x4*>(dw + (int64_t)blockIdx.x * dim);
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = tid + i * 256;
        if (c < nv) out[c] = dwp[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = tid + i * 256;
        if (c < nv) {
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(dw + 4 * c + e, dwp[i][e]);
        }
      }
    }
  }
}

// out[c] += sum_r part[r][c]: grid (cols / 256, row chunks); one atomic per column per chunk
__global__ __launch_bounds__(256) void colsum_add_kernel(const float* __restrict__ part, int nrows, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int per = (nrows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(nrows, r0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 4 <= r1; r += 4) {
    a0 += part[(int64_t)r * cols + c];
    a1 += part[(int64_t)(r + 1) * cols + c];
    a2 += part[(int64_t)(r + 2) * cols + c];
    a3 += part[(int64_t)(r + 3) * cols + c];
  }
  for (; r < r1; ++r) a0 += part[(int64_t)r * cols + c];
  if (r1 > r0) atomicAdd(out + c, (a0 + a1) + (a2 + a3));
}

// ------------------------------------------------------------------ LayerNorm backward (projector LN)
// y = (x-mean)*rstd*w + b (fp32 math); dy rows are gathered through row_map from the fp32 stream.
template <typename TA, int RPB, typename TD = float>     // TD: dtype of the stream gradient rows dy is gathered from
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TA* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                            const TD* __restrict__ dy, int64_t lddy, const int32_t* __restrict__ row_map,
                                                            TA* __restrict__ dx, int64_t lddx, float* __restrict__ dw, float* __restrict__ db,
                                                            int rows, int dim, float eps) {
  __shared__ float red[12];
  const int tid = threadIdx.x;
  constexpr int MAXC = 32;
  float dwp[MAXC], dbp[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) { dwp[i] = 0.f; dbp[i] = 0.f; }
  const int row0 = blockIdx.x * RPB;
  for (int rr = 0; rr < RPB; ++rr) {
    const int row = row0 + rr;
    if (row >= rows) break;
    const TA* xr = x + (int64_t)row * ldx;
    const TD* dyr = dy + (int64_t)(row_map ? row_map[row] : row) * lddy;
    float s = 0.f, ss = 0.f;
    for (int c = tid; c < dim; c += 256) { const float v = Cvt<TA>::ld(xr + c); s += v; ss = fmaf(v, v, ss); }
    s = wave_sum(s); ss = wave_sum(ss);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = s; red[4 + (tid >> 6)] = ss; }
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    ss = red[4] + red[5] + red[6] + red[7];
    const float mean = s / (float)dim;
    const float rstd = rsqrtf(fmaxf(ss / (float)dim - mean * mean, 0.f) + eps);
    float a1 = 0.f, a2 = 0.f;    // sum(dxhat), sum(dxhat*xhat)
    for (int c = tid; c < dim; c += 256) {
      const float xh = (Cvt<TA>::ld(xr + c) - mean) * rstd, g = Cvt<TD>::ld(dyr + c) * w[c];
      a1 += g; a2 = fmaf(g, xh, a2);
    }
    a1 = wave_sum(a1); a2 = wave_sum(a2);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = a1; red[4 + (tid >> 6)] = a2; }
    __syncthreads();
    a1 = (red[0] + red[1] + red[2] + red[3]) / (float)dim;
    a2 = (red[4] + red[5] + red[6] + red[7]) / (float)dim;
    TA* dxr = dx + (int64_t)row * lddx;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * 256;
      if (c < dim) {
        const float xh = (Cvt<TA>::ld(xr + c) - mean) * rstd, dyv = Cvt<TD>::ld(dyr + c);
        Cvt<TA>::st(dxr + c, rstd * (dyv * w[c] - a1 - xh * a2));
        dwp[i] += dyv * xh;
        dbp[i] += dyv;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * 256;
    if (c < dim) { atomicAdd(dw + c, dwp[i]); atomicAdd(db + c, dbp[i]); }
  }
}

// ------------------------------------------------------------------ SwiGLU on the interleaved layout
// gu[r, 32 j + t] = gate col 16 j + t (t<16), gu[r, 32 j + 16 + t] = up col 16 j + t
// 8 consecutive columns per thread (16-B bf16 accesses); F % 16 == 0 so a vector never straddles a 16-column block.
template <typename TA, int NT = 0>   // NT: bit 0 non-temporal loads, bit 1 non-temporal stores
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const TA* __restrict__ gu, int64_t ldg, TA* __restrict__ act, int64_t lda, int rows, int F, int inter) {
  const int F8 = F >> 3;
  const int64_t n = (int64_t)rows * F8;
  const int ustep = inter ? 16 : F;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / F8), c = (int)(i - (int64_t)r * F8) * 8;
    const TA* p = gu + (int64_t)r * ldg + (inter ? (c >> 4) * 32 + (c & 15) : c);
    float g[8], u[8], o[8];
    load8_s<(NT & 1) != 0>(p, g);
    load8_s<(NT & 1) != 0>(p + ustep, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = Cvt<TA>::rnd(g[e] * __builtin_amdgcn_rcpf(1.f + __expf(-g[e]))) * u[e];   // = silu() of the GEMM epilogue
    store8_s<(NT & 2) != 0>(act + (int64_t)r * lda + c, o);
  }
}
template <typename TA, int NT = 0>   // NT: bit 0 non-temporal loads, bit 1 non-temporal stores
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const TA* __restrict__ gu, int64_t ldg, const TA* __restrict__ dact, int64_t lda,
                                                         TA* __restrict__ dgu, int64_t lddg, int rows, int F, int inter) {
  const int F8 = F >> 3;
  const int64_t n = (int64_t)rows * F8;
  const int ustep = inter ? 16 : F;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / F8), c = (int)(i - (int64_t)r * F8) * 8;
    const int64_t o = inter ? (int64_t)(c >> 4) * 32 + (c & 15) : c;
    float g[8], u[8], da[8], dg[8], du[8];
    load8_s<(NT & 1) != 0>(gu + (int64_t)r * ldg + o, g);
    load8_s<(NT & 1) != 0>(gu + (int64_t)r * ldg + o + ustep, u);
    load8_s<(NT & 1) != 0>(dact + (int64_t)r * lda + c, da);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      swiglu_bwd_pair(g[e], u[e], da[e], dg[e], du[e]);
    }
    store8_s<(NT & 2) != 0>(dgu + (int64_t)r * lddg + o, dg);
    store8_s<(NT & 2) != 0>(dgu + (int64_t)r * lddg + o + ustep, du);
  }
}

// ------------------------------------------------------------------ RoPE backward + pack into dqkv
// dq [B,S,H,hd] (contiguous), dk/dv [B,Hkv,S,hd] -> dqkv [B*S, (H+2Hkv)*hd]; q,k rotated by -theta.
template <typename TA>
__global__ __launch_bounds__(256) void rope_bwd_pack_kernel(const TA* __restrict__ dq, const TA* __restrict__ dk, const TA* __restrict__ dv,
                                                            TA* __restrict__ dqkv, int64_t ld, const float* __restrict__ cos_sin,
                                                            int S, int H, int Hkv, int hd, int rope_pos0) {
  const int row = blockIdx.x;              // b*S + s
  const int b = row / S, s = row % S;
  const int slots = H + 2 * Hkv, half = hd / 2;
  for (int i = threadIdx.x; i < slots * half; i += 256) {
    const int slot = i / half, pr = i % half;
    const TA* src;
    if (slot < H) src = dq + ((int64_t)row * H + slot) * hd;
    else if (slot < H + Hkv) src = dk + (((int64_t)b * Hkv + (slot - H)) * S + s) * hd;
    else src = dv + (((int64_t)b * Hkv + (slot - H - Hkv)) * S + s) * hd;
    const float a = Cvt<TA>::ld(src + 2 * pr), bb = Cvt<TA>::ld(src + 2 * pr + 1);
    TA* dst = dqkv + (int64_t)row * ld + (int64_t)slot * hd + 2 * pr;
    if (slot < H + Hkv) {
      const float co = cos_sin[((int64_t)(rope_pos0 + s) * half + pr) * 2], si = cos_sin[((int64_t)(rope_pos0 + s) * half + pr) * 2 + 1];
      Cvt<TA>::st(dst, a * co + bb * si);
      Cvt<TA>::st(dst + 1, -a * si + bb * co);
    } else {
      Cvt<TA>::st(dst, a);
      Cvt<TA>::st(dst + 1, bb);
    }
  }
}

// ------------------------------------------------------------------ attention backward, generic (any hd <= 256)
// D[b,h,q] = sum_d dO*O
template <typename TA>
__global__ __launch_bounds__(64) void attn_rowdot_kernel(const TA* __restrict__ o, const TA* __restrict__ dout, float* __restrict__ D, int hd) {
  const int64_t row = blockIdx.x;       // (b*S + s)*H + h  (o, dout contiguous [B,S,H,hd])
  float a = 0.f;
  for (int d = threadIdx.x; d < hd; d += 64) a = fmaf(Cvt<TA>::ld(o + row * hd + d), Cvt<TA>::ld(dout + row * hd + d), a);
  a = wave_sum(a);
  if (threadIdx.x == 0) D[row] = a;     // index (b, s, h)
}

// bf16, hd = 64 / 128: hd/8 lanes per row with one 16-B load of each operand, 256 / (hd/8) rows per block (the one-wave-per-row
// form above moved 142 MB in 63 us; this one streams it at HBM rate)
template <int HD>
__global__ __launch_bounds__(256) void attn_rowdot_vec_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                              float* __restrict__ D, int64_t rows, int H = 1, int64_t ld_o = HD) {
  constexpr int LPR = HD / 8;                       // lanes per row: 16 (hd 128) or 8 (hd 64)
  const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const int sub = threadIdx.x % LPR;
  float a = 0.f;
  if (row < rows) {
    float x[8], y[8];
    const int64_t tok = row / H;                    // o may have a padded token stride (ld_o >= H * HD); dout is contiguous
    load8(o + tok * ld_o + (row - tok * H) * HD + sub * 8, x);
    load8(dout + row * HD + sub * 8, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) a = fmaf(x[e], y[e], a);
  }
#pragma unroll
  for (int off = LPR / 2; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
  if (row < rows && sub == 0) D[row] = a;
}

struct AttnBwdArgs {
  const void* q; const void* k; const void* v; const void* dout;   // q,dout [B,S,H,hd]; k [B,Hkv,S,hd]; v rows via (v_sb,v_ss,v_sh)
  const float* lse;   // [B,H,S]
  const float* D;     // [B,S,H]
  void* dq; void* dk; void* dv;      // dq [B,S,H,hd]; dk,dv [B,Hkv,S,hd]
  int64_t v_sb, v_ss, v_sh;
  int64_t k_sb, k_sh;           // k strides: batch, kv-head (seq stride = hd)
  int B, S, H, Hkv, hd, causal;
  float scale;
};

// dQ: one wave per (q, h, b): dq[d] = scale * sum_kv p*(dp - D) * k[kv][d]
template <typename TA>
__global__ __launch_bounds__(64) void attn_bwd_dq_generic(AttnBwdArgs p) {
  __shared__ float qs[256], dos[256];
  const int lane = threadIdx.x, qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (p.H / p.Hkv), hd = p.hd;
  const TA* Q = (const TA*)p.q + (((int64_t)b * p.S + qi) * p.H + h) * hd;
  const TA* DO = (const TA*)p.dout + (((int64_t)b * p.S + qi) * p.H + h) * hd;
  const TA* K = (const TA*)p.k + b * p.k_sb + hk * p.k_sh;
  const TA* V = (const TA*)p.v + b * p.v_sb + hk * p.v_sh;
  for (int d = lane; d < hd; d += 64) { qs[d] = Cvt<TA>::ld(Q + d); dos[d] = Cvt<TA>::ld(DO + d); }
  __syncthreads();
  const float lse = p.lse[((int64_t)b * p.H + h) * p.S + qi], Dq = p.D[((int64_t)b * p.S + qi) * p.H + h];
  const int kv_end = p.causal ? qi + 1 : p.S;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kv = 0; kv < kv_end; ++kv) {
    float s = 0.f, dp = 0.f;
    for (int d = lane; d < hd; d += 64) {
      s = fmaf(qs[d], Cvt<TA>::ld(K + (int64_t)kv * hd + d), s);
      dp = fmaf(dos[d], Cvt<TA>::ld(V + (int64_t)kv * p.v_ss + d), dp);
    }
    s = wave_sum(s); dp = wave_sum(dp);
    const float pr = __expf(s * p.scale - lse);
    const float ds = pr * (dp - Dq) * p.scale;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = lane + 64 * i;
      if (d < hd) acc[i] = fmaf(ds, Cvt<TA>::ld(K + (int64_t)kv * hd + d), acc[i]);
    }
  }
  TA* DQ = (TA*)p.dq + (((int64_t)b * p.S + qi) * p.H + h) * hd;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = lane + 64 * i;
    if (d < hd) Cvt<TA>::st(DQ + d, acc[i]);
  }
}

// dK, dV: one wave per (kv, hk, b): loops over the n_rep query heads and the queries that see kv
template <typename TA>
__global__ __launch_bounds__(64) void attn_bwd_dkv_generic(AttnBwdArgs p) {
  __shared__ float ks[256], vs[256];
  const int lane = threadIdx.x, kv = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const int hd = p.hd, nrep = p.H / p.Hkv;
  const TA* K = (const TA*)p.k + b * p.k_sb + hk * p.k_sh + (int64_t)kv * hd;
  const TA* V = (const TA*)p.v + b * p.v_sb + hk * p.v_sh + (int64_t)kv * p.v_ss;
  for (int d = lane; d < hd; d += 64) { ks[d] = Cvt<TA>::ld(K + d); vs[d] = Cvt<TA>::ld(V + d); }
  __syncthreads();
  float dk[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int rep = 0; rep < nrep; ++rep) {
    const int h = hk * nrep + rep;
    for (int qi = p.causal ? kv : 0; qi < p.S; ++qi) {
      const TA* Q = (const TA*)p.q + (((int64_t)b * p.S + qi) * p.H + h) * hd;
      const TA* DO = (const TA*)p.dout + (((int64_t)b * p.S + qi) * p.H + h) * hd;
      float s = 0.f, dp = 0.f;
      for (int d = lane; d < hd; d += 64) {
        s = fmaf(Cvt<TA>::ld(Q + d), ks[d], s);
        dp = fmaf(Cvt<TA>::ld(DO + d), vs[d], dp);
      }
      s = wave_sum(s); dp = wave_sum(dp);
      const float pr = __expf(s * p.scale - p.lse[((int64_t)b * p.H + h) * p.S + qi]);
      const float ds = pr * (dp - p.D[((int64_t)b * p.S + qi) * p.H + h]) * p.scale;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = lane + 64 * i;
        if (d < hd) {
          dk[i] = fmaf(ds, Cvt<TA>::ld(Q + d), dk[i]);
          dv[i] = fmaf(pr, Cvt<TA>::ld(DO + d), dv[i]);
        }
      }
    }
  }
  TA* DK = (TA*)p.dk + (((int64_t)b * p.Hkv + hk) * p.S + kv) * hd;
  TA* DV = (TA*)p.dv + (((int64_t)b * p.Hkv + hk) * p.S + kv) * hd;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = lane + 64 * i;
    if (d < hd) { Cvt<TA>::st(DK + d, dk[i]); Cvt<TA>::st(DV + d, dv[i]); }
  }
}

// ------------------------------------------------------------------ embedding / row reductions
// d_table[tok] += dh[row] for the text rows of [BOS | W image words | text]
template <typename TD>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ tokens, int64_t ld_tok, const TD* __restrict__ dh,
                                                        float* __restrict__ dtable, int T, int W, int dim, int vocab) {
  const int S = T + W, row = blockIdx.x, b = row / S, s = row % S;
  if (s >= 1 && s <= W) return;
  int64_t tok = tokens[(int64_t)b * ld_tok + (s == 0 ? 0 : s - W)];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  for (int c = threadIdx.x; c < dim; c += 256) atomicAdd(dtable + tok * dim + c, Cvt<TD>::ld(dh + (int64_t)row * dim + c));
}

// out[c] += sum_i src[row_idx ? row_idx[i] : i][c]
template <typename T>
__global__ __launch_bounds__(256) void rows_sum_kernel(const T* __restrict__ src, int64_t ld, const int32_t* __restrict__ row_idx,
                                                       int n_rows, int dim, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= dim) return;
  const int r0 = blockIdx.y * 64, r1 = min(n_rows, r0 + 64);
  float a = 0.f;
  for (int i = r0; i < r1; ++i) a += Cvt<T>::ld(src + (int64_t)(row_idx ? row_idx[i] : i) * ld + c);
  atomicAdd(out + c, a);
}

// dst[r, c] = (TD) src[r, c]
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast2d_kernel(const TS* __restrict__ src, int64_t lds_, TD* __restrict__ dst, int64_t ldd, int rows, int cols) {
  const int64_t n = (int64_t)rows * (cols / 8);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = i / (cols / 8), c = (i % (cols / 8)) * 8;
    float v[8];
    load8(src + (int64_t)r * lds_ + c, v);
    store8(dst + (int64_t)r * ldd + c, v);
  }
}

// dst[r, c] += src[r, c]  (scatter-accumulate of the block-diagonal LoRA gradient GEMMs into the adapter grads; the
// adapter path's residual add h += o)
template <typename T>
__global__ __launch_bounds__(256) void add2d_kernel(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src, int64_t lds_, int rows, int cols) {
  const int64_t n = (int64_t)rows * (cols / 4);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = i / (cols / 4), c = (i % (cols / 4)) * 4;
    T* d = dst + (int64_t)r * ldd + c;
    const T* s2 = src + (int64_t)r * lds_ + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) Cvt<T>::st(d + e, Cvt<T>::ld(d + e) + Cvt<T>::ld(s2 + e));
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int a3v_add2d(void* dst, int64_t ld_dst, const void* src, int64_t ld_src, int rows, int cols, int dtype, void* stream) {
  if (!dst || !src || rows <= 0 || cols <= 0) return A3V_ERR_ARG;
  if (cols % 4) return A3V_ERR_SHAPE;
  const int64_t n = (int64_t)rows * (cols / 4);
  const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  if (dtype == A3V_F32) hipLaunchKernelGGL(add2d_kernel<float>, dim3(blocks), dim3(256), 0, ST, (float*)dst, ld_dst, (const float*)src, ld_src, rows, cols);
  else if (dtype == A3V_BF16) hipLaunchKernelGGL(add2d_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, (bf16_t*)dst, ld_dst, (const bf16_t*)src, ld_src, rows, cols);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_cast(const void* src, int64_t ld_src, int src_dtype, void* dst, int64_t ld_dst, int dst_dtype, int rows, int cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return A3V_ERR_ARG;
  if (cols % 8 || ld_src % 8 || ld_dst % 8) return A3V_ERR_SHAPE;
  const int64_t n = (int64_t)rows * (cols / 8);
  const int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
  dim3 g(blocks), b(256);
  switch (src_dtype * 2 + dst_dtype) {
    case 2: hipLaunchKernelGGL((cast2d_kernel<float, bf16_t>), g, b, 0, ST, (const float*)src, ld_src, (bf16_t*)dst, ld_dst, rows, cols); break;
    case 1: hipLaunchKernelGGL((cast2d_kernel<bf16_t, float>), g, b, 0, ST, (const bf16_t*)src, ld_src, (float*)dst, ld_dst, rows, cols); break;
    case 3: hipLaunchKernelGGL((cast2d_kernel<float, float>), g, b, 0, ST, (const float*)src, ld_src, (float*)dst, ld_dst, rows, cols); break;
    case 0: hipLaunchKernelGGL((cast2d_kernel<bf16_t, bf16_t>), g, b, 0, ST, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, rows, cols); break;
    default: return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// dst[i] = (TD)(src[i] * scale): the DP reducer's wire-dtype conversions (fp32 gradient bucket -> bf16 wire bucket pre-scaled by
// 1/world, and back) in one pass each instead of mul_ + to() + copy_ (dp.py GradReducer; reference: FSDP reduce_dtype,
// main_finetune.py:251-255).  16-B accesses on the wider side, every line touched once.
namespace {
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void scale_cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n, float scale) {
  const int64_t n8 = n / 8, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    float v[8];
    load8_s<true>(src + i * 8, v);                   // gradient buckets are touched once per pass: streaming loads and stores
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= scale;
    store8_s<true>(dst + i * 8, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n8 * 8)) {
    const int64_t i = n8 * 8 + threadIdx.x;
    Cvt<TD>::st(dst + i, Cvt<TS>::ld(src + i) * scale);
  }
}
}  // namespace

extern "C" int a3v_scale_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, float scale, void* stream) {
  if (!src || !dst || n <= 0) return A3V_ERR_ARG;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return A3V_ERR_SHAPE;
  int64_t blocks = (n / 8 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks);
  dim3 g((unsigned)blocks), b(256);
  switch (src_dtype * 2 + dst_dtype) {
    case 2: hipLaunchKernelGGL((scale_cast_kernel<float, bf16_t>), g, b, 0, ST, (const float*)src, (bf16_t*)dst, n, scale); break;
    case 1: hipLaunchKernelGGL((scale_cast_kernel<bf16_t, float>), g, b, 0, ST, (const bf16_t*)src, (float*)dst, n, scale); break;
    case 3: hipLaunchKernelGGL((scale_cast_kernel<float, float>), g, b, 0, ST, (const float*)src, (float*)dst, n, scale); break;
    case 0: hipLaunchKernelGGL((scale_cast_kernel<bf16_t, bf16_t>), g, b, 0, ST, (const bf16_t*)src, (bf16_t*)dst, n, scale); break;
    default: return A3V_ERR_DTYPE;
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// Sum of squares of an fp32 range as A3V_SUMSQ_SLOTS partial sums (block b owns the 16-byte vectors b, b + SLOTS, ...; fixed order
// inside a block): the global-norm clip (util/clip_grad.py:59-210 of the reference) takes sqrt(sum of all partials).  Launched per
// gradient bucket on a side stream as soon as the bucket is final, so the 27 GB read of the clip overlaps the rest of the backward.
namespace {
// (slot s always sums the same vectors -- s, s + SLOTS, ... -- whatever the launch width: a narrow grid, which disturbs the GEMMs it
//  runs beside less, gives bit-identical sums)
__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[2][4];
  const int64_t n4 = n / 4, stride = (int64_t)A3V_SUMSQ_SLOTS * 256;
  int par = 0;
  for (int vb = blockIdx.x; vb < A3V_SUMSQ_SLOTS; vb += gridDim.x, par ^= 1) {
    if (vb != 0 && (int64_t)vb * 256 >= n4) {             // a slot with no vector of this (short) range: block-uniform, no barrier
      if (threadIdx.x == 0) out[vb] = 0.f;
      par ^= 1;
      continue;
    }
    float a = 0.f;
    for (int64_t i = (int64_t)vb * 256 + threadIdx.x; i < n4; i += stride) {
      const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + i);
      a = fmaf(v[0], v[0], a); a = fmaf(v[1], v[1], a); a = fmaf(v[2], v[2], a); a = fmaf(v[3], v[3], a);
    }
    if (vb == 0 && threadIdx.x < (int)(n - n4 * 4)) { const float t = x[n4 * 4 + threadIdx.x]; a = fmaf(t, t, a); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[par][threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[vb] = (red[par][0] + red[par][1]) + (red[par][2] + red[par][3]);
  }
}
}  // namespace

extern "C" int a3v_sumsq_partials(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n <= 0) return A3V_ERR_ARG;
  if ((uintptr_t)x & 15) return A3V_ERR_SHAPE;
  int nb = A3V_ENV_INT("A3V_SUMSQ_BLOCKS", A3V_SUMSQ_SLOTS);                 // launch width: default 1024 = one block per slot
  if (nb < 1 || nb > A3V_SUMSQ_SLOTS) nb = A3V_SUMSQ_SLOTS;
  // short ranges (the norm weights left over next to the GEMM-summed matrices): one block per slot that has data -- 1024 blocks for
  // 8 K floats only queue behind the GEMM the call runs beside (same sums: slot s always adds the same vectors)
  const int64_t useful = (n / 4 + 255) / 256;
  if (useful < nb) nb = useful < 1 ? 1 : (int)useful;
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(nb), dim3(256), 0, ST, x, n, out);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_transpose(const void* src, int64_t ld_src, int64_t bs_src, void* dst, int64_t ld_dst, int64_t bs_dst,
                             int R, int C, int Rpad, int batch, int dtype, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || Rpad < R || batch <= 0) return A3V_ERR_ARG;
  dim3 g((C + 63) / 64, (Rpad + 63) / 64, batch);
  if (dtype == A3V_BF16 && C % 8 == 0 && Rpad % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && bs_src % 8 == 0 && bs_dst % 8 == 0 &&
      ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    hipLaunchKernelGGL(transpose_bf16_vec_kernel, g, dim3(256), 0, ST, (const bf16_t*)src, ld_src, bs_src, (bf16_t*)dst, ld_dst, bs_dst, R, C, Rpad);
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  if (dtype == A3V_BF16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, g, dim3(256), 0, ST, (const bf16_t*)src, ld_src, bs_src, (bf16_t*)dst, ld_dst, bs_dst, R, C, Rpad);
  else if (dtype == A3V_F32) hipLaunchKernelGGL(transpose_kernel<float>, g, dim3(256), 0, ST, (const float*)src, ld_src, bs_src, (float*)dst, ld_dst, bs_dst, R, C, Rpad);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// two-level batched bf16 transpose (internal: the attention backward's K^T / Q^T / dO^T images in one launch each)
int a3v_transpose_2level(const bf16_t* src, int64_t ld_src, int64_t bs_in, int64_t bs_out, bf16_t* dst, int64_t ld_dst, int64_t bsd_in,
                         int64_t bsd_out, int R, int C, int Rpad, int n_in, int n_out, void* stream) {
  const bool vec = C % 8 == 0 && Rpad % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && bs_in % 8 == 0 && bs_out % 8 == 0 && bsd_in % 8 == 0 &&
                   bsd_out % 8 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 &&
                   (int64_t)n_in * n_out <= 65535;
  if (!vec) {
    for (int o = 0; o < n_out; ++o) {
      const int rc = a3v_transpose(src + o * bs_out, ld_src, bs_in, dst + o * bsd_out, ld_dst, bsd_in, R, C, Rpad, n_in, A3V_BF16, stream);
      if (rc) return rc;
    }
    return A3V_OK;
  }
  dim3 g((C + 63) / 64, (Rpad + 63) / 64, n_in * n_out);
  hipLaunchKernelGGL(transpose_bf16_vec_kernel, g, dim3(256), 0, ST, src, ld_src, bs_in, dst, ld_dst, bsd_in, R, C, Rpad, n_in, bs_out, bsd_out);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int64_t a3v_rmsnorm_bwd_scratch_floats(int rows, int dim) { return (int64_t)((rows + 7) / 8) * dim; }

template <typename TS>
static int rmsnorm_bwd_impl(const TS* x, int64_t ldx, const float* w, const void* dy, int64_t lddy, TS* dh, int64_t lddh,
                            float* dw, float* dw_scratch, int rows, int dim, float eps, int act_dtype, void* stream, bf16_t* dh_bf,
                            int64_t ld_bf) {
  if (!x || !w || !dy || !dh || rows <= 0) return A3V_ERR_ARG;
  if (dim > 8192) return A3V_ERR_SHAPE;
  constexpr uintptr_t SMASK = sizeof(TS) == 2 ? 7 : 15;
  if (dim % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddh % 4 == 0 && (act_dtype == A3V_BF16 || act_dtype == A3V_F32) &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dh)) & SMASK) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(dy) & (act_dtype == A3V_BF16 ? 7 : 15)) == 0) {
    constexpr int RV = 8;
    const int nb = (rows + RV - 1) / RV;
    dim3 gv(nb);
    const bool part = dw && dw_scratch && (reinterpret_cast<uintptr_t>(dw_scratch) & 15) == 0;
    float* dwo = part ? dw_scratch : dw;
#define A3V_RB(TT, MV, PP) hipLaunchKernelGGL((rmsnorm_bwd_vec_kernel<TT, RV, MV, PP, TS>), gv, dim3(256), 0, ST, x, ldx, w, (const TT*)dy, lddy, dh, lddh, dwo, rows, dim, eps, dh_bf, ld_bf)
    if (dim <= 4096) {
      if (act_dtype == A3V_BF16) { if (part) A3V_RB(bf16_t, 4, true); else A3V_RB(bf16_t, 4, false); }
      else { if (part) A3V_RB(float, 4, true); else A3V_RB(float, 4, false); }
    } else {
      if (act_dtype == A3V_BF16) { if (part) A3V_RB(bf16_t, 8, true); else A3V_RB(bf16_t, 8, false); }
      else { if (part) A3V_RB(float, 8, true); else A3V_RB(float, 8, false); }
    }
#undef A3V_RB
    A3V_LAUNCH_CHECK();
    if (part) {
      hipLaunchKernelGGL(colsum_add_kernel, dim3((dim + 255) / 256, 16), dim3(256), 0, ST, dw_scratch, nb, dim, dw);
      A3V_LAUNCH_CHECK();
    }
    return A3V_OK;
  }
  constexpr int RPB = 16;
  dim3 g((rows + RPB - 1) / RPB);
  if (act_dtype == A3V_BF16) hipLaunchKernelGGL((rmsnorm_bwd_kernel<bf16_t, RPB, TS>), g, dim3(256), 0, ST, x, ldx, w, (const bf16_t*)dy, lddy, dh, lddh, dw, rows, dim, eps);
  else if (act_dtype == A3V_F32) hipLaunchKernelGGL((rmsnorm_bwd_kernel<float, RPB, TS>), g, dim3(256), 0, ST, x, ldx, w, (const float*)dy, lddy, dh, lddh, dw, rows, dim, eps);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  if (dh_bf) {
    if (sizeof(TS) != 4) return A3V_ERR_ARG;
    return a3v_cast(dh, lddh, A3V_F32, dh_bf, ld_bf, A3V_BF16, rows, dim, stream);     // (scalar form of the kernel: the copy as a pass)
  }
  return A3V_OK;
}

extern "C" int a3v_rmsnorm_bwd(const float* x, int64_t ldx, const float* w, const void* dy, int64_t lddy, float* dh, int64_t lddh,
                               float* dw, float* dw_scratch, int rows, int dim, float eps, int act_dtype, void* stream) {
  return rmsnorm_bwd_impl<float>(x, ldx, w, dy, lddy, dh, lddh, dw, dw_scratch, rows, dim, eps, act_dtype, stream, nullptr, 0);
}

extern "C" int a3v_rmsnorm_bwd_cast(const float* x, int64_t ldx, const float* w, const void* dy, int64_t lddy, float* dh, int64_t lddh,
                                    float* dw, float* dw_scratch, int rows, int dim, float eps, int act_dtype, void* dh_bf16,
                                    int64_t ld_bf16, void* stream) {
  if (!dh_bf16 || (ld_bf16 & 3) || (reinterpret_cast<uintptr_t>(dh_bf16) & 7)) return A3V_ERR_ARG;
  return rmsnorm_bwd_impl<float>(x, ldx, w, dy, lddy, dh, lddh, dw, dw_scratch, rows, dim, eps, act_dtype, stream, (bf16_t*)dh_bf16, ld_bf16);
}

// bf16 residual stream (training under FSDP MixedPrecision(param_dtype=bf16) + autocast, main_finetune.py:241-263, engine_finetune.py:44-50:
// embeddings, block outputs and therefore the stream and its gradient are bf16 tensors): x, dy and the accumulated dh all bf16
extern "C" int a3v_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const float* w, const void* dy, int64_t lddy, void* dh, int64_t lddh,
                                    float* dw, float* dw_scratch, int rows, int dim, float eps, void* stream) {
  return rmsnorm_bwd_impl<bf16_t>((const bf16_t*)x, ldx, w, dy, lddy, (bf16_t*)dh, lddh, dw, dw_scratch, rows, dim, eps, A3V_BF16, stream, nullptr, 0);
}

static int layernorm_bwd_impl(const void* x, int64_t ldx, const float* w, const void* dy, int dy_dtype, int64_t lddy, const int32_t* row_map,
                              void* dx, int64_t lddx, float* dw, float* db, int rows, int dim, float eps, int act_dtype, void* stream) {
  if (!x || !w || !dy || !dx || !dw || !db || rows <= 0) return A3V_ERR_ARG;
  if (dim > 8192) return A3V_ERR_SHAPE;
  constexpr int RPB = 16;
  dim3 g((rows + RPB - 1) / RPB);
  if (dy_dtype == A3V_F32) {
    if (act_dtype == A3V_BF16) hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, RPB, float>), g, dim3(256), 0, ST, (const bf16_t*)x, ldx, w, (const float*)dy, lddy, row_map, (bf16_t*)dx, lddx, dw, db, rows, dim, eps);
    else if (act_dtype == A3V_F32) hipLaunchKernelGGL((layernorm_bwd_kernel<float, RPB, float>), g, dim3(256), 0, ST, (const float*)x, ldx, w, (const float*)dy, lddy, row_map, (float*)dx, lddx, dw, db, rows, dim, eps);
    else return A3V_ERR_DTYPE;
  } else if (dy_dtype == A3V_BF16 && act_dtype == A3V_BF16) {
    hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, RPB, bf16_t>), g, dim3(256), 0, ST, (const bf16_t*)x, ldx, w, (const bf16_t*)dy, lddy, row_map, (bf16_t*)dx, lddx, dw, db, rows, dim, eps);
  } else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_layernorm_bwd(const void* x, int64_t ldx, const float* w, const float* dy, int64_t lddy, const int32_t* row_map,
                                 void* dx, int64_t lddx, float* dw, float* db, int rows, int dim, float eps, int act_dtype, void* stream) {
  return layernorm_bwd_impl(x, ldx, w, dy, A3V_F32, lddy, row_map, dx, lddx, dw, db, rows, dim, eps, act_dtype, stream);
}

// the same with the stream gradient in bf16 (bf16 residual stream, see a3v_rmsnorm_bwd_bf16)
extern "C" int a3v_layernorm_bwd_bf16(const void* x, int64_t ldx, const float* w, const void* dy, int64_t lddy, const int32_t* row_map,
                                      void* dx, int64_t lddx, float* dw, float* db, int rows, int dim, float eps, void* stream) {
  return layernorm_bwd_impl(x, ldx, w, dy, A3V_BF16, lddy, row_map, dx, lddx, dw, db, rows, dim, eps, A3V_BF16, stream);
}

// cache policy of the streaming training kernels (A3V_STREAM_NT: bit 0 non-temporal loads, bit 1 stores; read per launch)
static int stream_nt() {
  return A3V_ENV_INT("A3V_STREAM_NT", 3) & 3;      // default: both (SwiGLU forward 102 -> 91 us, backward 174 -> 154 us at 7B size, tools/stream_nt_bench.py)
}

extern "C" int a3v_swiglu_fwd(const void* gu, int64_t ldg, void* act, int64_t lda, int rows, int F, int interleaved, int dtype, void* stream) {
  const int inter = interleaved;
  if (!gu || !act || rows <= 0 || F <= 0 || (F % 16)) return A3V_ERR_ARG;
  if ((ldg % 8) || (lda % 8)) return A3V_ERR_SHAPE;
  int64_t n = (int64_t)rows * (F / 8);
  int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
  const int nt = stream_nt();
  if (dtype == A3V_BF16 && nt == 3) hipLaunchKernelGGL((swiglu_fwd_kernel<bf16_t, 3>), dim3(blocks), dim3(256), 0, ST, (const bf16_t*)gu, ldg, (bf16_t*)act, lda, rows, F, inter);
  else if (dtype == A3V_BF16 && nt == 1) hipLaunchKernelGGL((swiglu_fwd_kernel<bf16_t, 1>), dim3(blocks), dim3(256), 0, ST, (const bf16_t*)gu, ldg, (bf16_t*)act, lda, rows, F, inter);
  else if (dtype == A3V_BF16) hipLaunchKernelGGL(swiglu_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, (const bf16_t*)gu, ldg, (bf16_t*)act, lda, rows, F, inter);
  else if (dtype == A3V_F32) hipLaunchKernelGGL(swiglu_fwd_kernel<float>, dim3(blocks), dim3(256), 0, ST, (const float*)gu, ldg, (float*)act, lda, rows, F, inter);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_swiglu_bwd(const void* gu, int64_t ldg, const void* dact, int64_t lda, void* dgu, int64_t lddg, int rows, int F,
                              int interleaved, int dtype, void* stream) {
  const int inter = interleaved;
  if (!gu || !dact || !dgu || rows <= 0 || F <= 0 || (F % 16)) return A3V_ERR_ARG;
  if ((ldg % 8) || (lda % 8) || (lddg % 8)) return A3V_ERR_SHAPE;
  int64_t n = (int64_t)rows * (F / 8);
  int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
  const int nt = stream_nt();
  if (dtype == A3V_BF16 && nt == 3) hipLaunchKernelGGL((swiglu_bwd_kernel<bf16_t, 3>), dim3(blocks), dim3(256), 0, ST, (const bf16_t*)gu, ldg, (const bf16_t*)dact, lda, (bf16_t*)dgu, lddg, rows, F, inter);
  else if (dtype == A3V_BF16 && nt == 1) hipLaunchKernelGGL((swiglu_bwd_kernel<bf16_t, 1>), dim3(blocks), dim3(256), 0, ST, (const bf16_t*)gu, ldg, (const bf16_t*)dact, lda, (bf16_t*)dgu, lddg, rows, F, inter);
  else if (dtype == A3V_BF16) hipLaunchKernelGGL(swiglu_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, (const bf16_t*)gu, ldg, (const bf16_t*)dact, lda, (bf16_t*)dgu, lddg, rows, F, inter);
  else if (dtype == A3V_F32) hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3(blocks), dim3(256), 0, ST, (const float*)gu, ldg, (const float*)dact, lda, (float*)dgu, lddg, rows, F, inter);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_rope_bwd_pack(const void* dq, const void* dk, const void* dv, void* dqkv, int64_t ld, const float* cos_sin,
                                 int B, int S, int H, int Hkv, int hd, int rope_pos0, int dtype, void* stream) {
  if (!dq || !dk || !dv || !dqkv || !cos_sin || B <= 0 || S <= 0 || (hd & 1)) return A3V_ERR_ARG;
  if (dtype == A3V_BF16) hipLaunchKernelGGL(rope_bwd_pack_kernel<bf16_t>, dim3(B * S), dim3(256), 0, ST, (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)dv, (bf16_t*)dqkv, ld, cos_sin, S, H, Hkv, hd, rope_pos0);
  else if (dtype == A3V_F32) hipLaunchKernelGGL(rope_bwd_pack_kernel<float>, dim3(B * S), dim3(256), 0, ST, (const float*)dq, (const float*)dk, (const float*)dv, (float*)dqkv, ld, cos_sin, S, H, Hkv, hd, rope_pos0);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// Attention backward (self-attention, Sq == Sk == S, queries/keys at the same positions):
// inputs q,out,dout [B,S,H,hd] contiguous; k [B,Hkv,S,hd]; v rows addressed by (v_sb, v_ss, v_sh) element strides
// (e.g. straight out of the fused qkv activation); lse [B,H,S]; scratch D [B,S,H] floats.
// outputs dq [B,S,H,hd]; dk, dv [B,Hkv,S,hd].
extern "C" int a3v_attention_bwd_mfma(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb,
                                      int64_t v_ss, int64_t v_sh, const void* dout, const float* lse, const float* D, void* dq,
                                      void* dk, void* dv, void* workspace, int B, int S, int H, int Hkv, int hd, int causal,
                                      void* stream);

extern "C" int a3v_attention_bwd(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb, int64_t v_ss, int64_t v_sh,
                                 const void* out, const void* dout, const float* lse, float* D, void* dq, void* dk, void* dv,
                                 void* workspace, int B, int S, int H, int Hkv, int hd, int causal, int dtype, void* stream) {
  if (!q || !k || !v || !out || !dout || !lse || !D || !dq || !dk || !dv || B <= 0 || S <= 0) return A3V_ERR_ARG;
  if (hd > 256 || (H % Hkv)) return A3V_ERR_SHAPE;
  AttnBwdArgs p;
  p.q = q; p.k = k; p.v = v; p.dout = dout; p.lse = lse; p.D = D; p.dq = dq; p.dk = dk; p.dv = dv;
  p.v_sb = v_sb; p.v_ss = v_ss; p.v_sh = v_sh;
  p.k_sb = k_sb; p.k_sh = k_sh;
  p.B = B; p.S = S; p.H = H; p.Hkv = Hkv; p.hd = hd; p.causal = causal;
  p.scale = 1.0f / sqrtf((float)hd);
  if (dtype == A3V_BF16 && workspace && (hd == 64 || hd == 128)) {     // MFMA kernels (a3v_attn_bwd.hip)
    const int64_t rows = (int64_t)B * S * H;
    if (((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout)) & 15) != 0)
      hipLaunchKernelGGL(attn_rowdot_kernel<bf16_t>, dim3(B * S * H), dim3(64), 0, ST, (const bf16_t*)out, (const bf16_t*)dout, D, hd);
    else if (hd == 128)
      hipLaunchKernelGGL(attn_rowdot_vec_kernel<128>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, ST, (const bf16_t*)out, (const bf16_t*)dout, D, rows);
    else
      hipLaunchKernelGGL(attn_rowdot_vec_kernel<64>, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, ST, (const bf16_t*)out, (const bf16_t*)dout, D, rows);
    A3V_LAUNCH_CHECK();
    return a3v_attention_bwd_mfma(q, k, k_sb, k_sh, v, v_sb, v_ss, v_sh, dout, lse, D, dq, dk, dv, workspace, B, S, H, Hkv, hd, causal, stream);
  }
  if (dtype == A3V_BF16) {
    hipLaunchKernelGGL(attn_rowdot_kernel<bf16_t>, dim3(B * S * H), dim3(64), 0, ST, (const bf16_t*)out, (const bf16_t*)dout, D, hd);
    hipLaunchKernelGGL(attn_bwd_dq_generic<bf16_t>, dim3(S, H, B), dim3(64), 0, ST, p);
    hipLaunchKernelGGL(attn_bwd_dkv_generic<bf16_t>, dim3(S, Hkv, B), dim3(64), 0, ST, p);
  } else if (dtype == A3V_F32) {
    hipLaunchKernelGGL(attn_rowdot_kernel<float>, dim3(B * S * H), dim3(64), 0, ST, (const float*)out, (const float*)dout, D, hd);
    hipLaunchKernelGGL(attn_bwd_dq_generic<float>, dim3(S, H, B), dim3(64), 0, ST, p);
    hipLaunchKernelGGL(attn_bwd_dkv_generic<float>, dim3(S, Hkv, B), dim3(64), 0, ST, p);
  } else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" __attribute__((visibility("hidden"))) int a3v_attention_bwd_mfma_packed(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb,
                                             int64_t v_ss, int64_t v_sh, const void* dout, const float* lse, const float* D, void* dqkv,
                                             int64_t ld_qkv, const float* cos_sin, int rope_pos0, int B, int S, int H, int Hkv, int hd,
                                             int causal, void* stream);

// a3v_attention_bwd followed by a3v_rope_bwd_pack in one pass (bf16, hd 64 / 128 only: the MFMA kernels store the rotated-back
// gradients straight into the fused-qkv gradient).  D: scratch [B, S, H] floats.
extern "C" int a3v_attention_bwd_packed(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb, int64_t v_ss,
                                        int64_t v_sh, const void* out, int64_t ld_out, const void* dout, const float* lse, float* D, void* dqkv,
                                        int64_t ld_qkv, const float* cos_sin, int rope_pos0, int B, int S, int H, int Hkv, int hd, int causal,
                                        int dtype, void* stream) {
  if (!q || !k || !v || !out || !dout || !lse || !D || !dqkv || !cos_sin || B <= 0 || S <= 0) return A3V_ERR_ARG;
  if (dtype != A3V_BF16) return A3V_ERR_DTYPE;
  if ((hd != 64 && hd != 128) || (H % Hkv) || ld_out < (int64_t)H * hd || ld_out % 8) return A3V_ERR_SHAPE;
  if (((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout)) & 15) != 0) return A3V_ERR_SHAPE;
  const int64_t rows = (int64_t)B * S * H;
  if (hd == 128)
    hipLaunchKernelGGL(attn_rowdot_vec_kernel<128>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, ST, (const bf16_t*)out, (const bf16_t*)dout, D, rows, H, ld_out);
  else
    hipLaunchKernelGGL(attn_rowdot_vec_kernel<64>, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, ST, (const bf16_t*)out, (const bf16_t*)dout, D, rows, H, ld_out);
  A3V_LAUNCH_CHECK();
  return a3v_attention_bwd_mfma_packed(q, k, k_sb, k_sh, v, v_sb, v_ss, v_sh, dout, lse, D, dqkv, ld_qkv, cos_sin, rope_pos0, B, S, H, Hkv, hd, causal,
                                       stream);
}

extern "C" int a3v_embed_bwd(const int64_t* tokens, int64_t ld_tok, const float* dh, float* dtable, int B, int T, int W, int dim,
                             int vocab, void* stream) {
  if (!tokens || !dh || !dtable || B <= 0 || T <= 0) return A3V_ERR_ARG;
  hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(B * (T + W)), dim3(256), 0, ST, tokens, ld_tok, dh, dtable, T, W, dim, vocab);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_embed_bwd_bf16(const int64_t* tokens, int64_t ld_tok, const void* dh, float* dtable, int B, int T, int W, int dim,
                                  int vocab, void* stream) {
  if (!tokens || !dh || !dtable || B <= 0 || T <= 0) return A3V_ERR_ARG;
  hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, dim3(B * (T + W)), dim3(256), 0, ST, tokens, ld_tok, (const bf16_t*)dh, dtable, T, W, dim, vocab);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_rows_sum(const void* src, int64_t ld, const int32_t* row_idx, int n_rows, int dim, float* out, int dtype, void* stream) {
  if (!src || !out || n_rows <= 0 || dim <= 0) return A3V_ERR_ARG;
  dim3 g((dim + 255) / 256, (n_rows + 63) / 64);
  if (dtype == A3V_BF16) hipLaunchKernelGGL(rows_sum_kernel<bf16_t>, g, dim3(256), 0, ST, (const bf16_t*)src, ld, row_idx, n_rows, dim, out);
  else if (dtype == A3V_F32) hipLaunchKernelGGL(rows_sum_kernel<float>, g, dim3(256), 0, ST, (const float*)src, ld, row_idx, n_rows, dim, out);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// ------------------------------------------------------------------ AdamW (decoupled weight decay), one parameter tensor per launch
// torch.optim.AdamW's update (main_finetune.py:138 builds the reference optimizer with betas (0.9, 0.95)):
//   p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// fp32 state, 28 B of HBM traffic per parameter: four 16-B loads and three 16-B stores per lane per iteration, every line touched
// once.  Optionally also writes the bf16 image of the updated parameter (the GEMM operand of the next step).
namespace {
template <int NT>      // NT bit 0: non-temporal loads, bit 1: non-temporal stores (every byte is touched once per step)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n4, int64_t n, float decay, float b1, float b2,
                                                    float step_size, float inv_bc2_sqrt, float eps, bf16_t* __restrict__ img,
                                                    const float* __restrict__ gscale) {
  const float gs = gscale ? *gscale : 1.f;       // clip coefficient (device scalar): g * coef rounded to fp32 = the value grad.mul_(coef) stores
  if (!(gs >= 0.f)) return;                      // negative or NaN: the caller's "this step saw a non-finite loss / gradient norm" flag
                                                 // -> the update is a no-op (masters, moments and bf16 images stay as they were)
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i) : reinterpret_cast<const f32x4*>(p)[i];
    f32x4 gg = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i) : reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) gg[e] = __fmul_rn(gg[e], gs);      // rounded product (never contracted into the fma below)
    f32x4 mm = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m) + i) : reinterpret_cast<const f32x4*>(m)[i];
    f32x4 vv = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i) : reinterpret_cast<const f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pp[e] *= decay;
      mm[e] += (1.f - b1) * (gg[e] - mm[e]);
      vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
      pp[e] -= step_size * mm[e] / (sqrtf(vv[e]) * inv_bc2_sqrt + eps);
    }
    if (NT & 2) {
      __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p) + i);
      __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m) + i);
      __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
    } else {
      reinterpret_cast<f32x4*>(p)[i] = pp;
      reinterpret_cast<f32x4*>(m)[i] = mm;
      reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (img) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(pp[e]);
      if (NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(img) + i);
      else reinterpret_cast<bf16x4*>(img)[i] = o;
    }
  }
  // tail (n % 4 elements), first block only
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    float pp = p[i] * decay;
    const float gg = __fmul_rn(g[i], gs);
    const float mm = m[i] + (1.f - b1) * (gg - m[i]);
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    pp -= step_size * mm / (sqrtf(vv) * inv_bc2_sqrt + eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (img) img[i] = f2bf(pp);
  }
}
}  // namespace

extern "C" int a3v_adamw_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int64_t step, void* bf16_image, const float* grad_scale,
                                void* stream);
extern "C" int a3v_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int64_t step, void* bf16_image, void* stream) {
  return a3v_adamw_scaled(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, bf16_image, nullptr, stream);
}

// a3v_adamw with the gradient multiplied by a DEVICE scalar on the way in (the global-norm clip coefficient of
// util/clip_grad.py:187-193 without the separate grad.mul_ pass over every gradient and without a host read of the norm)
extern "C" int a3v_adamw_scaled(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int64_t step, void* bf16_image, const float* grad_scale,
                                void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return A3V_ERR_ARG;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return A3V_ERR_SHAPE;
  if (bf16_image && ((uintptr_t)bf16_image & 7)) return A3V_ERR_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  // every byte of p / g / m / v / image is touched once per step: non-temporal loads AND stores (default 3) are 6.8 % faster than
  // the default cache policy (271.6 -> 253.2 us on a 45-M-element tensor, tools/adamw_bench.py; either alone: nothing)
  const int nt = A3V_ENV_INT("A3V_ADAMW_NT", 3);               // 0..3 (bit 0 loads, bit 1 stores; A/B runs)
#define A3V_ADAMW_LAUNCH(V) hipLaunchKernelGGL(adamw_kernel<V>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n4, n, decay, \
                     beta1, beta2, step_size, inv_bc2_sqrt, eps, (bf16_t*)bf16_image, grad_scale)
  switch (nt & 3) {
    case 1: A3V_ADAMW_LAUNCH(1); break;
    case 2: A3V_ADAMW_LAUNCH(2); break;
    case 3: A3V_ADAMW_LAUNCH(3); break;
    default: A3V_ADAMW_LAUNCH(0); break;
  }
#undef A3V_ADAMW_LAUNCH
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// ------------------------------------------------------------------ AdamW of a [rows, cols] matrix that also writes the TRANSPOSED bf16 image
// Full fine-tune input gradients run as dX = dY . (W^T)^T on the NT ring kernel (the NN kernel's transpose reads make it 5-10 % slower
// per product): W^T [cols][ldt] has to follow W every step.  A separate a3v_transpose pass re-reads and re-writes 4 B per parameter
// (13.5 GB at 7B, ~6 ms); here the update walks the matrix in 64 x 64 tiles (256-B row segments: every fp32 line still whole), and the
// bf16 tile leaves twice -- as rows of the forward image and, through an 8.3-KiB LDS patch, as 128-B row segments of the transposed one:
// +2 B per parameter.  Same expressions as adamw_kernel, element for element.
namespace {
template <int TR>                                  // tile rows: 128 where rows % 128 == 0 (256-B segments of the transposed image), else 64
__global__ __launch_bounds__(256) void adamw_tile_t_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, int rows, int cols, float decay, float b1, float b2,
                                                           float step_size, float inv_bc2_sqrt, float eps, bf16_t* __restrict__ img,
                                                           bf16_t* __restrict__ imgt, int64_t ldt, const float* __restrict__ gscale) {
  const float gs = gscale ? *gscale : 1.f;
  if (!(gs >= 0.f)) return;
  constexpr int PITCH = TR + 2;                   // bf16 elements per patch row (c-major: patch[c][r]); an odd number of dwords
  __shared__ bf16_t patch2[2][64 * PITCH];        // alternating patches: ONE barrier per tile (a patch is re-written two tiles later, behind the next tile's barrier)
  int flip = 0;
  const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
  const int tiles_c = cols >> 6, ntiles = (rows / TR) * tiles_c;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int r0 = (t / tiles_c) * TR, c0 = (t % tiles_c) << 6;
    bf16_t* patch = patch2[flip];
    flip ^= 1;
#pragma unroll
    for (int ps = 0; ps < TR / 16; ++ps) {
      const int r = ps * 16 + tr;
      const int64_t i = ((int64_t)(r0 + r) * cols + c0 + tc * 4) >> 2;
      f32x4 pp = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
      f32x4 gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) gg[e] = __fmul_rn(gg[e], gs);
      f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m) + i);
      f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pp[e] *= decay;
        mm[e] += (1.f - b1) * (gg[e] - mm[e]);
        vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
        pp[e] -= step_size * mm[e] / (sqrtf(vv[e]) * inv_bc2_sqrt + eps);
      }
      __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p) + i);
      __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m) + i);
      __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = f2bf(pp[e]);
        patch[(tc * 4 + e) * PITCH + r] = o[e];
      }
      if (img) __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(img) + i);
    }
    __syncthreads();
    constexpr int LPR = TR / 4;                   // lanes per transposed row (8-B pieces): 16 or 32
#pragma unroll
    for (int ps = 0; ps < 64 * LPR / 256; ++ps) {
      const int c = ps * (256 / LPR) + tid / LPR, rq = (tid % LPR) * 4;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = patch[c * PITCH + rq + e];
      __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(imgt + (int64_t)(c0 + c) * ldt + r0 + rq));
    }
  }
}
}  // namespace

extern "C" int a3v_adamw_scaled_t(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int rows, int cols, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int64_t step, void* bf16_image,
                                  void* bf16_image_t, int64_t ldt, const float* grad_scale, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !bf16_image_t || rows <= 0 || cols <= 0 || step < 1) return A3V_ERR_ARG;
  if ((rows & 63) || (cols & 63) || ldt < rows || (ldt & 3)) return A3V_ERR_SHAPE;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return A3V_ERR_SHAPE;
  if ((bf16_image && ((uintptr_t)bf16_image & 7)) || ((uintptr_t)bf16_image_t & 7)) return A3V_ERR_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  const int tr = (rows & 127) || A3V_ENV_INT("A3V_ADAMW_T_ROWS", 128) == 64 ? 64 : 128;
  int64_t blocks = (int64_t)(rows / tr) * (cols >> 6);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (tr == 128) hipLaunchKernelGGL(adamw_tile_t_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, rows, cols,
                                    decay, beta1, beta2, step_size, inv_bc2_sqrt, eps, (bf16_t*)bf16_image, (bf16_t*)bf16_image_t, ldt, grad_scale);
  else hipLaunchKernelGGL(adamw_tile_t_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, rows, cols,
                          decay, beta1, beta2, step_size, inv_bc2_sqrt, eps, (bf16_t*)bf16_image, (bf16_t*)bf16_image_t, ldt, grad_scale);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// ------------------------------------------------------------------ adapter gradients: the diagonal blocks of a fused group's dB^T
// dB^T [Rp, N] = t^T . dy is computed for the whole fused group (block-diagonal B: only rows j r .. (j+1) r of the columns of module j
// matter); module j's gradient [n_j, r] += the transpose of its block.  One launch per group (was one torch add_ per module).
namespace {
struct GbScatter { float* dst[4]; int row0[4]; int nj[4]; int n; };
__global__ __launch_bounds__(256) void lora_gb_scatter_kernel(const float* __restrict__ gbt, int64_t ld, int r, GbScatter sc) {
  const int j = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;          // element of module j's [n_j, r] gradient
  if (e >= sc.nj[j] * r) return;
  const int n = e / r, c = e - n * r;
  sc.dst[j][e] += gbt[(int64_t)(j * r + c) * ld + sc.row0[j] + n];
}
}  // namespace

extern "C" int a3v_lora_gb_scatter(const float* gbt, int64_t ld, int r, int n_mods, float* const* dst, const int* row0, const int* nj, void* stream) {
  if (!gbt || !dst || !row0 || !nj || r <= 0 || n_mods <= 0 || n_mods > 4) return A3V_ERR_ARG;
  GbScatter sc{};
  int mx = 0;
  for (int j = 0; j < n_mods; ++j) {
    if (!dst[j] || nj[j] <= 0 || row0[j] < 0) return A3V_ERR_ARG;
    sc.dst[j] = dst[j]; sc.row0[j] = row0[j]; sc.nj[j] = nj[j];
    mx = nj[j] > mx ? nj[j] : mx;
  }
  hipLaunchKernelGGL(lora_gb_scatter_kernel, dim3((mx * r + 255) / 256, n_mods), dim3(256), 0, (hipStream_t)stream, gbt, ld, r, sc);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// ------------------------------------------------------------------ multi-tensor AdamW (the small tensors of a step in ONE launch)
// A LoRA step updates ~520 tensors of 4 K .. 176 K elements (adapters, norm weights): one launch each was 519 launches per step,
// plus one a3v_lora_refresh per adapter to re-write its rows / columns of the fused bf16 group images.  Here blockIdx.y walks a
// device-resident table of tensors that share the hyper-parameters (one torch param group), and the bf16 value of every updated
// element goes to up to TWO strided destinations: element (i, j) of a [rows, cols] parameter -> d1[i s1r + j s1c], d2[i s2r + j s2c]
// (lora_a [r, in]: its rows of A and its columns of A^T; lora_b [n_j, r]: its columns of B and its rows of B^T).  Same arithmetic
// as adamw_kernel, element for element.
namespace {
__global__ __launch_bounds__(256) void adamw_multi_kernel(const a3v_adamw_tensor* __restrict__ tab, float decay, float b1, float b2,
                                                          float step_size, float inv_bc2_sqrt, float eps, const float* __restrict__ gscale) {
  const float gs = gscale ? *gscale : 1.f;
  if (!(gs >= 0.f)) return;
  const a3v_adamw_tensor t = tab[blockIdx.y];
  const int64_t n4 = t.n / 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const bool vec_img = t.d1 && !t.d2 && t.s1c == 1 && t.s1r == t.cols && (t.cols & 3) == 0;      // contiguous same-shape image
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = reinterpret_cast<const f32x4*>(t.p)[i];
    f32x4 gg = reinterpret_cast<const f32x4*>(t.g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) gg[e] = __fmul_rn(gg[e], gs);
    f32x4 mm = reinterpret_cast<const f32x4*>(t.m)[i];
    f32x4 vv = reinterpret_cast<const f32x4*>(t.v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pp[e] *= decay;
      mm[e] += (1.f - b1) * (gg[e] - mm[e]);
      vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
      pp[e] -= step_size * mm[e] / (sqrtf(vv[e]) * inv_bc2_sqrt + eps);
    }
    reinterpret_cast<f32x4*>(t.p)[i] = pp;
    reinterpret_cast<f32x4*>(t.m)[i] = mm;
    reinterpret_cast<f32x4*>(t.v)[i] = vv;
    if (vec_img) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(pp[e]);
      reinterpret_cast<bf16x4*>(t.d1)[i] = o;
    } else if (t.d1 || t.d2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t idx = i * 4 + e, r = idx / t.cols, c = idx - r * t.cols;
        const bf16_t o = f2bf(pp[e]);
        if (t.d1) reinterpret_cast<bf16_t*>(t.d1)[r * t.s1r + c * t.s1c] = o;
        if (t.d2) reinterpret_cast<bf16_t*>(t.d2)[r * t.s2r + c * t.s2c] = o;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(t.n - n4 * 4)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    float pp = t.p[i] * decay;
    const float gg = __fmul_rn(t.g[i], gs);
    const float mm = t.m[i] + (1.f - b1) * (gg - t.m[i]);
    const float vv = b2 * t.v[i] + (1.f - b2) * gg * gg;
    pp -= step_size * mm / (sqrtf(vv) * inv_bc2_sqrt + eps);
    t.p[i] = pp; t.m[i] = mm; t.v[i] = vv;
    const int64_t r = i / t.cols, c = i - r * t.cols;
    if (t.d1) reinterpret_cast<bf16_t*>(t.d1)[r * t.s1r + c * t.s1c] = f2bf(pp);
    if (t.d2) reinterpret_cast<bf16_t*>(t.d2)[r * t.s2r + c * t.s2c] = f2bf(pp);
  }
}
}  // namespace

extern "C" int a3v_adamw_multi(const a3v_adamw_tensor* table_dev, int n_tensors, int64_t max_n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, const float* grad_scale, void* stream) {
  if (!table_dev || n_tensors <= 0 || max_n <= 0 || step < 1) return A3V_ERR_ARG;
  if (n_tensors > 65535) return A3V_ERR_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  int64_t blocks = (max_n / 4 + 255) / 256;
  if (blocks > 64) blocks = 64;            // a tensor's blocks stride over it; with hundreds of tensors in the grid the chip is full anyway
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)blocks, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, table_dev, decay, beta1,
                     beta2, step_size, inv_bc2_sqrt, eps, grad_scale);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// ------------------------------------------------------------------ LoRA step images: one adapter's share of a fused group
// After an optimizer step the bf16 images of a fused adapter group (A [Rp, in] stacked lora_a, B [N, Rp] block-diagonal lora_b,
// and their transposes At [in, Rp], Bt [Rp, Npad] for the backward GEMMs; model/peft.py:40-64 parameters) only change in the r
// rows / columns this adapter owns.  One launch re-writes all four from the fp32 masters (was four strided torch copies).
namespace {
__global__ __launch_bounds__(256) void lora_refresh_kernel(const float* __restrict__ wa, const float* __restrict__ wb, int r, int in_f, int nj,
                                                           bf16_t* __restrict__ A, int64_t lda, bf16_t* __restrict__ At, int64_t ldat,
                                                           bf16_t* __restrict__ B, int64_t ldb, bf16_t* __restrict__ Bt, int64_t ldbt,
                                                           int col0, int row0) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < in_f) {                    // column t of lora_a [r, in]: A[col0 + i][t] and At[t][col0 + i]
    for (int i = 0; i < r; ++i) {
      const bf16_t v = f2bf(wa[(int64_t)i * in_f + t]);
      A[(int64_t)(col0 + i) * lda + t] = v;
      At[(int64_t)t * ldat + col0 + i] = v;
    }
  }
  if (t < nj) {                      // row t of lora_b [nj, r]: B[row0 + t][col0 + i] and Bt[col0 + i][row0 + t]
    for (int i = 0; i < r; ++i) {
      const bf16_t v = f2bf(wb[(int64_t)t * r + i]);
      B[(int64_t)(row0 + t) * ldb + col0 + i] = v;
      Bt[(int64_t)(col0 + i) * ldbt + row0 + t] = v;
    }
  }
}
}  // namespace

extern "C" int a3v_lora_refresh(const float* lora_a, const float* lora_b, int r, int in_f, int nj, void* A, int64_t lda, void* At,
                                int64_t ldat, void* B, int64_t ldb, void* Bt, int64_t ldbt, int col0, int row0, void* stream) {
  if (!lora_a || !lora_b || !A || !At || !B || !Bt || r <= 0 || in_f <= 0 || nj <= 0 || col0 < 0 || row0 < 0) return A3V_ERR_ARG;
  const int n = in_f > nj ? in_f : nj;
  hipLaunchKernelGGL(lora_refresh_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, lora_a, lora_b, r, in_f, nj, (bf16_t*)A, lda, (bf16_t*)At, ldat,
                     (bf16_t*)B, ldb, (bf16_t*)Bt, ldbt, col0, row0);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

