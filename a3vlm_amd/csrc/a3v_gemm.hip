// C[M,N] = epilogue(A[M,K] @ W[N,K]^T) for gfx950.
//
//  * gemm_nt_bf16_kernel : MFMA (v_mfma_f32_16x16x32_bf16) 128x128x64 tile, 4 waves (2x2),
//    each wave 64x64 = 4x4 MFMA tiles; both operands K-contiguous ("NT"), staged
//    HBM -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), two LDS stages.
//    LDS image: [row][8 x 16-B slot], slot' = slot ^ ((row>>1)&7) -- the permutation is
//    applied on the per-lane SOURCE address (LDS-DMA destinations are lane-linear) and
//    again on the ds_read_b128 address; it makes every 16-lane ds_read_b128 group hit
//    16 distinct 16-B slots of the 256-B bank row (conflict-free).
//    MFMA is issued as D = W_frag x A_frag so a lane ends up with 4 CONSECUTIVE n for
//    one m: 8-byte bf16 / 16-byte fp32 row-contiguous stores and a lane-local epilogue
//    (bias, GELU, residual, SwiGLU on interleaved w1/w3 row blocks).
//    blockIdx -> tile map is XCD-aware (8 XCDs, private L2s): each XCD walks a
//    contiguous range of GROUP_M-tall tile groups.
//  * gemm_skinny_bf16_kernel : M <= 16 (decode).  One wave per 16 W-rows x K-slice,
//    W streamed straight from HBM into MFMA operand registers (no LDS round trip),
//    split-K partials reduced by a second tiny kernel that also applies the epilogue.
//  * gemm_nt_f32_kernel : fp32 parity path (plain FMA, 64x64x16 tile).
#include "a3v_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int GROUP_M = 8;

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const void* bias;
  const void* res;
  int64_t lda, ldw, ldc, ldr;
  int M, N, K, epi;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }

// Stage one 128-row x 64-k bf16 tile (16 KiB) with LDS-DMA.  16 chunks of 8 rows; a wave
// issues 4 chunk loads, 1 KiB each: lane -> (row = chunk*8 + lane/8, physical slot = lane%8).
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int64_t ld, int row0,
                                           int last_row, int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = wave * 4 + i;
    const int r = c * 8 + (lane >> 3);
    const int s = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < last_row ? gr : last_row;
    const bf16_t* src = G + (int64_t)gr * ld + k0 + s * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + c * 1024), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile map (bijective for any grid size) ----
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int per_group = GROUP_M * p.tiles_n;
  const int group = bid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = bid - group * per_group;
  const int tm = first_m + in_g % gsz;
  const int tn = in_g / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_tile(p.A, p.lda, m0, p.M - 1, 0, lds, wave, lane);
  stage_tile(p.W, p.ldw, n0, p.N - 1, 0, lds + BM * BK * 2, wave, lane);
  __syncthreads();

  const int frow = lane & 15;            // row inside a 16-row MFMA tile
  const int fsw = (lane >> 1) & 7;       // ((row>>1)&7) -- tile bases are multiples of 16
  const int fks = lane >> 4;             // k-slot 0..3 inside a K=32 slice
  for (int t = 0; t < nk; ++t) {
    char* cur = lds + (t & 1) * STAGE_BYTES;
    if (t + 1 < nk) {
      char* nxt = lds + ((t + 1) & 1) * STAGE_BYTES;
      stage_tile(p.A, p.lda, m0, p.M - 1, (t + 1) * BK, nxt, wave, lane);
      stage_tile(p.W, p.ldw, n0, p.N - 1, (t + 1) * BK, nxt + BM * BK * 2, wave, lane);
    }
    const char* At = cur + (wm * 64 + frow) * 128;
    const char* Wt = cur + BM * BK * 2 + (wn * 64 + frow) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = ((kk * 4 + fks) ^ fsw) << 4;
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(At + i * 16 * 128 + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(Wt + j * 16 * 128 + off);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4 ----
  const int epi = p.epi;
  const int mrow = lane & 15;
  const int ncol = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + mrow;
    if (m >= p.M) continue;
    if (epi & A3V_EPI_SWIGLU) {
      // W rows interleaved in blocks of 16: even 16-block = w1 (gate), odd = w3 (up)
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int n = n0 + wn * 64 + j * 16;        // interleaved row of the gate block
        if (n >= p.N) continue;
        const int oc = (n >> 1) + ncol;             // output column
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = rbf(acc[i][j][r]);
          const float u = rbf(acc[i][j + 1][r]);
          o[r] = f2bf(rbf(silu(g)) * u);
        }
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + oc) = o;
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + ncol;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      if (epi & A3V_EPI_BIAS) {
        const bf16x4 b = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.bias) + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf2f(b[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);   // the bf16 value F.linear returns
      if (epi & A3V_EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(gelu_erf(v[r]));
      } else if (epi & A3V_EPI_QUICKGELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(quick_gelu(v[r]));
      }
      if (epi & A3V_EPI_RES_F32) {
        const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + (int64_t)m * p.ldr + n);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = rr[r] + v[r];
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = o;
        continue;
      }
      if (epi & A3V_EPI_RESIDUAL) {
        const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bf2f(rr[r]) + v[r];
      }
      if (epi & A3V_EPI_OUT_F32) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = v[r];
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = o;
      } else {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Skinny GEMM (M <= 16): HBM-bound weight streaming.
// ------------------------------------------------------------------------------------
struct SkinnyArgs {
  const bf16_t* A;
  const bf16_t* W;
  float* part;
  int64_t lda, ldw;
  int M, N, K, split, kslice, n_tiles;
};

__global__ __launch_bounds__(256) void gemm_skinny_bf16_kernel(SkinnyArgs p) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= p.n_tiles * p.split) return;
  const int nt = wid % p.n_tiles, ks = wid / p.n_tiles;
  const int n0 = nt * 16;
  const int row = lane & 15, kq = (lane >> 4) * 8;
  int wr = n0 + row;
  wr = wr < p.N ? wr : p.N - 1;
  const int ar = row < p.M ? row : p.M - 1;
  const bf16_t* wp = p.W + (int64_t)wr * p.ldw + ks * p.kslice + kq;
  const bf16_t* ap = p.A + (int64_t)ar * p.lda + ks * p.kslice + kq;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 256 <= p.kslice; k += 256) {
    bf16x8 w[8], a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k + q * 32));
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const bf16x8*>(ap + k + q * 32);
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q], a[q], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q + 1], a[q + 1], acc1, 0, 0, 0);
    }
  }
  for (; k < p.kslice; k += 32) {
    const bf16x8 w = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k));
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + k);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc0, 0, 0, 0);
  }
  // D[n = (lane>>4)*4 + r][m = lane&15]
  const int m = lane & 15;
  const int n = n0 + (lane >> 4) * 4;
  if (m < p.M && n < p.N) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = acc0[r] + acc1[r];
    *reinterpret_cast<f32x4*>(p.part + ((int64_t)ks * p.M + m) * p.N + n) = o;
  }
}

struct SkinnyEpiArgs {
  const float* part;
  void* C;
  const void* res;
  int64_t ldc, ldr;
  int M, N, split, epi;
};

__global__ __launch_bounds__(256) void gemm_skinny_epilogue_kernel(SkinnyEpiArgs p) {
  // one thread per 4 output columns
  const int ncols = (p.epi & A3V_EPI_SWIGLU) ? p.N / 2 : p.N;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int per_row = ncols / 4;
  if (idx >= p.M * per_row) return;
  const int m = idx / per_row, c = (idx % per_row) * 4;
  float v[4];
  if (p.epi & A3V_EPI_SWIGLU) {
    const int ng = (c >> 4) * 32 + (c & 15);
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.split; ++s) {
      const float* base = p.part + ((int64_t)s * p.M + m) * p.N;
      g += *reinterpret_cast<const f32x4*>(base + ng);
      u += *reinterpret_cast<const f32x4*>(base + ng + 16);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rbf(silu(rbf(g[r]))) * rbf(u[r]);
  } else {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.split; ++s)
      a += *reinterpret_cast<const f32x4*>(p.part + ((int64_t)s * p.M + m) * p.N + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rbf(a[r]);
    if (p.epi & A3V_EPI_RESIDUAL) {
      const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.res) + (int64_t)m * p.ldr + c);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bf2f(rr[r]);
    }
  }
  if (p.epi & A3V_EPI_OUT_F32) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = v[r];
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + c) = o;
  } else {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + c) = o;
  }
}

// ------------------------------------------------------------------------------------
// fp32 parity GEMM: 64(m) x 64(tile rows of W) x 16, 256 threads, 4x4 outputs per thread.
// A thread owns tile columns {2tx, 2tx+1, 32+2tx, 33+2tx}; with SWIGLU the tile's rows
// 0..31 are gate rows and 32..63 the matching up rows, so the pairing is thread-local.
// ------------------------------------------------------------------------------------
struct GemmF32Args {
  const float* A;
  const float* W;
  float* C;
  const float* bias;
  const float* res;
  int64_t lda, ldw, ldc, ldr;
  int M, N, K, epi;
};

__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(GemmF32Args p) {
  __shared__ float As[16][64 + 4];
  __shared__ float Ws[16][64 + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const bool swi = (p.epi & A3V_EPI_SWIGLU) != 0;
  const int m0 = blockIdx.y * 64;
  const int c0 = blockIdx.x * (swi ? 32 : 64);  // first output column of the tile
  // loader mapping: thread -> (tile row lr, 4 consecutive k)
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  int arow = m0 + lr;
  arow = arow < p.M ? arow : p.M - 1;
  int wrow;
  if (swi) {
    const int c = c0 + (lr & 31);
    wrow = (c >> 4) * 32 + (c & 15) + 16 * (lr >> 5);
  } else {
    wrow = c0 + lr;
  }
  wrow = wrow < p.N ? wrow : p.N - 1;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.A + (int64_t)arow * p.lda + k0 + lk);
    const f32x4 w = *reinterpret_cast<const f32x4*>(p.W + (int64_t)wrow * p.ldw + k0 + lk);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) { As[lk + e][lr] = a[e]; Ws[lk + e][lr] = w[e]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float av[4], wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[k][ty * 4 + i];
      wv[0] = Ws[k][2 * tx]; wv[1] = Ws[k][2 * tx + 1];
      wv[2] = Ws[k][32 + 2 * tx]; wv[3] = Ws[k][33 + 2 * tx];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
    if (swi) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = c0 + 2 * tx + j;
        if (c >= p.N / 2) continue;
        const float g = acc[i][j], u = acc[i][j + 2];
        p.C[(int64_t)m * p.ldc + c] = (g / (1.f + expf(-g))) * u;
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = c0 + (j < 2 ? 2 * tx + j : 32 + 2 * tx + (j - 2));
      if (n >= p.N) continue;
      float v = acc[i][j];
      if (p.epi & A3V_EPI_BIAS) v += p.bias[n];
      if (p.epi & A3V_EPI_GELU) v = gelu_erf(v);
      else if (p.epi & A3V_EPI_QUICKGELU) v = v / (1.f + expf(-1.702f * v));
      if (p.epi & (A3V_EPI_RESIDUAL | A3V_EPI_RES_F32)) v = p.res[(int64_t)m * p.ldr + n] + v;
      p.C[(int64_t)m * p.ldc + n] = v;
    }
  }
}

}  // namespace

extern "C" int a3v_version(void) { return 100; }

extern "C" int a3v_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                           int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                           int epilogue, int dtype, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !C) return A3V_ERR_ARG;
  if ((epilogue & A3V_EPI_BIAS) && !bias) return A3V_ERR_ARG;
  if ((epilogue & (A3V_EPI_RESIDUAL | A3V_EPI_RES_F32)) && !residual) return A3V_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == A3V_F32) {
    if (K % 16 || lda % 4 || ldw % 4) return A3V_ERR_SHAPE;
    if ((epilogue & A3V_EPI_SWIGLU) && (N % 32)) return A3V_ERR_SHAPE;
    GemmF32Args p{(const float*)A, (const float*)W, (float*)C, (const float*)bias, (const float*)residual,
                  lda, ldw, ldc, ldr, M, N, K, epilogue};
    const int ncols = (epilogue & A3V_EPI_SWIGLU) ? N / 2 : N;
    const int tile_c = (epilogue & A3V_EPI_SWIGLU) ? 32 : 64;
    dim3 grid((ncols + tile_c - 1) / tile_c, (M + 63) / 64);
    hipLaunchKernelGGL(gemm_nt_f32_kernel, grid, dim3(256), 0, st, p);
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  if (dtype != A3V_BF16) return A3V_ERR_DTYPE;
  if (K % BK || lda % 8 || ldw % 8 || N % 4 || ldc % 4) return A3V_ERR_SHAPE;
  if ((epilogue & A3V_EPI_SWIGLU) && (N % 32)) return A3V_ERR_SHAPE;
  if ((epilogue & (A3V_EPI_RESIDUAL | A3V_EPI_RES_F32)) && (ldr % 4)) return A3V_ERR_SHAPE;
  GemmArgs p;
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C; p.bias = bias; p.res = residual;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.epi = epilogue;
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
  hipLaunchKernelGGL(gemm_nt_bf16_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_gemm_skinny_split(int M, int N, int K) {
  (void)M;
  const int n_tiles = (N + 15) / 16;
  int want = (2048 + n_tiles - 1) / n_tiles;
  if (want > 8) want = 8;
  if (want < 1) want = 1;
  int split = 1;
  for (int s = want; s >= 1; --s)
    if (K % (32 * s) == 0) { split = s; break; }
  return split;
}

extern "C" int a3v_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                               int M, int N, int K, const void* residual, int64_t ldr, int epilogue,
                               void* partial, void* stream) {
  if (M <= 0 || M > 16 || N <= 0 || K <= 0 || !A || !W || !C || !partial) return A3V_ERR_ARG;
  if (K % 32 || lda % 8 || ldw % 8 || N % 4 || ldc % 4) return A3V_ERR_SHAPE;
  if ((epilogue & A3V_EPI_SWIGLU) && (N % 32)) return A3V_ERR_SHAPE;
  if (epilogue & ~(A3V_EPI_RESIDUAL | A3V_EPI_SWIGLU | A3V_EPI_OUT_F32)) return A3V_ERR_ARG;
  if ((epilogue & A3V_EPI_RESIDUAL) && (!residual || (ldr % 4))) return A3V_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  SkinnyArgs p;
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.part = (float*)partial;
  p.lda = lda; p.ldw = ldw; p.M = M; p.N = N; p.K = K;
  p.split = a3v_gemm_skinny_split(M, N, K);
  p.kslice = K / p.split;
  p.n_tiles = (N + 15) / 16;
  const int waves = p.n_tiles * p.split;
  hipLaunchKernelGGL(gemm_skinny_bf16_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, p);
  A3V_LAUNCH_CHECK();
  SkinnyEpiArgs e{(const float*)partial, C, residual, ldc, ldr, M, N, p.split, epilogue};
  const int ncols = (epilogue & A3V_EPI_SWIGLU) ? N / 2 : N;
  const int threads = M * (ncols / 4);
  hipLaunchKernelGGL(gemm_skinny_epilogue_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, e);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}
