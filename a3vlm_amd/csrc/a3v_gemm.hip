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
//  * gemv_dma_bf16_kernel : M <= 16 (decode), K % 128 == 0.  W streamed once from HBM by coalesced LDS-DMA
//    through wave-private rings, A slice shared per block in LDS, split-K across blocks with an in-kernel
//    last-block fix-up (one launch).  gemm_skinny1_bf16_kernel is the direct-to-VGPR form for other K.
//  * gemm_nt_f32_kernel : fp32 parity path (plain FMA, 64x64x16 tile).
#include "a3v_common.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int BK = 64;
#ifndef A3V_GROUP_M
#define A3V_GROUP_M 8
#endif
constexpr int GROUP_M = A3V_GROUP_M;
constexpr int GEMM_EPI_RAW = 1 << 20;     // internal: fp32 output without the bf16 rounding of the accumulator
constexpr int GEMM_EPI_SCALE = 1 << 22;   // internal: fp8 operands -- accumulator *= sa[m] * sw[n] (per-row scales of A and W) first
constexpr int GEMM_EPI_ROPEKV = 1 << 21;  // internal: fused-qkv epilogue (a3v_gemm_qkv_rope): RoPE on q / k, k -> K cache, v -> V^T cache

// destination of the fused-qkv epilogue: C row m = b*S + s; columns [q heads | k heads | v heads], hd = 1 << hd_shift each
struct RopeKvArgs {
  bf16_t* q_out;         // [rows][ldq], head-major columns (may alias nothing else the GEMM reads)
  bf16_t* k_cache;       // [B][Hkv][Smax][hd]
  bf16_t* vt_cache;      // [B][Hkv][hd][Smax]
  const float* cos_sin;  // [pos][hd/2][2]
  bf16_t* v_rows;        // optional [rows][ldv]: v also token-major (the attention backward reads it that way)
  int64_t ldq, ldv;
  int S, H, Hkv, hd_shift, Smax, start_pos, rope_pos0, m_off;
};

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const void* bias;
  const void* res;
  int64_t lda, ldw, ldc, ldr;
  int M, N, K, epi;
  int tiles_m, tiles_n;
  int dbg;   // ablation switches for tuning runs (0 in production): 1 = no DMA in the k-loop, 2 = no ds_read in the k-loop
  int skew;  // ring kernel: XCD x starts x * skew cycles late, so the eight XCDs' store bursts do not hit HBM together
  int slow_epi;   // 1: interior tiles also take the general epilogue (A3V_GEMM_FAST_EPI=0; equality tests and A/B runs)
  int nt_store;   // fast epilogue forms: non-temporal stores of the output tile (A3V_GEMM_NT_STORE, read per launch)
  float* sumsq;   // fp32 outputs (weight gradients): slot (tile * 8 + wave) <- sum of squares of the values this wave stored (NULL: off)
  int64_t c_split;   // split-K (128x128 kernel, gridDim.y slices): byte stride between the slices' output planes
  RopeKvArgs rk;     // GEMM_EPI_ROPEKV only
  const float* sa;   // GEMM_EPI_SCALE: per-row dequantisation scales of A [M] and W [N]
  const float* sw;
  int xmap;          // ring kernel: 1 = every round of gridDim.x tiles is cut into eight runs, one per XCD (see tile_of)
  unsigned* xsync;   // ring kernel, A3V_GEMM_LOCKSTEP=1: eight zeroed counters; the blocks of an XCD start each tile round together
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// erf-GELU for the bf16 epilogues: Abramowitz-Stegun 7.1.26 (|erf error| < 1.5e-7, far below the bf16 rounding that follows)
// with one v_rcp and one v_exp instead of libm's branchy erff (the c_fc epilogue of the ViT cost 25 us of an 85-us GEMM)
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erf_abs = 1.f - poly * __expf(-z * z);
  return 0.5f * x * (1.f + copysignf(erf_abs, x));
}
// x * sigmoid(a x) with v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU instructions): the result is rounded to
// bf16 right after, and the SwiGLU epilogue of a 256x256 tile was VALU-bound on it (13 k cycles per tile)
__device__ __forceinline__ float quick_gelu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// Stage one ROWS-row x 64-k bf16 tile with LDS-DMA.  Chunks of 8 rows (1 KiB = one wave
// instruction); lane -> (row = chunk*8 + lane/8, physical 16-B slot = lane%8).
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int64_t ld, int row0,
                                           int last_row, int k0, char* lds_tile, int wave, int lane) {
  constexpr int PER_WAVE = ROWS / 8 / NWAVES;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int c = wave * PER_WAVE + i;
    const int r = c * 8 + (lane >> 3);
    const int s = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < last_row ? gr : last_row;
    const bf16_t* src = G + (int64_t)gr * ld + k0 + s * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + c * 1024), 16, 0, 0);
  }
}

// Weight-gradient GEMMs can hand the global-norm clip its sum of squares for free: every wave adds up the squares of the fp32
// values it stores and writes ONE partial to slot (tile_m * tiles_n + tile_n) * 8 + (wave's position in the tile) -- a layout that
// depends only on the output coordinates, not on the block order (the caller zeroes the slots once per step and sums them).
__device__ __forceinline__ void sumsq_flush(const GemmArgs& p, float ss, int mbase, int nbase, int lane) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (lane == 0) p.sumsq[(((int64_t)(mbase >> 8) * p.tiles_n + (nbase >> 8)) << 3) + (((mbase >> 7) & 1) << 2) + ((nbase >> 6) & 3)] = ss;
}

// Compiler-level ordering point between the two halves of a wave-private LDS transpose (the hardware executes one wave's LDS
// instructions in order; what has to be prevented is the compiler moving a read of one vector type across writes of another)
#define LDS_ORDER() asm volatile("" ::: "memory")

// Fast forms of the epilogue for wave tiles that lie INSIDE C (16 TM x 64 outputs, no ragged edge) and the output kinds of the
// decoder's big linears: plain bf16, bf16 residual, fp32 residual stream, fp32 (raw or rounded, with or without accumulation),
// SwiGLU.  Same arithmetic as the general form below, bit for bit; what differs is the cost.  The general form tests the
// epilogue flags and both bounds per 16 x 16 tile and forms every address with a 64-bit multiply: its instruction stream for
// one 256 x 256 tile measured 21-25 k cycles where the stores alone need 3.5 k (one CU) to 8 k (all CUs storing at once,
// tools/ubench/stores.hip).  Here the kind is decided once per wave tile, addresses advance by a constant stride, residuals
// are read in the layout they are stored in (whole 128 / 256-byte row segments), and every value passes through the wave's
// private 4-KiB LDS patch so that it leaves as 16 bytes per lane.  Returns false when the tile is not eligible.
// SET selects which forms an instantiation carries: EPI_SET_COMMON the kinds of the decoder's linears, EPI_SET_PRE additionally
// bias / activation in front of them (the ViT), EPI_SET_ROPE the fused-qkv form alone.  The 256x256 kernels hold 128 accumulator
// registers and ~110 more across the k-loop; compiled together, the forms' peak pushed the k-loop's invariants into scratch
// (100 dwords per lane, 5-20 % of every kind's speed), so each big-tile kernel is instantiated per set and picked by the host.
constexpr int EPI_SET_COMMON = 1, EPI_SET_PRE = 2, EPI_SET_ROPE = 4, EPI_SET_SUMSQ = 8;   // SUMSQ: GemmArgs::sumsq honoured (weight-gradient kernels only)
template <int TM, int TN, int SET>
__device__ __forceinline__ bool gemm_epilogue_fast(f32x4 (&acc)[TM][TN], const GemmArgs& p, int mbase, int nbase, int lane, char* stage) {
  if constexpr (TN != 4 || ((TM % 4) != 0 && TM != 6)) {          // (TM = 6: the 192-row ring tiles)
    return false;
  } else {
    // opaque copies: everything below is recomputed per tile AFTER the k-loop instead of being hoisted out of the persistent tile
    // loop and kept (or spilled) across the k-loop, whose 244 live VGPRs leave no room
    int kind = p.epi & (0xffff | GEMM_EPI_RAW | GEMM_EPI_ROPEKV | GEMM_EPI_SCALE);
    int64_t ldc_ = p.ldc, ldr_ = p.ldr;
    uintptr_t c_ = reinterpret_cast<uintptr_t>(p.C), r_ = reinterpret_cast<uintptr_t>(p.res);
    asm volatile("" : "+s"(kind), "+s"(ldc_), "+s"(ldr_), "+s"(c_), "+s"(r_));
    if (!stage || p.slow_epi || mbase + TM * 16 > p.M || nbase + 64 > p.N || (c_ & 15)) return false;
    const int mrow = lane & 15, g = lane >> 4;
    float ss = 0.f;                                      // p.sumsq: squares of the fp32 values this wave stores
    const bool nts = p.nt_store != 0;                      // output tiles are written once and read by a later kernel: streaming stores
    auto st_c = [&](auto* ptr, auto val) { if (nts) __builtin_nontemporal_store(val, ptr); else *ptr = val; };
    const int pre = (SET & EPI_SET_PRE) ? kind & (A3V_EPI_BIAS | A3V_EPI_GELU | A3V_EPI_QUICKGELU) : 0;   // applied in the accumulator layout
    if (pre) kind &= ~pre;
    if constexpr ((SET & EPI_SET_COMMON) != 0) {
    if (kind == 0 || kind == A3V_EPI_RESIDUAL || kind == A3V_EPI_RES_F32 || kind == A3V_EPI_SWIGLU_BWD) {
      // bf16 staging: chunk = two 16-row tiles x 64 columns = 32 rows x 128 B; 8-byte slot s of row r at slot s ^ (r & 14),
      // read back as 16-byte pairs: pair q of row r from physical pair q ^ ((r >> 1) & 7) (see the general form)
      const bool f32 = kind == A3V_EPI_RES_F32;
      if (f32 ? ((ldc_ & 3) || (ldr_ & 3) || (r_ & 15)) : (ldc_ & 7)) return false;       // (before anything touches the accumulators)
      if ((kind == A3V_EPI_RESIDUAL || kind == A3V_EPI_SWIGLU_BWD) && ((ldr_ & 7) || (r_ & 15))) return false;
      if (kind == A3V_EPI_SWIGLU_BWD && (p.N & 7)) return false;
      uintptr_t b_ = reinterpret_cast<uintptr_t>(p.bias);
      asm volatile("" : "+s"(b_));
      if ((pre & A3V_EPI_BIAS) && (b_ & 7)) return false;
      // bias / activation in front (the ViT's linears): y = act(bf16(acc + bias)), each step rounded to bf16 as the general form
      // does; applied as the values are staged (a separate pass over the 128 accumulators kept the k-loop's state in scratch)
      float bv[4][4];
      if constexpr ((SET & EPI_SET_PRE) != 0) {
        if (pre) {
          if (f32) return false;                               // (fp32 stream with bias: not a shape of the path; general form)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bf16x4 b4 = {};
            if (pre & A3V_EPI_BIAS) b4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(b_) + nbase + j * 16 + g * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][e] = (pre & A3V_EPI_BIAS) ? bf2f(b4[e]) : 0.f;
          }
        }
      }
      auto pre_value = [&](float v, int j, int e) {
        v = rbf(v + bv[j][e]);
        if (pre & A3V_EPI_GELU) v = rbf(gelu_erf_fast(v));
        else if (pre & A3V_EPI_QUICKGELU) v = rbf(quick_gelu(v));
        return v;
      };
      const int l3 = lane >> 3, q = lane & 7;
      char* const wr = stage + mrow * 128;
      const int wx = mrow & 14;
      const char* const rd = stage + l3 * 128;
      const int rx = l3 >> 1;
      auto put = [&](int ic) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[2 * ic + ii][j][e];
              if constexpr ((SET & EPI_SET_PRE) != 0) { if (pre) v = pre_value(v, j, e); }
              o[e] = f2bf(v);
            }
            *reinterpret_cast<bf16x4*>(wr + ii * 2048 + (((j * 4 + g) ^ wx) << 3)) = o;
          }
        LDS_ORDER();        // the reads below are of another vector type: keep the compiler from moving them across these writes
      };
      auto get = [&](int it) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(rd + it * 1024 + ((q ^ rx ^ ((it & 1) << 2)) << 4));
        LDS_ORDER();        // ... and the next chunk's writes behind this read
        return v;
      };
      if (!f32) {
        bf16_t* cp = reinterpret_cast<bf16_t*>(c_) + (int64_t)(mbase + l3) * ldc_ + nbase + q * 8;
        const int64_t cstep = 8 * ldc_;
        if (kind == 0) {
#pragma unroll
          for (int ic = 0; ic < TM / 2; ++ic) {
            put(ic);
#pragma unroll
            for (int it = 0; it < 4; ++it) { st_c(reinterpret_cast<bf16x8*>(cp), get(it)); cp += cstep; }
          }
        } else if (kind == A3V_EPI_SWIGLU_BWD) {
          // the tile is d(act): gate / up read from `res` (gu [M, 2 N]) as whole 16-byte row segments, d(gate) -> C[., n], d(up) -> C[., N + n]
          const bf16_t* gp = reinterpret_cast<const bf16_t*>(r_) + (int64_t)(mbase + l3) * ldr_ + nbase + q * 8;
          const int64_t rstep = 8 * ldr_;
          int un = p.N;                                     // columns between the gate and the up half (gu and C alike)
          asm volatile("" : "+s"(un));
#pragma unroll
          for (int ic = 0; ic < TM / 2; ++ic) {             // one 32-row chunk's gate + up in flight at a time (32 VGPRs)
            bf16x8 gg[4], uu[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              gg[it] = *reinterpret_cast<const bf16x8*>(gp);
              uu[it] = *reinterpret_cast<const bf16x8*>(gp + un);
              gp += rstep;
            }
            put(ic);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const bf16x8 v = get(it);
              bf16x8 og, ou;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float dg, du;
                swiglu_bwd_pair(bf2f(gg[it][e]), bf2f(uu[it][e]), bf2f(v[e]), dg, du);
                og[e] = f2bf(dg);
                ou[e] = f2bf(du);
              }
              st_c(reinterpret_cast<bf16x8*>(cp), og);
              st_c(reinterpret_cast<bf16x8*>(cp + un), ou);
              cp += cstep;
            }
          }
        } else {
          const bf16_t* rp = reinterpret_cast<const bf16_t*>(r_) + (int64_t)(mbase + l3) * ldr_ + nbase + q * 8;
          const int64_t rstep = 8 * ldr_;
#pragma unroll
          for (int half = 0; half < (TM + 2) / 4; ++half) { // the loads of two 32-row chunks in flight at a time (32 VGPRs); TM = 6: 2 + 1
            bf16x8 rr[2][4];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int it = 0; it < 4; ++it) {
                if (half * 2 + c < TM / 2) { rr[c][it] = *reinterpret_cast<const bf16x8*>(rp); rp += rstep; }
              }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              if (half * 2 + c >= TM / 2) continue;
              put(half * 2 + c);
#pragma unroll
              for (int it = 0; it < 4; ++it) {
                const bf16x8 v = get(it);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(rr[c][it][e]) + bf2f(v[e]));
                st_c(reinterpret_cast<bf16x8*>(cp), o);
                cp += cstep;
              }
            }
          }
        }
      } else {
        // fp32 residual stream: out = res + bf16(acc); a lane's 8 values of a row are 32 contiguous bytes of res / C
        float* cp = reinterpret_cast<float*>(c_) + (int64_t)(mbase + l3) * ldc_ + nbase + q * 8;
        const float* rp = reinterpret_cast<const float*>(r_) + (int64_t)(mbase + l3) * ldr_ + nbase + q * 8;
        const int64_t cstep = 8 * ldc_, rstep = 8 * ldr_;
#pragma unroll
        for (int ic = 0; ic < TM / 2; ++ic) {               // one 32-row chunk's residual in flight at a time (32 VGPRs)
          f32x4 rr[4][2];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            rr[it][0] = *reinterpret_cast<const f32x4*>(rp);
            rr[it][1] = *reinterpret_cast<const f32x4*>(rp + 4);
            rp += rstep;
          }
          put(ic);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const bf16x8 v = get(it);
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = rr[it][0][e] + bf2f(v[e]); o1[e] = rr[it][1][e] + bf2f(v[4 + e]); }
            if constexpr ((SET & EPI_SET_SUMSQ) != 0) {
              if (p.sumsq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ss = fmaf(o0[e], o0[e], fmaf(o1[e], o1[e], ss));
              }
            }
            st_c(reinterpret_cast<f32x4*>(cp), o0);
            st_c(reinterpret_cast<f32x4*>(cp + 4), o1);
            cp += cstep;
          }
        }
        if constexpr ((SET & EPI_SET_SUMSQ) != 0) { if (p.sumsq) sumsq_flush(p, ss, mbase, nbase, lane); }
      }
      return true;
    }
    if (kind == (A3V_EPI_RES_F32 | GEMM_EPI_RAW) || kind == (A3V_EPI_OUT_F32 | GEMM_EPI_RAW) || kind == A3V_EPI_OUT_F32) {
      // fp32 staging: chunk = one 16-row tile x 64 columns = 16 rows x 256 B, 16-byte slot s of row r at slot s ^ r
      const bool accum = (kind & A3V_EPI_RES_F32) != 0, raw = (kind & GEMM_EPI_RAW) != 0;
      if ((ldc_ & 3) || (accum && ((ldr_ & 3) || (r_ & 15)))) return false;
      const int q = lane & 15;
      char* const wr = stage + mrow * 256;
      const char* const rd = stage + g * 256;
      float* cp = reinterpret_cast<float*>(c_) + (int64_t)(mbase + g) * ldc_ + nbase + q * 4;
      const float* rp = accum ? reinterpret_cast<const float*>(r_) + (int64_t)(mbase + g) * ldr_ + nbase + q * 4 : nullptr;
      const int64_t cstep = 4 * ldc_, rstep = 4 * ldr_;
#pragma unroll
      for (int half = 0; half < TM / 2; ++half) {           // two 16-row tiles' worth of the accumulated-into rows in flight (32 VGPRs)
        f32x4 rr[2][4];
        if (accum) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int it = 0; it < 4; ++it) { rr[i][it] = *reinterpret_cast<const f32x4*>(rp); rp += rstep; }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[half * 2 + i][j];
            if (!raw) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
            }
            *reinterpret_cast<f32x4*>(wr + (((j * 4 + g) ^ mrow) << 4)) = v;
          }
          LDS_ORDER();
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            f32x4 v = *reinterpret_cast<const f32x4*>(rd + it * 1024 + ((q ^ (it * 4 + g)) << 4));
            LDS_ORDER();
            if (accum) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rr[i][it][e] + v[e];
            }
            if constexpr ((SET & EPI_SET_SUMSQ) != 0) {
              if (p.sumsq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ss = fmaf(v[e], v[e], ss);
              }
            }
            st_c(reinterpret_cast<f32x4*>(cp), v);
            cp += cstep;
          }
        }
      }
      if constexpr ((SET & EPI_SET_SUMSQ) != 0) { if (p.sumsq) sumsq_flush(p, ss, mbase, nbase, lane); }
      return true;
    }
    if (kind == A3V_EPI_SWIGLU && (TM % 4) == 0) {
      // 64 interleaved columns -> 32 output columns = 64 B per row.  chunk = four 16-row tiles = 64 rows x 64 B; 8-byte slot
      // s = 4 jp + g of row r at slot s ^ (((r >> 2) & 3) << 1) (rows r, r+4, r+8, r+12 share banks: the XOR separates them and keeps
      // 16-byte pairs together); read back as pairs, 16 rows x 64 B per store instruction
      if (ldc_ & 7) return false;
      const int rr_ = lane >> 2, qq = lane & 3;
      char* const wr = stage + mrow * 64;
      const int wx = ((mrow >> 2) & 3) << 1;
      const char* const rd = stage + rr_ * 64 + ((qq ^ ((rr_ >> 2) & 3)) << 4);
      bf16_t* cp = reinterpret_cast<bf16_t*>(c_) + (int64_t)(mbase + rr_) * ldc_ + (nbase >> 1) + qq * 8;
      const int64_t cstep = 16 * ldc_;
#pragma unroll
      for (int ic = 0; ic < TM / 4; ++ic) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float gt = rbf(acc[4 * ic + ii][2 * jp][e]);
              const float up = rbf(acc[4 * ic + ii][2 * jp + 1][e]);
              o[e] = f2bf(rbf(silu(gt)) * up);
            }
            *reinterpret_cast<bf16x4*>(wr + ii * 1024 + (((jp * 4 + g) ^ wx) << 3)) = o;
          }
        LDS_ORDER();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(rd + it * 1024);
          LDS_ORDER();
          st_c(reinterpret_cast<bf16x8*>(cp), v);
          cp += cstep;
        }
      }
      return true;
    }
    }
    if constexpr ((SET & EPI_SET_ROPE) != 0) {
    if (kind == GEMM_EPI_ROPEKV && !pre) {
      // Fused qkv epilogue.  The wave's 64 columns lie inside ONE head slot (hd >= 64), so the whole wave tile is q, k or v.
      //   q, k : rotated in the accumulator layout (cos / sin rows read there), then through the bf16 transpose: q as whole
      //          128-byte row segments of q_out, k as 128-byte segments of the token's K-cache row;
      //   v    : token-major rows of v_rows (training) through the same transpose, and the V^T cache through a TRANSPOSING
      //          stage ([64 d][32 tokens]): 16-byte stores of 8 consecutive tokens of one d (2-byte aligned: legal here,
      //          tools/ubench/unaligned.hip) instead of one 2-byte store per element.
      RopeKvArgs k = p.rk;                                   // opaque per-tile copy: none of it is hoisted out of the tile loop
      asm volatile("" : "+s"(k.q_out), "+s"(k.k_cache), "+s"(k.vt_cache), "+s"(k.cos_sin), "+s"(k.v_rows), "+s"(k.ldq), "+s"(k.ldv));
      asm volatile("" : "+s"(k.S), "+s"(k.H), "+s"(k.Hkv), "+s"(k.hd_shift), "+s"(k.Smax), "+s"(k.start_pos), "+s"(k.rope_pos0), "+s"(k.m_off));
      const int hd = 1 << k.hd_shift;
      if (k.hd_shift < 6 || k.S < 32 || (k.ldq & 7) || (reinterpret_cast<uintptr_t>(k.q_out) & 15) || (reinterpret_cast<uintptr_t>(k.k_cache) & 15) ||
          (k.v_rows && ((k.ldv & 7) || (reinterpret_cast<uintptr_t>(k.v_rows) & 15))))
        return false;
      int slot = nbase >> k.hd_shift, d0 = nbase & (hd - 1), mg0 = mbase + k.m_off, S_ = k.S;
      asm volatile("" : "+s"(slot), "+s"(d0), "+s"(mg0), "+s"(S_));
      const int l3 = lane >> 3, q = lane & 7;
      char* const wr = stage + mrow * 128;
      const int wx = mrow & 14;
      const char* const rd = stage + l3 * 128;
      const int rx = l3 >> 1;
      auto get = [&](int it) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(rd + it * 1024 + ((q ^ rx ^ ((it & 1) << 2)) << 4));
        LDS_ORDER();
        return v;
      };
      auto put_plain = [&](int ic) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[2 * ic + ii][j][e]);
            *reinterpret_cast<bf16x4*>(wr + ii * 2048 + (((j * 4 + g) ^ wx) << 3)) = o;
          }
        LDS_ORDER();
      };
      if (slot < k.H + k.Hkv) {
        // q / k: rotated as the chunk is staged.  The cos / sin rows of a 32-row chunk (8 x 16 B per lane) are requested one chunk
        // ahead -- before the previous chunk's stores, so the loads never queue behind them -- and nothing of the tile is copied.
        int sqr = (mg0 + mrow) % S_;                        // token position of this lane's row in tile i; +16 per tile
        const float* csb = k.cos_sin + ((d0 + g * 4) >> 1) * 2;
        f32x4 csA[2][4], csB[2][4];
        auto load_cs = [&](f32x4 (&cs)[2][4]) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            asm volatile("" : "+v"(sqr));
            const float* csr = csb + (((int64_t)(k.rope_pos0 + sqr)) << k.hd_shift);      // (pos << (hd_shift - 1)) * 2 floats
#pragma unroll
#ifdef ROPE_NO_CS      // timing build (wrong results): no cos / sin loads -- they are 21 of the epilogue's 28 us per launch (r04g_decode_fixup_and_rope_epilogue.txt)
            for (int j = 0; j < 4; ++j) cs[ii][j] = f32x4{1.f, 0.f, 1.f, 0.f};
            (void)csr;
#else
            for (int j = 0; j < 4; ++j) cs[ii][j] = *reinterpret_cast<const f32x4*>(csr + j * 16);
#endif
            sqr += 16;
            if (sqr >= S_) sqr -= S_;
          }
        };
        auto put_rot = [&](int ic, const f32x4 (&cs)[2][4]) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int i = 2 * ic + ii;
              const float v0 = rbf(acc[i][j][0]), v1 = rbf(acc[i][j][1]), v2 = rbf(acc[i][j][2]), v3 = rbf(acc[i][j][3]);
              bf16x4 o;
              o[0] = f2bf(v0 * cs[ii][j][0] - v1 * cs[ii][j][1]);
              o[1] = f2bf(v0 * cs[ii][j][1] + v1 * cs[ii][j][0]);
              o[2] = f2bf(v2 * cs[ii][j][2] - v3 * cs[ii][j][3]);
              o[3] = f2bf(v2 * cs[ii][j][3] + v3 * cs[ii][j][2]);
              *reinterpret_cast<bf16x4*>(wr + ii * 2048 + (((j * 4 + g) ^ wx) << 3)) = o;
            }
          LDS_ORDER();
        };
        const bool isq = slot < k.H;
        bf16_t* cp = k.q_out + nbase + (int64_t)(mg0 + l3) * k.ldq + q * 8;
        const int64_t cstep = 8 * k.ldq;
        int b = (mg0 + l3) / S_, sq = (mg0 + l3) - b * S_;
        bf16_t* const kb = k.k_cache + d0 + q * 8;
        const int hk = slot - k.H;
        load_cs(csA);
#pragma unroll
        for (int ic = 0; ic < TM / 2; ++ic) {
          if (ic + 1 < TM / 2) { if (ic & 1) load_cs(csA); else load_cs(csB); }
          if (ic & 1) put_rot(ic, csB); else put_rot(ic, csA);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            if (isq) {
              asm volatile("" : "+v"(cp));
              st_c(reinterpret_cast<bf16x8*>(cp), get(it));
              cp += cstep;
            } else {
              asm volatile("" : "+v"(sq), "+v"(b));
              *reinterpret_cast<bf16x8*>(kb + ((((int64_t)b * k.Hkv + hk) * k.Smax + k.start_pos + sq) << k.hd_shift)) = get(it);
              sq += 8;
              if (sq >= S_) { sq -= S_; ++b; }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        return true;
      }
      if (k.v_rows) {
        bf16_t* cp = k.v_rows + (nbase - ((k.H + k.Hkv) << k.hd_shift)) + (int64_t)(mg0 + l3) * k.ldv + q * 8;
        const int64_t cstep = 8 * k.ldv;
#pragma unroll
        for (int ic = 0; ic < TM / 2; ++ic) {
          put_plain(ic);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            asm volatile("" : "+v"(cp));
            st_c(reinterpret_cast<bf16x8*>(cp), get(it));
            cp += cstep;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // V^T cache: [64 d][32 tokens] bf16 per chunk; the 16-byte token group c of row d sits at group c ^ ((d >> 2) & 3)
      {
        const int hv = slot - k.H - k.Hkv;
        int b0 = mg0 / S_, sq0 = mg0 - b0 * S_;             // first token of the chunk (wave-uniform)
        const int dr = lane >> 2, c = lane & 3;
#pragma unroll
        for (int ic = 0; ic < TM / 2; ++ic) {
          asm volatile("" : "+s"(sq0), "+s"(b0));
          if (sq0 + 32 <= S_) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  *reinterpret_cast<bf16_t*>(stage + (j * 16 + g * 4 + e) * 64 + (((ii * 2 + (mrow >> 3)) ^ g) << 4) + (mrow & 7) * 2) =
                      f2bf(acc[2 * ic + ii][j][e]);
            LDS_ORDER();
            bf16_t* const vb = k.vt_cache + ((((int64_t)b0 * k.Hkv + hv) << k.hd_shift) + d0) * k.Smax + k.start_pos + sq0 + c * 8;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int d = it * 16 + dr;
              const bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + d * 64 + ((c ^ ((d >> 2) & 3)) << 4));
              LDS_ORDER();
              bf16_t* const dst = vb + (int64_t)d * k.Smax;               // 2-byte-aligned 16-byte store: the compiler would split it
              // (s_nop 1 inside the statement: hipcc pads no hazards of an asm store, and a 16-byte store reads its data registers a
              //  couple of states after issue -- without it the next instruction the compiler places here may overwrite them first;
              //  seen as zeros / stale values in a few V^T rows per launch once a re-schedule put a register write right behind)
              asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            }
          } else {
            // the chunk's 32 tokens straddle two batch elements: element-wise, as the general form does
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
              const int t = sq0 + ii * 16 + mrow;
              const int bb = t >= S_ ? b0 + 1 : b0, ss = t >= S_ ? t - S_ : t;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                bf16_t* dst = k.vt_cache + ((((int64_t)bb * k.Hkv + hv) << k.hd_shift) + d0 + j * 16 + g * 4) * k.Smax + k.start_pos + ss;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(int64_t)e * k.Smax] = f2bf(acc[2 * ic + ii][j][e]);
              }
            }
          }
          sq0 += 32;
          if (sq0 >= S_) { sq0 -= S_; ++b0; }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      return true;
    }
    }
    return false;
  }
}

// Epilogue shared by the tile kernels.  The lane holds, for each (i, j) MFMA tile,
// C[m = mbase + 16 i + (lane&15)][n = nbase + 16 j + 4 (lane>>4) + 0..3].
// `stage`: optional wave-private 4-KiB LDS scratch for gemm_epilogue_fast (tiles inside C, the common output kinds); everything
// else takes the general form below, whose stores are 8 bytes per lane scattered over 16 rows.
template <int TM, int TN, bool F8 = false, int SET = EPI_SET_COMMON>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[TM][TN], const GemmArgs& p, int mbase, int nbase, int lane, char* stage = nullptr) {
  // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4 ----
  if constexpr (!F8) {
    if (gemm_epilogue_fast<TM, TN, SET>(acc, p, mbase, nbase, lane, stage)) return;
  }
  const int epi = p.epi;
  const int mrow = lane & 15;
  const int ncol = (lane >> 4) * 4;
  RopeKvArgs rk_ = p.rk;                                   // opaque per-tile copy (see gemm_epilogue_fast): not hoisted out of a persistent tile loop
  if (epi & GEMM_EPI_ROPEKV) {
    asm volatile("" : "+s"(rk_.q_out), "+s"(rk_.k_cache), "+s"(rk_.vt_cache), "+s"(rk_.cos_sin), "+s"(rk_.v_rows), "+s"(rk_.ldq), "+s"(rk_.ldv));
    asm volatile("" : "+s"(rk_.S), "+s"(rk_.H), "+s"(rk_.Hkv), "+s"(rk_.hd_shift), "+s"(rk_.Smax), "+s"(rk_.start_pos), "+s"(rk_.rope_pos0), "+s"(rk_.m_off));
  }
  if constexpr (F8) {
    // fp8 operands: the accumulator holds sum_k qa[m][k] qw[n][k]; the value of the product is that times sa[m] sw[n].
    // All scale loads are issued before the first store of the tile: a load behind a store would wait for the store to drain.
    f32x4 swv[TN];
    float sam[TM];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nbase + j * 16 + ncol;
      swv[j] = n < p.N ? *reinterpret_cast<const f32x4*>(p.sw + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = mbase + i * 16 + mrow;
      sam[i] = m < p.M ? p.sa[m] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] *= sam[i] * swv[j][r];
  }
  if (epi & A3V_EPI_SWIGLU) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = mbase + i * 16 + mrow;
      if (m >= p.M) continue;
      // W rows interleaved in blocks of 16: even 16-block = w1 (gate), odd = w3 (up)
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const int n = nbase + j * 16;        // interleaved row of the gate block
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
    }
    return;
  }
  if (epi & A3V_EPI_SWIGLU_BWD) {
    // the product is d(act): d(gate) / d(up) from the forward's gate / up (res = gu [M, 2 N]); loads of half the tile before its stores
    constexpr int HM = TM >= 2 ? TM / 2 : 1, NH = TM / HM;
#pragma unroll
    for (int half = 0; half < NH; ++half) {
      bf16x4 gg[HM][TN], uu[HM][TN];
#pragma unroll
      for (int ih = 0; ih < HM; ++ih) {
        const int m = mbase + (half * HM + ih) * 16 + mrow;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = nbase + j * 16 + ncol;
          gg[ih][j] = bf16x4{};
          uu[ih][j] = bf16x4{};
          if (m < p.M && n < p.N) {
            const bf16_t* gr = reinterpret_cast<const bf16_t*>(p.res) + (int64_t)m * p.ldr + n;
            gg[ih][j] = *reinterpret_cast<const bf16x4*>(gr);
            uu[ih][j] = *reinterpret_cast<const bf16x4*>(gr + p.N);
          }
        }
      }
#pragma unroll
      for (int ih = 0; ih < HM; ++ih) {
        const int i = half * HM + ih, m = mbase + i * 16 + mrow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = nbase + j * 16 + ncol;
          if (n >= p.N) continue;
          bf16x4 og, ou;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float dg, du;
            swiglu_bwd_pair(bf2f(gg[ih][j][r]), bf2f(uu[ih][j][r]), rbf(acc[i][j][r]), dg, du);
            og[r] = f2bf(dg);
            ou[r] = f2bf(du);
          }
          bf16_t* cr = reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n;
          *reinterpret_cast<bf16x4*>(cr) = og;
          *reinterpret_cast<bf16x4*>(cr + p.N) = ou;
        }
      }
    }
    return;
  }
  // Phase 1 -- values: bias, the bf16 rounding F.linear applies, activation, rotary embedding, residual; everything that LOADS
  // (bias, cos/sin rows, residual) happens here, before the tile's first store: on this ISA a load issued behind stores can
  // only be waited for together with them, and at the end of a tile round the whole chip's store burst takes microseconds to
  // drain.  The loads of half the tile (TM/2 x TN vectors) are in flight at a time; the results overwrite the accumulators.
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = half * (TM / 2); i < (half + 1) * (TM / 2); ++i) {
      const int m = mbase + i * 16 + mrow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nbase + j * 16 + ncol;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        if (epi & A3V_EPI_BIAS) {
          const bf16x4 b = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.bias) + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bf2f(b[r]);
        }
        if (!(epi & GEMM_EPI_RAW)) {      // split-K planes keep the raw fp32 partial sums (rounded once by the reduce pass)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);   // the bf16 value F.linear returns
        }
        if (epi & GEMM_EPI_ROPEKV) {
          // v[] = the bf16 qkv values of token (b, s), columns n..n+3 of head slot n >> hd_shift: rotate the two (even, odd)
          // pairs of q / k by the token's position (LLM/llama_ens5.py:123-135 apply_rotary_emb); v passes through
          const RopeKvArgs& k = rk_;
          if (epi & A3V_EPI_RESIDUAL) {      // an additive term of the projection (the LoRA branch, model/peft.py:89-95): the
            // reference adds it to the bf16 linear output and rounds, before the rotation
            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rbf(bf2f(rr[r]) + v[r]);
          }
          const int mg = m + k.m_off;
          const int sq = mg - (mg / k.S) * k.S;
          const int slot = n >> k.hd_shift, d = n & ((1 << k.hd_shift) - 1);
          if (slot < k.H + k.Hkv) {
            const f32x4 cs = *reinterpret_cast<const f32x4*>(k.cos_sin + (((int64_t)(k.rope_pos0 + sq) << (k.hd_shift - 1)) + (d >> 1)) * 2);
            const float o0 = v[0] * cs[0] - v[1] * cs[1], o1 = v[0] * cs[1] + v[1] * cs[0];
            const float o2 = v[2] * cs[2] - v[3] * cs[3], o3 = v[2] * cs[3] + v[3] * cs[2];
            v[0] = o0; v[1] = o1; v[2] = o2; v[3] = o3;
          }
        } else {
          if (epi & A3V_EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rbf(gelu_erf_fast(v[r]));
          } else if (epi & A3V_EPI_QUICKGELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rbf(quick_gelu(v[r]));
          }
          if (epi & A3V_EPI_RES_F32) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rr[r] + v[r];
          } else if (epi & A3V_EPI_RESIDUAL) {
            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = bf2f(rr[r]) + v[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = v[r];
      }
    }
  }
  // Phase 2 -- stores only (ragged tiles and the rarer output kinds: 8 bytes per lane straight from the accumulator layout)
  float gss = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mbase + i * 16 + mrow;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nbase + j * 16 + ncol;
      if (n >= p.N) continue;
      if (epi & GEMM_EPI_ROPEKV) {
        // q in place of the qkv row, k into the K cache, v transposed into the V^T cache (and token-major into v_rows)
        const RopeKvArgs& k = rk_;
        const int mg = m + k.m_off;
        const int b = mg / k.S, sq = mg - b * k.S;
        const int slot = n >> k.hd_shift, d = n & ((1 << k.hd_shift) - 1);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[i][j][r]);
        if (slot < k.H + k.Hkv) {
          bf16_t* dst = slot < k.H ? k.q_out + (int64_t)mg * k.ldq + n
                                   : k.k_cache + ((((int64_t)b * k.Hkv + (slot - k.H)) * k.Smax + k.start_pos + sq) << k.hd_shift) + d;
          *reinterpret_cast<bf16x4*>(dst) = o;
        } else {
          bf16_t* dst = k.vt_cache + ((((int64_t)b * k.Hkv + (slot - k.H - k.Hkv)) << k.hd_shift) + d) * k.Smax + k.start_pos + sq;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(int64_t)r * k.Smax] = o[r];
          if (k.v_rows) *reinterpret_cast<bf16x4*>(k.v_rows + (int64_t)mg * k.ldv + (n - ((k.H + k.Hkv) << k.hd_shift))) = o;
        }
        continue;
      }
      if (epi & (A3V_EPI_RES_F32 | A3V_EPI_OUT_F32)) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = acc[i][j];
        if constexpr ((SET & EPI_SET_SUMSQ) != 0) {
          if (p.sumsq) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gss = fmaf(acc[i][j][r], acc[i][j][r], gss);
          }
        }
      } else {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[i][j][r]);
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n) = o;
      }
    }
  }
  if constexpr ((SET & EPI_SET_SUMSQ) != 0) { if (p.sumsq) sumsq_flush(p, gss, mbase, nbase, lane); }
}

// TBM x TBN block tile, WAVES_M x WAVES_N waves, each wave (TBM/WAVES_M) x (TBN/WAVES_N).
template <int TBM, int TBN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_nt_bf16_kernel(GemmArgs p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WTM = TBM / WAVES_M, WTN = TBN / WAVES_N;   // wave tile
  constexpr int TM = WTM / 16, TN = WTN / 16;               // MFMA tiles per wave
  constexpr int STAGE = (TBM + TBN) * BK * 2;
  static_assert(TN % 2 == 0, "SwiGLU pairing needs an even number of 16-column tiles per wave");
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

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
  const int m0 = tm * TBM, n0 = tn * TBN;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // split-K: slice z of gridDim.y takes the k-tiles [z nk/S, (z+1) nk/S) and writes its own output plane (summed by
  // splitk_reduce_kernel); skinny problems (N or M = 64) otherwise run as a few blocks with a long serial K loop
  int nk = p.K / BK;
  if (gridDim.y > 1) {
    const int z = blockIdx.y, S = gridDim.y;
    const int t0 = (int)(((int64_t)z * nk) / S), t1 = (int)(((int64_t)(z + 1) * nk) / S);
    p.A += (int64_t)t0 * BK;
    p.W += (int64_t)t0 * BK;
    p.C = reinterpret_cast<char*>(p.C) + (int64_t)z * p.c_split;
    nk = t1 - t0;
  }
  if (nk > 0) {
    stage_tile<TBM, NW>(p.A, p.lda, m0, p.M - 1, 0, lds, wave, lane);
    stage_tile<TBN, NW>(p.W, p.ldw, n0, p.N - 1, 0, lds + TBM * BK * 2, wave, lane);
  }
  __syncthreads();

  const int frow = lane & 15;            // row inside a 16-row MFMA tile
  const int fsw = (lane >> 1) & 7;       // ((row>>1)&7) -- tile bases are multiples of 16
  const int fks = lane >> 4;             // k-slot 0..3 inside a K=32 slice
  for (int t = 0; t < nk; ++t) {
    char* cur = lds + (t & 1) * STAGE;
    if (t + 1 < nk) {
      char* nxt = lds + ((t + 1) & 1) * STAGE;
      stage_tile<TBM, NW>(p.A, p.lda, m0, p.M - 1, (t + 1) * BK, nxt, wave, lane);
      stage_tile<TBN, NW>(p.W, p.ldw, n0, p.N - 1, (t + 1) * BK, nxt + TBM * BK * 2, wave, lane);
    }
    const char* At = cur + (wm * WTM + frow) * 128;
    const char* Wt = cur + TBM * BK * 2 + (wn * WTN + frow) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = ((kk * 4 + fks) ^ fsw) << 4;
      bf16x8 af[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(Wt + j * 16 * 128 + off);
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(At + i * 16 * 128 + off);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // the loop's last __syncthreads() is behind every read of the stage buffers: a private 4 KiB per wave for whole-row stores
  gemm_epilogue<TM, TN, false, (TM > 4 ? EPI_SET_COMMON : EPI_SET_COMMON | EPI_SET_PRE)>(acc, p, m0 + wm * WTM, n0 + wn * WTN, lane, nk > 0 ? lds + wave * 4096 : nullptr);
}

// Adapter-sized products (N <= 64: t = x A^T, dt = dy B) stream one long operand once; what bounds them is how many of its bytes
// the chip has outstanding, not MFMA.  The two-stage kernel above keeps ONE 8-KiB k-tile of the streamed operand in flight per
// block (548 blocks x 8 KiB = 4.4 MB over the chip: 3.6 TB/s measured at 8728 x 64 x 22016).  Same 64 x 64 tile, same fragment
// reads and MFMA order -- bit-identical planes -- with NST stages: NST - 1 k-tiles stay in flight across the block barrier
// (counted waits: 4 LDS-DMA instructions per thread and stage), one barrier per k-tile.
template <int N> __device__ __forceinline__ void vm_wait_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// TBM rows of the streamed operand per block (4 waves x TBM / 4 rows), NST stages of (TBM + 64) x 128 B.
template <int TBM, int NST>
__global__ __launch_bounds__(256) void gemm_nt_skinny_kernel(GemmArgs p) {
  constexpr int TBN = 64, NW = 4, TN = 4, TM = TBM / 64, WTM = TBM / 4;
  constexpr int STAGE = (TBM + TBN) * BK * 2;            // 16 / 24 / 40 KiB
  constexpr int PER = TBM / 32 + 2;                      // LDS-DMA instructions per thread and stage
  static_assert(PER * (NST - 2) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(1024))) char lds_dyn[];
  char* const lds = lds_dyn;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * TBM;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int nk = p.K / BK;
  {
    const int z = blockIdx.y, S = gridDim.y;
    const int t0 = (int)(((int64_t)z * nk) / S), t1 = (int)(((int64_t)(z + 1) * nk) / S);
    p.A += (int64_t)t0 * BK;
    p.W += (int64_t)t0 * BK;
    p.C = reinterpret_cast<char*>(p.C) + (int64_t)z * p.c_split;
    nk = t1 - t0;
  }
  auto stage = [&](int t) __attribute__((always_inline)) {
    char* dst = lds + (t % NST) * STAGE;
    stage_tile<TBM, NW>(p.A, p.lda, m0, p.M - 1, t * BK, dst, wave, lane);
    stage_tile<TBN, NW>(p.W, p.ldw, 0, p.N - 1, t * BK, dst + TBM * BK * 2, wave, lane);
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) stage(s);
  const int frow = lane & 15, fsw = (lane >> 1) & 7, fks = lane >> 4;
  for (int t = 0; t < nk; ++t) {
    // k-tile t has landed when at most the pieces of the k-tiles issued after it are outstanding
    const int ahead = min(NST - 2, nk - 1 - t);
    if (ahead >= 3) vm_wait_imm<3 * PER>();
    else if (ahead == 2) vm_wait_imm<2 * PER>();
    else if (ahead == 1) vm_wait_imm<PER>();
    else vm_wait_imm<0>();
    // every wave's pieces of k-tile t are in, every wave is done with k-tile t - 1.  A bare s_barrier: __syncthreads() carries a fence
    // that hipcc lowers to vmcnt(0), i.e. it would drain the k-tiles this loop exists to keep in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + NST - 1 < nk) stage(t + NST - 1);            // into the buffer of k-tile t - 1
    const char* cur = lds + (t % NST) * STAGE;
    const char* At = cur + (wave * WTM + frow) * 128;
    const char* Wt = cur + TBM * BK * 2 + frow * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int off = ((kk * 4 + fks) ^ fsw) << 4;
      bf16x8 af[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(Wt + j * 16 * 128 + off);
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(At + i * 16 * 128 + off);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the epilogue's 4-KiB patches lie over the stage buffers
  gemm_epilogue<TM, TN, false, EPI_SET_COMMON | EPI_SET_PRE>(acc, p, m0 + wave * WTM, 0, lane, nk > 0 ? lds + wave * 4096 : nullptr);
}

// Epilogue for v_mfma_f32_32x32x16 accumulators (D = W_frag x A_frag): for tile (i, j) the lane holds
// C[m = mbase + 32 i + (lane&31)][n = nbase + 32 j + 8 g + 4 (lane>>5) + 0..3], g = reg/4.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue32(f32x16 (&acc)[TM][TN], const GemmArgs& p, int mbase, int nbase, int lane) {
  const int epi = p.epi;
  const int mrow = lane & 31;
  const int hh4 = (lane >> 5) * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mbase + i * 32 + mrow;
    if (m >= p.M) continue;
    if (epi & A3V_EPI_SWIGLU) {
      // 16-row interleave: tile rows 0..15 = gate, 16..31 = up  ->  reg groups g (gate) / g+2 (up)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nbase + j * 32;
        if (n >= p.N) continue;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int oc = (n >> 1) + 8 * g + hh4;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gt = rbf(acc[i][j][4 * g + r]);
            const float up = rbf(acc[i][j][4 * (g + 2) + r]);
            o[r] = f2bf(rbf(silu(gt)) * up);
          }
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + oc) = o;
        }
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nbase + j * 32 + 8 * g + hh4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r];
        if (epi & A3V_EPI_BIAS) {
          const bf16x4 b = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.bias) + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bf2f(b[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
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
// 256x256x64 "ping-pong" kernel: 8 waves = 2 groups of 4 (wr = wave/4 owns 128 rows of A; the
// waves w and w+4 share a SIMD).  Each group alternates a LOAD interval (24 ds_read_b128 for a
// whole K-tile of its 128x64 wave tile, + its share of the LDS-DMA for a later K-tile) and an
// MFMA interval (64 v_mfma_f32_16x16x32_bf16 from registers); the groups run ONE barrier apart,
// so on every SIMD one wave feeds the matrix pipe while its partner reads LDS / issues DMA.
//
//   interval:   1      2      3      4      5      6
//   group 0:   L(0)   M(0)   L(1)   M(1)   L(2)   M(2) ...      L(t): reads buf[t&1]
//   group 1:    -     L(0)   M(0)   L(1)   M(1)   L(2) ...
//   DMA issue:               t=2           t=3           ...    tile t+2 -> buf[t&1]: by group 0 in
//   DMA wait :                      t=2           t=3    ...    L(t+1), by group 1 in M(t) (same interval:
//                                                                 the first one after BOTH groups read tile t)
// Ordering rules used (guide: "read a staged buffer one phase AFTER the wait that retires it"):
// every wave waits vmcnt(0) for its own DMA pieces in the interval after it issued them, then the
// interval barrier publishes them; readers start one barrier later.  A wave retires its own
// ds_reads (lgkmcnt(0)) BEFORE the barrier that ends its LOAD interval, so a buffer is only
// re-staged after every read of it has returned.
#ifdef PP_PIN
#define PP_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define PP_SB() do {} while (0)
#endif
#define A3V_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define A3V_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define A3V_BARRIER()                      \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
  } while (0)

#ifdef A3V_EXPERIMENTS   // measured and not dispatched: built only with `make EXPERIMENTS=1`
template <int DBG, int SCHED>
__global__ __launch_bounds__(512) void gemm_nt_bf16_pp_kernel(GemmArgs p) {
  constexpr int TBM = 256, TBN = 256, NW = 8, WTM = 128, WTN = 64, TM = 8, TN = 4;
  constexpr int STAGE = (TBM + TBN) * BK * 2;   // 64 KiB
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // Persistent over tiles: block b takes tiles b, b + gridDim.x, ... of the XCD-aware order below (gridDim.x is a multiple of 8,
  // so a block's tiles all map to its own XCD).  The next tile's first two k-stages are issued BEFORE this tile's epilogue:
  // the store burst of a tile round (every CU writes its 128 KiB at once) drains while the next operands are already in flight,
  // and there is no workgroup launch between tiles.
  const int ntiles = p.tiles_m * p.tiles_n;
  auto tile_of = [&](int vb, int& tm0, int& tn0) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int per_group = GROUP_M * p.tiles_n;
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int in_g = bid - group * per_group;
    tm0 = (first_m + in_g % gsz) * TBM;
    tn0 = (in_g / gsz) * TBN;
  };
  int m0, n0, sm0, sn0;     // tile being computed / tile being staged
  tile_of(blockIdx.x, m0, n0);
  sm0 = m0; sn0 = n0;
  f32x4 acc[TM][TN];

  const int nk = p.K / BK;
  // LDS-DMA through buffer descriptors: rows past M / N are out of range and read as zero (no
  // clamping VALU), addresses are {SGPR descriptor, 32-bit VGPR offset, SGPR k-offset}.
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
  // per-lane byte offset inside an 8-row chunk: row = lane/8, 16-B slot = (lane%8) ^ ((chunk*4 + lane/16) & 7)
  const unsigned lr = lane >> 3;
  unsigned voA[2], voW[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const unsigned sl = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
    voA[par] = (unsigned)((lr * p.lda + sl * 8) * 2);
    voW[par] = (unsigned)((lr * p.ldw + sl * 8) * 2);
  }
  // one 1-KiB DMA piece: c in [0,8): 0..3 -> A chunks, 4..7 -> W chunks of this wave
  auto stage_piece = [&](int t, int c) {
    const int ch = wave * 4 + (c & 3);
    char* dst = lds + (t & 1) * STAGE + (c >= 4 ? TBM * BK * 2 : 0) + ch * 1024;
    if (c < 4) {
      // (row-chunk + k) offset is wave-uniform and changes with t: an SGPR sum added per piece, so
      // nothing per-piece stays live in VGPRs across the loop
      const unsigned so = (unsigned)(((int64_t)(sm0 + ch * 8) * p.lda + t * BK) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voA[c & 1] + so, 0, 0, 0);
    } else {
      const unsigned so = (unsigned)(((int64_t)(sn0 + ch * 8) * p.ldw + t * BK) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, voW[c & 1] + so, 0, 0, 0);
    }
  };
  auto stage = [&](int t) {
#pragma unroll
    for (int c = 0; c < 8; ++c) stage_piece(t, c);
  };
  stage(0);
  if (nk > 1) stage(1);

  const int frow = lane & 15, fsw = (lane >> 1) & 7, fks = lane >> 4;
  const int off0 = ((0 * 4 + fks) ^ fsw) << 4, off1 = ((1 * 4 + fks) ^ fsw) << 4;
  const int a_base = (wr * WTM + frow) * 128;
  const int w_base = TBM * BK * 2 + (wc * WTN + frow) * 128;
  bf16x8 af0[TM], af1[TM], wf0[TN], wf1[TN];

#define PP_READ_FRAGS(cur)                                                                           \
  do {                                                                                               \
    const char* At_ = (cur) + a_base;                                                                \
    const char* Wt_ = (cur) + w_base;                                                                \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                 \
      wf0[j] = *reinterpret_cast<const bf16x8*>(Wt_ + j * 2048 + off0);                              \
      wf1[j] = *reinterpret_cast<const bf16x8*>(Wt_ + j * 2048 + off1);                              \
    }                                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                 \
      af0[i] = *reinterpret_cast<const bf16x8*>(At_ + i * 2048 + off0);                              \
      af1[i] = *reinterpret_cast<const bf16x8*>(At_ + i * 2048 + off1);                              \
    }                                                                                                \
  } while (0)

#define PP_MFMA_ALL()                                                                                \
  do {                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[j], af0[i], acc[i][j], 0, 0, 0);     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[j], af1[i], acc[i][j], 0, 0, 0);     \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  } while (0)

  constexpr bool do_dma = !(DBG & 1), do_rd = !(DBG & 2);
  // DBG == 4: cycle stamps (s_memtime) of block 0, waves 0 and 4, into the buffer passed as `bias`
  unsigned long long* stamps = (DBG == 4 && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0)
                                   ? (unsigned long long*)p.bias + (wave ? 1 : 0) * 64 * 8 : nullptr;
#define PP_STAMP(t, k) do { if (DBG == 4 && stamps && (t) < 64) stamps[(t) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  for (int vb = blockIdx.x;;) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  A3V_WAIT_VM0();
  A3V_BARRIER();
  if constexpr (SCHED == 1) {
    // Schedule 1: both groups issue the DMA of tile t+1 in their own LOAD interval (the MFMA intervals
    // carry no VMEM issue) and wait for it at the end of that interval.  That only works if the DMA
    // hits in L2, so every wave also TOUCHES one 128-B line of tile t+3 per K-tile (one dword per lane:
    // waves 0-3 cover the 256 A rows, waves 4-7 the 256 W rows) -- a software L2 prefetch that takes
    // the fabric (MALL/HBM) latency out of the DMA's completion time.  The touch is the youngest VMEM
    // op at the interval's wait, hence vmcnt(1).
    const unsigned pf_vo = (unsigned)((int64_t)((wave & 3) * 64 + lane) * (wr ? p.ldw : p.lda) * 2
                                      + (int64_t)(wr ? n0 : m0) * (wr ? p.ldw : p.lda) * 2);
    const auto pf_rs = wr ? rsW : rsA;
    int pf = 0;
    if (wr == 1) A3V_BARRIER();
    for (int t = 0; t < nk; ++t) {
      PP_STAMP(t, 0);
      asm volatile("" ::"v"(pf));            // retire the previous touch (compiler-inserted vmcnt)
      PP_READ_FRAGS(lds + (t & 1) * STAGE);
      if (do_dma && t >= 1 && t + 1 < nk) stage(t + 1);
      {
        const int tp = min(t + 3, nk - 1);
        pf = __builtin_amdgcn_raw_buffer_load_b32(pf_rs, pf_vo, tp * BK * 2, 0);
      }
      A3V_WAIT_LGKM0();
      PP_STAMP(t, 1);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      PP_STAMP(t, 2);
      A3V_BARRIER();
      PP_STAMP(t, 3);
      PP_MFMA_ALL();
      PP_STAMP(t, 4);
      A3V_BARRIER();
      PP_STAMP(t, 5);
    }
    asm volatile("" ::"v"(pf));
    if (wr == 0) A3V_BARRIER();
  } else if (wr == 0) {
    for (int t = 0; t < nk; ++t) {
      PP_STAMP(t, 0);
      if (do_rd || t == 0) PP_READ_FRAGS(lds + (t & 1) * STAGE);
      if (do_dma && t >= 1 && t + 1 < nk) stage(t + 1);
      A3V_WAIT_LGKM0();
      PP_STAMP(t, 1);
      A3V_BARRIER();
      PP_STAMP(t, 2);
      PP_MFMA_ALL();
      PP_STAMP(t, 3);
      A3V_WAIT_VM0();
      PP_STAMP(t, 4);
      A3V_BARRIER();
      PP_STAMP(t, 5);
    }
    A3V_BARRIER();
  } else {
    A3V_BARRIER();
    for (int t = 0; t < nk; ++t) {
      PP_STAMP(t, 0);
      if (do_rd || t == 0) PP_READ_FRAGS(lds + (t & 1) * STAGE);
      A3V_WAIT_LGKM0();
      PP_STAMP(t, 1);
      A3V_WAIT_VM0();
      PP_STAMP(t, 2);
      A3V_BARRIER();
      PP_STAMP(t, 3);
#ifndef PP_INTERLEAVE
      if (do_dma && t + 2 < nk) stage(t + 2);
      PP_MFMA_ALL();
#else
      {
        // DMA pieces spread through the MFMA stream: one piece per 8 MFMAs
        const bool has = do_dma && (t + 2 < nk);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (has) stage_piece(t + 2, c);
          PP_SB();
          if (c < 4) {
#pragma unroll
            for (int i = 2 * c; i < 2 * c + 2; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[j], af0[i], acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 2 * (c - 4); i < 2 * (c - 4) + 2; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[j], af1[i], acc[i][j], 0, 0, 0);
          }
          PP_SB();
        }
        __builtin_amdgcn_s_setprio(0);
      }
#endif
      PP_STAMP(t, 4);
      A3V_BARRIER();
      PP_STAMP(t, 5);
    }
  }
  // every read of both stage buffers is behind the last barrier: stage the next tile now, store this one after
  const int nb = vb + (int)gridDim.x;
  if (nb < ntiles) {
    tile_of(nb, sm0, sn0);
    stage(0);
    if (nk > 1) stage(1);
  }
  {
    int lane_e = lane;                       // opaque copy: keeps the epilogue's per-lane address arithmetic from being hoisted
    asm volatile("" : "+v"(lane_e));         // out of the tile loop (and held in VGPRs across the k-loop)
    gemm_epilogue<TM, TN>(acc, p, m0 + wr * WTM, n0 + wc * WTN, lane_e);
  }
  if (nb >= ntiles) break;
  vb = nb; m0 = sm0; n0 = sn0;
  }
#undef PP_STAMP
#undef PP_READ_FRAGS
#undef PP_MFMA_ALL
}
#endif  // A3V_EXPERIMENTS

typedef __attribute__((ext_vector_type(4))) int i32x4r;
// ------------------------------------------------------------------------------------
// "Ring" form of the ping-pong kernel: the same tile, fragments, MFMA stream and epilogue, but the LDS is cut into three
// rings that together use all 160 KiB of the CU, so every LDS-DMA piece has THREE OR FOUR intervals (1.5 - 2 K-tile periods)
// to land instead of two.  The two-stage kernel above waits ~2600 cycles for a stage that it issued one K-tile period
// (2176 cycles of MFMA) earlier: the k-loop runs at the DMA's completion latency, not at the matrix pipe's rate.
//
//   A_top ring : rows   0..127 of the A tile (only group 0 reads them), 2 slots x 16 KiB   slot(t) = t & 1
//   A_bot ring : rows 128..255 of the A tile (only group 1 reads them), 2 slots x 16 KiB   slot(t) = t & 1
//   W ring     : the 256 W rows (both groups read them),                3 slots x 32 KiB   slot(t) = t % 3
//
//   interval I:   2t             2t+1            2t+2            2t+3
//   group 0:      L(t)           M(t)            L(t+1)          M(t+1)
//   group 1:      M(t-1)         L(t)            M(t)            L(t+1)
//   freed at the barrier that ENDS the interval:
//                 A_top(t)       A_bot(t), W(t)
//   issued in L (the wave's partner on the SIMD is in its MFMA interval), 8 pieces per wave per K-tile:
//     group 0 in L(t)  : A_bot(t+1) -> needed I = 2t+3 (3 intervals),   W rows 0..127 of tile t+2   -> needed I = 2t+4 (4)
//     group 1 in L(t)  : W rows 128..255 of tile t+2 and A_top(t+2)     -> needed I = 2t+4 (3 intervals)
//   counted waits (loads retire in order; a wave only ever waits for its OWN pieces, the barrier publishes them):
//     group 0, end of L(t): vmcnt(12) = A_bot(t), the first 4 pieces of its previous burst, has landed (group 1 reads it next)
//              end of M(t): vmcnt(8)  = the rest of that burst (W half of tile t+1) has landed
//     group 1, end of L(t): vmcnt(8)  = its previous burst (W half and A_top of tile t+1) has landed
//   The last two K-tiles issue shorter bursts, so their counts shrink accordingly (the switch below).
// ------------------------------------------------------------------------------------
// EARLY: the barrier that ends an MFMA interval is executed EARLY tile-rows before the interval's last MFMA.  Nothing after it
// needs the barrier (the tail MFMAs read registers only), and the partner wave on the SIMD -- released by the same barrier --
// starts its own MFMA stream while this wave is still feeding the pipe: no matrix-pipe bubble at the hand-over.
template <int DBG, bool M32, int EARLY, bool STAGED = true, int SET = EPI_SET_COMMON, bool LW = false, bool LATEW = true, bool CONT = true, int TBM_ = 256, bool F8 = false>   // F8 (round 5): OCP e4m3fn operands (the W8A8 prefill): a K-tile is still 128 BYTES per row -- 128 elements -- so rings, DMA pieces, swizzle and fragment reads are the bf16 kernel's; the two 16-byte fragments of a lane feed ONE v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) instead of two 16x16x32 bf16 MFMAs, and the per-row scales sa[m] sw[n] are applied to the accumulators in front of the SAME staged epilogues (the two-stage fp8 kernel it replaces was one tile per block with the general epilogue: 21-25 k cycles of an 83 k-cycle tile); TBM_ (round 5): 192 = a 192 x 256 tile (six 16-row MFMA tiles per wave, 12-KiB A halves, 7 instead of 8 DMA pieces per wave and LOAD interval): M = 8728 x N = 4096 is 2.875 rounds of these instead of 2.19 rounds of 256 x 256 -- the rows beyond whole rounds cost no split-K planes; CONT (product; r03): the DMA stream runs on ACROSS the block's tiles -- the last two LOAD intervals of a tile fetch K-tiles 0 / 1 of the block's next tile into the ring slots they would have used anyway, so there is no prologue burst, no pipeline drain / refill and no block-wide barrier between tiles (see the boundary notes in the body); LATEW (product; r03 A/B +1 % on every shape, bit-equal): group 0 waits for its W pieces of tile t+1 at the TOP of L(t+1) instead of between its last MFMA of M(t) and the barrier that hands the matrix pipe over; SET: which fast epilogue forms (gemm_epilogue_fast); M32: v_mfma_f32_32x32x16_bf16 (4x2 tiles per wave) instead of 16x16x32 (8x4)
__global__ __launch_bounds__(512) void gemm_nt_bf16_ring_kernel(GemmArgs p) {
  static_assert(TBM_ == 256 || (TBM_ == 192 && !M32 && LATEW && !LW), "the 192-row form exists for the product schedule only");
  static_assert(!F8 || (!M32 && EARLY == 0 && DBG == 0), "fp8 operands: 16x16x128 MFMA, product schedule");
  constexpr int EB = F8 ? 1 : 2;                        // bytes per operand element
  constexpr int BKE = 128 / EB;                         // elements per K-tile (128 bytes per row either way)
  constexpr int TBM = TBM_, TBN = 256, WTM = TBM / 2, WTN = 64, TM = M32 ? 4 : WTM / 16, TN = M32 ? 2 : 4;
  constexpr int AH = WTM * BK * 2;                      // 16 KiB (12 KiB): one group's half of an A K-tile
  constexpr int APW = WTM / 32;                         // 1-KiB pieces (8 rows) of an A half per wave of a group: 4 (3)
  constexpr int BURST = APW + 4;                        // pieces per wave and LOAD interval: A half + W half
  constexpr int WT = TBN * BK * 2;                      // 32 KiB: a W K-tile
  constexpr int ATOP = 0, ABOT = 2 * AH, WB = 4 * AH;   // ring bases
  __shared__ __attribute__((aligned(1024))) char lds[4 * AH + 3 * WT];   // 163840 B = all of the CU's LDS (147456 B with 192-row tiles)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int ntiles = p.tiles_m * p.tiles_n;
  auto tile_of = [&](int vb, int& tm0, int& tn0) {
    int bid;
    const int G = (int)gridDim.x, full = ntiles / G, rnd = vb / G;
    if (p.xmap & 4) {
      // super-tile walk (host sets it when G == 256 and both tile counts are multiples of 16): a round is ONE 16 x 16 block of tiles
      // (32 operand panels from HBM instead of the 40 of an 8 x 32 strip), XCD x takes its 8 x 4 sub-block (still 12 panels per L2)
      const int x = vb & 7, i = (vb & 255) >> 3, nsc = p.tiles_n >> 4;
      tm0 = ((rnd / nsc) * 16 + (x >> 2) * 8 + (i & 7)) * TBM;
      tn0 = ((rnd % nsc) * 16 + (x & 3) * 4 + (i >> 3)) * TBN;
      return;
    }
    if (p.xmap && rnd < full) {
      // round-major: the eight XCDs work on the SAME run of G consecutive tiles (8 tile rows x G/8 columns), XCD x on columns
      // [x G/64, (x+1) G/64) of it -- the A panels of the row group are shared by all XCDs through the Infinity Cache instead of
      // every XCD streaming its own eight panels from HBM
      bid = rnd * G + (vb & 7) * (G >> 3) + ((vb - rnd * G) >> 3);
    } else {
      const int base = p.xmap ? full * G : 0, R = ntiles - base, v = vb - base;
      const int xcd = v & 7, q = R >> 3, r = R & 7;
      bid = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    const int per_group = GROUP_M * p.tiles_n;
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int in_g = bid - group * per_group;
    tm0 = (first_m + in_g % gsz) * TBM;
    const int tn = in_g / gsz;
    // xmap & 2: odd row groups walk the columns backwards, so a new group starts on the W panels the last one ended on (still in the
    // Infinity Cache) instead of on the ones evicted longest ago
    tn0 = (((p.xmap & 2) && (group & 1)) ? p.tiles_n - 1 - tn : tn) * TBN;
  };
  int m0, n0, sm0, sn0;     // tile being computed / tile being staged
  tile_of(blockIdx.x, m0, n0);
  sm0 = m0; sn0 = n0;
  typedef typename std::conditional<M32, f32x16, f32x4>::type acc_t;
  acc_t acc[TM][TN];

  // split-K (gridDim.y slices; the tail rows of the hybrid dispatch): slice z takes the k-tiles [z nk/S, (z+1) nk/S) and writes
  // its own raw fp32 plane (summed, rounded and finished by splitk_epilogue_kernel)
  if (gridDim.y > 1) {
    const int nk_all = p.K / BKE, z = blockIdx.y, S = gridDim.y;
    const int t0 = (int)(((int64_t)z * nk_all) / S), t1 = (int)(((int64_t)(z + 1) * nk_all) / S);
    p.A = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(p.A) + (int64_t)t0 * 128);
    p.W = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(p.W) + (int64_t)t0 * 128);
    p.K = (t1 - t0) * BKE;
    p.C = reinterpret_cast<char*>(p.C) + (int64_t)z * p.c_split;
  }
  const int nk = p.K / BKE;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * EB), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)(((int64_t)(p.N - 1) * p.ldw + p.K) * EB), 0x00020000);
  // per-lane byte offset inside an 8-row chunk: row = lane/8, 16-B slot = (lane%8) ^ ((chunk*4 + lane/16) & 7)
  const unsigned lr = lane >> 3;
  unsigned voA[2], voW[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const unsigned sl = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
    voA[par] = (unsigned)(lr * p.lda * EB + sl * 16);
    voW[par] = (unsigned)(lr * p.ldw * EB + sl * 16);
  }
  // one 1-KiB piece = 8 rows x 128 B; `row` = first row inside the A (W) tile, `par` = its chunk index & 1 (swizzle key)
  // (`tm0` / `tn0`: first row of the tile the K-tile belongs to -- the tile being computed, or with CONT the block's next one)
  auto piece_a = [&](int tm0, int row, int t, int par, char* dst) {
    const unsigned so = (unsigned)((int64_t)(tm0 + row) * p.lda * EB + (DBG == 7 ? (t & 3) : t) * 128);   // DBG 7 (timing experiment): the same four K-tiles over and over = cache-resident operands
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voA[par] + so, 0, 0, 0);
  };
  auto piece_w = [&](int tn0, int row, int t, int par, char* dst) {
    const unsigned so = (unsigned)((int64_t)(tn0 + row) * p.ldw * EB + (DBG == 7 ? (t & 3) : t) * 128);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, voW[par] + so, 0, 0, 0);
  };
  // tile prologue (all 8 waves, 14 pieces each): K-tile 0 whole, A_top and W of K-tile 1
  auto prologue = [&]() {
#pragma unroll
    for (int c = 0; c < APW; ++c) {
      const int ch = (wave & 3) * APW + c;              // chunks of A(0): waves 0-3 -> A_top, waves 4-7 -> A_bot (chunk parity = swizzle key)
      piece_a(sm0, wr * WTM + ch * 8, 0, ch & 1, lds + (wr ? ABOT : ATOP) + ch * 1024);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) piece_w(sn0, (wave * 4 + c) * 8, 0, c & 1, lds + WB + (wave * 4 + c) * 1024);
    if (nk > 1) {
      if constexpr (TBM == 256) {
#pragma unroll
        for (int c = 0; c < 2; ++c) piece_a(sm0, (wave * 2 + c) * 8, 1, c & 1, lds + ATOP + AH + (wave * 2 + c) * 1024);
      } else {                                          // 12 chunks over 8 waves: waves 0-3 two each, waves 4-7 one each
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int ch = wave < 4 ? wave * 2 + c : 8 + (wave - 4);
          if (c == 0 || wave < 4) piece_a(sm0, ch * 8, 1, ch & 1, lds + ATOP + AH + ch * 1024);
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) piece_w(sn0, (wave * 4 + c) * 8, 1, c & 1, lds + WB + WT + (wave * 4 + c) * 1024);
    }
  };
  prologue();
  if (p.skew > 0 && (blockIdx.x & 7)) {                 // persistent blocks: blockIdx & 7 = XCD; the delay persists over the tile walk
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), d = (unsigned long long)(blockIdx.x & 7) * (unsigned)p.skew;
    while (__builtin_amdgcn_s_memtime() - t0 < d) __builtin_amdgcn_s_sleep(16);
  }

  // fragments.  16x16x32: lane -> row (lane&15), 16-B slots (lane>>4) and 4 + (lane>>4) of the 64-k row (two MFMAs per tile);
  // 32x32x16: lane -> row (lane&31), slots 2 kk + (lane>>5), kk = 0..3 (four MFMAs per tile).  24 ds_read_b128 either way.
  const int frow = M32 ? (lane & 31) : (lane & 15), fsw = (lane >> 1) & 7;
  constexpr int NKK = M32 ? 4 : 2;
  int offk[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) offk[kk] = (M32 ? ((kk * 2 + (lane >> 5)) ^ fsw) : ((kk * 4 + (lane >> 4)) ^ fsw)) << 4;
  const int a_base = frow * 128;                        // inside the group's own A ring slot
  const int w_base = (wc * WTN + frow) * 128;
  constexpr int TSTRIDE = (M32 ? 32 : 16) * 128;        // bytes between the row tiles of a fragment set
  bf16x8 af[NKK][TM], wf[NKK][TN];

#define RG_READ_FRAGS(aslot, wslot)                                                                  \
  do {                                                                                               \
    const char* At_ = (aslot) + a_base;                                                              \
    const char* Wt_ = (wslot) + w_base;                                                              \
    _Pragma("unroll") for (int kk = 0; kk < NKK; ++kk)                                               \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
        wf[kk][j] = *reinterpret_cast<const bf16x8*>(Wt_ + j * TSTRIDE + offk[kk]);                  \
    _Pragma("unroll") for (int kk = 0; kk < NKK; ++kk)                                               \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        af[kk][i] = *reinterpret_cast<const bf16x8*>(At_ + i * TSTRIDE + offk[kk]);                  \
  } while (0)

#define RG_MFMA_PART(tail)                                                                           \
  do {                                                                                               \
    if constexpr (F8) {                                                                              \
      if (!(tail)) {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                               \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                             \
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                            \
                __builtin_shufflevector(__builtin_bit_cast(i32x4r, wf[0][j]), __builtin_bit_cast(i32x4r, wf[1][j]), 0, 1, 2, 3, 4, 5, 6, 7), \
                __builtin_shufflevector(__builtin_bit_cast(i32x4r, af[0][i]), __builtin_bit_cast(i32x4r, af[1][i]), 0, 1, 2, 3, 4, 5, 6, 7), \
                acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);                                      \
      }                                                                                              \
    } else                                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < NKK; ++kk)                                               \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                               \
        const bool in_tail = (kk == NKK - 1) && (i >= TM - EARLY);                                   \
        if (in_tail == (tail)) {                                                                     \
          _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
            if constexpr (M32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0); \
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);               \
          }                                                                                          \
        }                                                                                            \
      }                                                                                              \
  } while (0)
#define RG_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

  // DBG == 4: cycle stamps (s_memtime) of block 0, waves 0 and 4, into the buffer passed as `bias`
  unsigned long long* stamps = (DBG == 4 && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0)
                                   ? (unsigned long long*)p.bias + (wave ? 1 : 0) * 64 * 8 : nullptr;
#define RG_STAMP(t, k) do { if (DBG == 4 && stamps && (t) < 64) stamps[(t) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  const int g4 = wave & 3;
  int tile_no = 0;
  // tile-level stamps (DBG == 4), column 7 of rows 5 n .. 5 n + 4 for the block's n-th tile: k-loop entry, k-loop exit,
  // next tile's prologue issued, epilogue issued, (next row group) next k-loop entry
#define RG_TSTAMP(k) do { if (DBG == 4 && stamps && tile_no < 12) stamps[(tile_no * 5 + (k)) * 8 + 7] = __builtin_amdgcn_s_memtime(); } while (0)
  // Tile boundaries with CONT (cont: the block has a next tile and nk >= 2).  The ping-pong keeps its two barriers per K-tile,
  //   X(t) = [group 0: end of L(t) | group 1: end of M(t-1)]      Y(t) = [group 0: end of M(t) | group 1: end of L(t)]
  // and the boundary only stretches the interval between Y(nk-1) and X(0') of the next tile:
  //   group 0:  ... M(nk-1) Y(nk-1) [store tile]          L(0') X(0') M(0') ...
  //   group 1:  ... L(nk-1) Y(nk-1) M(nk-1) [store tile]        X(0') L(0') ...
  // * the pieces of K-tiles 0' / 1' are issued by the LOAD intervals nk-2 / nk-1 exactly as those of t+1 / t+2 inside a tile (same
  //   slots, same counts: the A parity `pa` of K-tile 0 and the W slot `wcur` simply run on), so the counted waits stay the steady ones;
  // * the epilogue's staging patches live in the W slot of K-tile nk-1: every read of it is behind Y(nk-1), and it is the slot W(2')
  //   goes to -- each wave's pieces of W(2') land in that wave's OWN 4-KiB patch (rows 8 (4 g4 + c) .. of its group's half), issued
  //   after the wave's own epilogue, so no block-wide barrier is needed before the slot is reused;
  // * the stores of the epilogue count in vmcnt like the DMA pieces: one vmcnt(0) behind them (they were issued thousands of cycles
  //   after the last pieces) and the next k-loop's counted waits see DMA pieces only.
  // Without CONT (and for nk == 1): prologue burst for the next tile before the epilogue, vmcnt(0) + block barrier at the k-loop entry.
  int wcur = 0, pa = 0;                                  // W ring slot / A parity of K-tile 0 of the current tile
  bool fresh = true;                                     // the tile's K-tiles 0 / 1 come from a prologue burst (first tile, or no CONT)
  for (int vb = blockIdx.x;;) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{};
    if (p.xsync && gridDim.y == 1 && tid == 0) {
      // the XCD's blocks (blockIdx & 7) start their n-th tile together: tiles that share operand panels then stream them through
      // the XCD's L2 in step instead of drifting apart over the rounds.  Bounded spin: a block that cannot see its peers goes on alone.
      const int x = blockIdx.x & 7, G = (int)gridDim.x, nbx = (G - x + 7) >> 3;
      unsigned target = 0;
      for (int r = 0; r <= tile_no; ++r) {
        const int rem = ntiles - G * r - x;
        target += (unsigned)min(max((rem + 7) >> 3, 0), nbx);
      }
      __hip_atomic_fetch_add(p.xsync + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int it = 0; it < 400 && __hip_atomic_load(p.xsync + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++it)
        __builtin_amdgcn_s_sleep(4);
    }
    const int nb = vb + (int)gridDim.x;
    const bool more = nb < ntiles;
    const bool cont = CONT && more && nk >= 2;           // K-tiles nk, nk+1 of this k-loop are K-tiles 0, 1 of the block's next tile
    if (cont) tile_of(nb, sm0, sn0);                     // (sm0, sn0): the tile being staged = the next one from here on
    if (fresh) {
      A3V_WAIT_VM0();
      A3V_BARRIER();
      wcur = 0; pa = 0;
    }
    RG_TSTAMP(0);
    // The k-loop is split into its steady part (t + 2 < nk: every piece belongs to this tile, no condition, addresses advance by a
    // constant) and the last two iterations, which may stage the next tile (TAIL): with the selects in every iteration the LOAD
    // interval grew from ~900 to ~1140 cycles and the whole K-tile period with it (the two intervals are co-critical).
    if (wr == 0) {
      auto iter = [&](int t, auto tailc) {
        constexpr bool TAIL = decltype(tailc)::value;
        RG_STAMP(t, 0);
        const int wn2 = wcur == 0 ? 2 : wcur - 1;        // slot of K-tile t + 2
        if (!TAIL || t + 1 < nk || cont) {
          const bool nx = TAIL && t + 1 >= nk;
          const int tm = nx ? sm0 : m0, tk = nx ? t + 1 - nk : t + 1;
#pragma unroll
          for (int c = 0; c < APW; ++c) piece_a(tm, WTM + (g4 * APW + c) * 8, tk, (g4 * APW + c) & 1, lds + ABOT + ((pa ^ (t + 1)) & 1) * AH + (g4 * APW + c) * 1024);
        }
        if (!TAIL || t + 2 < nk || cont) {
          const bool nx = TAIL && t + 2 >= nk;
          const int tn = nx ? sn0 : n0, tk = nx ? t + 2 - nk : t + 2;
#pragma unroll
          for (int c = 0; c < 4; ++c) piece_w(tn, (g4 * 4 + c) * 8, tk, c & 1, lds + WB + wn2 * WT + (g4 * 4 + c) * 1024);
        }
        if constexpr (LATEW) {
          // everything older than the bursts of L(t-1) and L(t) has landed: in particular this group's W half of tile t (the reads
          // below); in steady state 8 + 8 pieces may stay in flight, the last two tiles of the block's last k-loop issue shorter bursts
          if (!TAIL || t + 2 < nk || cont) vm_wait_imm<2 * BURST>();          // (16 with 256-row tiles)
          else if (t + 2 == nk) vm_wait_imm<BURST + APW>();                      // L(t-1): A + W, L(t): A only (12)
          else vm_wait_imm<APW>();                                                // only A_bot(nk-1) of L(nk-2) may be in flight (4)
        }
        RG_READ_FRAGS(lds + ATOP + ((pa ^ t) & 1) * AH, lds + WB + wcur * WT);
        A3V_WAIT_LGKM0();
        RG_STAMP(t, 1);
        if constexpr (LW) {                              // variant: W(t+1) is waited for HERE (one interval earlier), so nothing stands
          if (!TAIL || t + 2 < nk || cont) RG_VMCNT(8);  // between this group's last MFMA and the barrier that releases the other group
          else if (t + 2 == nk) RG_VMCNT(4);
          else RG_VMCNT(0);                               // (LW: 256-row tiles only)
        } else {
          if (!TAIL || t + 2 < nk || cont) vm_wait_imm<BURST + 4>();              // this burst + the W half of the previous one (12)
          else if (t + 2 == nk) vm_wait_imm<APW + 4>();                          // this burst (A only) + the W half of L(t-1) (8)
          else RG_VMCNT(0);
        }
        RG_STAMP(t, 2);
        A3V_BARRIER();
        RG_STAMP(t, 3);
        __builtin_amdgcn_s_setprio(1);
        RG_MFMA_PART(false);
        __builtin_amdgcn_s_setprio(0);                 // never wait (vmcnt / barrier) at raised priority: measured -20 %
        RG_STAMP(t, 4);
        if constexpr (!LW && !LATEW) {
          if (!TAIL || t + 2 < nk || cont) RG_VMCNT(8);
          else if (t + 2 == nk) RG_VMCNT(4);
          else RG_VMCNT(0);
        }
        RG_STAMP(t, 5);
        if constexpr (EARLY > 0) __builtin_amdgcn_sched_barrier(0);   // keep the tail MFMAs behind the barrier, the others before it
        A3V_BARRIER();
        if constexpr (EARLY > 0) __builtin_amdgcn_sched_barrier(0);
        RG_STAMP(t, 6);
        if constexpr (EARLY > 0) {
          __builtin_amdgcn_s_setprio(1);
          RG_MFMA_PART(true);
          __builtin_amdgcn_s_setprio(0);
        }
        wcur = wcur == 2 ? 0 : wcur + 1;
      };
      int t = 0;
      for (; t + 2 < nk; ++t) iter(t, std::false_type{});
      for (; t < nk; ++t) iter(t, std::true_type{});
      if (!cont) A3V_BARRIER();                          // (with cont the other group's matching barrier is X(0') of the next tile)
    } else {
      if (fresh) A3V_BARRIER();
      auto iter = [&](int t, auto tailc) {
        constexpr bool TAIL = decltype(tailc)::value;
        RG_STAMP(t, 0);
        const int wn2 = wcur == 0 ? 2 : wcur - 1;
        if (!TAIL || t + 2 < nk || cont) {
          const bool nx = TAIL && t + 2 >= nk;
          const int tn = nx ? sn0 : n0, tm = nx ? sm0 : m0, tk = nx ? t + 2 - nk : t + 2;
#pragma unroll
          for (int c = 0; c < 4; ++c) piece_w(tn, 128 + (g4 * 4 + c) * 8, tk, c & 1, lds + WB + wn2 * WT + WT / 2 + (g4 * 4 + c) * 1024);
#pragma unroll
          for (int c = 0; c < APW; ++c) piece_a(tm, (g4 * APW + c) * 8, tk, (g4 * APW + c) & 1, lds + ATOP + ((pa ^ t) & 1) * AH + (g4 * APW + c) * 1024);
        }
        RG_READ_FRAGS(lds + ABOT + ((pa ^ t) & 1) * AH, lds + WB + wcur * WT);
        A3V_WAIT_LGKM0();
        RG_STAMP(t, 1);
        if (!TAIL || t + 2 < nk || cont) vm_wait_imm<BURST>();                   // its previous burst (W half + A_top of tile t+1) has landed (8)
        else RG_VMCNT(0);
        RG_STAMP(t, 2);
        A3V_BARRIER();
        RG_STAMP(t, 3);
        __builtin_amdgcn_s_setprio(1);
        RG_MFMA_PART(false);
        __builtin_amdgcn_s_setprio(0);                 // never wait (vmcnt / barrier) at raised priority: measured -20 %
        RG_STAMP(t, 4);
        RG_STAMP(t, 5);
        if (!TAIL || t + 1 < nk || !cont) {                       // X(t+1) (or the block barrier that ends a k-loop without cont); with cont X(0') follows the epilogue
          if constexpr (EARLY > 0) __builtin_amdgcn_sched_barrier(0);   // keep the tail MFMAs behind the barrier, the others before it
          A3V_BARRIER();
          if constexpr (EARLY > 0) __builtin_amdgcn_sched_barrier(0);
        }
        RG_STAMP(t, 6);
        if constexpr (EARLY > 0) {
          __builtin_amdgcn_s_setprio(1);
          RG_MFMA_PART(true);
          __builtin_amdgcn_s_setprio(0);
        }
        wcur = wcur == 2 ? 0 : wcur + 1;
      };
      int t = 0;
      for (; t + 2 < nk; ++t) iter(t, std::false_type{});
      for (; t < nk; ++t) iter(t, std::true_type{});
    }
    RG_TSTAMP(1);
    // every read of the rings is behind the last barrier.  No cont: stage the next tile now (prologue burst), store this one after.
    if (more && !cont) {
      tile_of(nb, sm0, sn0);
      prologue();
    }
    RG_TSTAMP(2);
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      // staging patches: the W slot of K-tile nk - 1 (cont: the one slot no piece of the next tile is in flight to); the slot behind the
      // prologue's two otherwise
      const int wst = cont ? (wcur == 0 ? 2 : wcur - 1) : 2;
      if constexpr (F8) {
        // acc = sum_k qa[m][k] qw[n][k]: the product's value is that times sa[m] sw[n] (what the general epilogue's F8 form applies);
        // done here so that the staged forms run on fp8 tiles too (the host leaves GEMM_EPI_SCALE out of p.epi for this kernel)
        const int mr_ = m0 + wr * WTM + (lane_e & 15), nc_ = n0 + wc * WTN + (lane_e >> 4) * 4;
        f32x4 swv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) swv[j] = nc_ + j * 16 < p.N ? *reinterpret_cast<const f32x4*>(p.sw + nc_ + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float sam = mr_ + i * 16 < p.M ? p.sa[mr_ + i * 16] : 0.f;
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] *= sam * swv[j][r];
        }
      }
      if constexpr (M32) gemm_epilogue32<TM, TN>(acc, p, m0 + wr * WTM, n0 + wc * WTN, lane_e);
      else gemm_epilogue<TM, TN, false, SET>(acc, p, m0 + wr * WTM, n0 + wc * WTN, lane_e, STAGED ? lds + WB + wst * WT + wave * 4096 : nullptr);
    }
    RG_TSTAMP(3);
    ++tile_no;
    if (!more) break;
    if (cont) {
      A3V_WAIT_LGKM0();                                  // the wave's own patch reads are done: its next pieces may land in the patch
      A3V_WAIT_VM0();                                    // stores (and every older piece) retired: the counted waits count pieces again
      if (wr == 1) A3V_BARRIER();                        // X(0')
      pa ^= nk & 1;
      fresh = false;
    } else {
      fresh = true;
    }
    vb = nb; m0 = sm0; n0 = sn0;
  }
#undef RG_TSTAMP
#undef RG_STAMP
#undef RG_VMCNT
#undef RG_READ_FRAGS
#undef RG_MFMA_PART
}

// ------------------------------------------------------------------------------------
// One-wave-per-SIMD form: 256 threads = 4 waves (2 x 2), each wave owns a 128 x 128 quarter of the 256 x 256 tile (256 fp32
// accumulators in the accumulation registers + two fragment buffers of 64 VGPRs: the 512-register budget of a wave that has its
// SIMD to itself).  No second wave to hand the matrix pipe to: the fragment reads of sub-stage s+1 are issued between the MFMAs
// of sub-stage s by the same wave, and the only synchronisation is one workgroup barrier per 32-k sub-stage.
//   LDS: ring of 5 sub-stages x 32 KiB; a sub-stage = 32 k of the A tile (256 rows x 64 B) + 32 k of the W tile (256 x 64 B);
//        the 16-byte slot q of row r sits at q ^ ((r >> 2) & 3): conflict-free ds_read_b128 of 16 rows x one slot.
//   DMA: each wave issues 8 one-KiB pieces (16 rows x 64 B) per sub-stage, four sub-stages (two K-tile periods) ahead.
//   iteration s:  vmcnt(own pieces of s+1 landed) ; barrier  (=> stage s+1 visible, stage s read by everyone)
//                 per MFMA row (8 MFMAs): two ds_read_b128 of stage s+1 into the other buffer and one DMA piece of stage s+4
//                 -> slot (s+4) % 5 (stage s-1's slot).  All eight pieces in one burst from four waves at once cost 520
//                 cycles per sub-stage: the CU's address unit takes ~16 cycles per piece and the issuing wave stalls
//   LDS read traffic per K-tile and CU: 128 KiB (8-wave kernels: 192 KiB).
// ------------------------------------------------------------------------------------
#ifdef A3V_EXPERIMENTS   // measured and not dispatched: built only with `make EXPERIMENTS=1`
template <int DBG, int SET = EPI_SET_COMMON>   // DBG 4: cycle stamps; 8: no DMA in the k-loop, 9: no DMA and no fragment reads, 10: every k-loop piece out of bounds = issued but fetching nothing (timing experiments, wrong results)
__global__ __launch_bounds__(256) void gemm_nt_bf16_w4_kernel(GemmArgs p) {
  constexpr int TBM = 256, TBN = 256, KS = 32, NST = 5;
  constexpr int HALF = 256 * KS * 2;                    // 16 KiB: the A (or W) part of a sub-stage
  constexpr int STG = 2 * HALF;                         // 32 KiB
  __shared__ __attribute__((aligned(1024))) char lds[NST * STG];   // 163840 B
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int ntiles = p.tiles_m * p.tiles_n;
  auto tile_of = [&](int vb, int& tm0, int& tn0) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int per_group = GROUP_M * p.tiles_n;
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int in_g = bid - group * per_group;
    tm0 = (first_m + in_g % gsz) * TBM;
    tn0 = (in_g / gsz) * TBN;
  };
  int m0, n0, sm0, sn0;
  tile_of(blockIdx.x, m0, n0);
  sm0 = m0; sn0 = n0;
  if (gridDim.y > 1) {                                  // split-K slices as in the ring kernel
    const int nk_all = p.K / BK, z = blockIdx.y, S = gridDim.y;
    const int t0 = (int)(((int64_t)z * nk_all) / S), t1 = (int)(((int64_t)(z + 1) * nk_all) / S);
    p.A += (int64_t)t0 * BK;
    p.W += (int64_t)t0 * BK;
    p.K = (t1 - t0) * BK;
    p.C = reinterpret_cast<char*>(p.C) + (int64_t)z * p.c_split;
  }
  const int ns = p.K / KS;                              // even (K % 64 == 0)
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
  // DMA lane -> (row lane/4 of the 16-row piece, physical slot lane%4) <- logical slot (lane%4) ^ ((row >> 2) & 3)
  const unsigned lsl = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned voA = (unsigned)(((lane >> 2) * p.lda + lsl * 8) * 2);
  const unsigned voW = (unsigned)(((lane >> 2) * p.ldw + lsl * 8) * 2);
  // piece c (0..3: A rows, 4..7: W rows) of this wave's 8 one-KiB pieces of sub-stage s of tile (sm0, sn0)
  // `kill` = 0x80000000 pushes the offset past the buffer's bound: the piece fetches nothing (zeros land in a free slot) but still
  // counts in vmcnt, so the k-loop needs no branches and one constant wait count
  auto piece = [&](int s, int c, unsigned kill) {
    char* const dst = lds + (s % NST) * STG;
    const int ch = wave * 4 + (c & 3);
    if (c < 4) {
      const unsigned so = (unsigned)(((int64_t)(sm0 + ch * 16) * p.lda + s * KS) * 2) | kill;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + ch * 1024), 16, voA + so, 0, 0, 0);
    } else {
      const unsigned so = (unsigned)(((int64_t)(sn0 + ch * 16) * p.ldw + s * KS) * 2) | kill;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + HALF + ch * 1024), 16, voW + so, 0, 0, 0);
    }
  };
  auto stage = [&](int s) {
#pragma unroll
    for (int c = 0; c < 8; ++c) piece(s, c, 0u);
  };
  auto prologue = [&]() {                               // stages 0..3, the two halves of a line back to back (ns is even)
#pragma unroll
    for (int s = 0; s < 4; s += 2)
      if (s < ns) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { piece(s, c, 0u); piece(s + 1, c, 0u); }
      }
  };
  prologue();

  // fragments of one sub-stage: A row tile i -> row wr*128 + 16 i + (lane & 15), logical slot lane >> 4; W likewise with wc
  const int fl = lane & 15;
  const int fq = ((lane >> 4) ^ ((fl >> 2) & 3)) << 4;
  const int a_off = (wr * 128 + fl) * 64 + fq;
  const int w_off = HALF + (wc * 128 + fl) * 64 + fq;
  bf16x8 af[2][8], wf[2][8];
  f32x4 accl[8][4], accr[8][4];                         // columns 0..63 / 64..127 of the wave's quarter (two epilogue calls)

  unsigned long long* stamps = (DBG == 4 && blockIdx.x == 0 && lane == 0) ? (unsigned long long*)p.bias + wave * 64 * 8 : nullptr;
#define W4_STAMP(s, k) do { if (DBG == 4 && stamps && (s) < 64) stamps[(s) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define W4_TSTAMP(k) do { if (DBG == 4 && stamps && tile_no < 12) stamps[tile_no * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  int tile_no = 0;
#define W4_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
  // The k-loop is written instruction by instruction (volatile asm keeps the order): accumulators pinned to the accumulation
  // registers ("+a", in place), fragment reads as explicit ds_read_b128 placed between the MFMA rows.  The compiler's own
  // scheduling of the builtin form moved half of the accumulators into VGPRs and shuffled them through v_accvgpr moves.
#define W4_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define W4_MFMA(c, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a))
  // MFMA j of row g; after it, this wave's slot j of the row: wave PH reads its two fragments behind MFMAs 2 PH and 2 PH + 1
  // (row 7: both behind MFMA PH, so the last reads have >= 4 MFMAs to return in) and issues its DMA piece behind MFMA
  // (2 PH + 4) % 8.  The four waves run in lock step (one barrier per sub-stage): without the stagger they hit the LDS and the
  // address unit in the same cycle and each piece stalls its wave ~50 cycles (40 % of the kernel).
#define W4_SLOT(cur, nxt, s, PH, g, j, EVEN)                                                                  \
  do {                                                                                                        \
    if ((j) < 4) W4_MFMA(accl[g][(j) & 3], wf[cur][j], af[cur][g]);                                           \
    else W4_MFMA(accr[g][(j) & 3], wf[cur][j], af[cur][g]);                                                   \
    if (DBG != 9 && (j) == ((g) < 7 ? 2 * (PH) : (PH))) W4_READ(af[nxt][g], ra_, (g) * 1024);                 \
    if (DBG != 9 && (j) == ((g) < 7 ? 2 * (PH) + 1 : (PH))) W4_READ(wf[nxt][g], rw_, (g) * 1024);             \
    if (DBG < 8 && (EVEN) && (j) == ((2 * (PH) + 4) & 7)) piece((s) + 4, g, kill_);                           \
    if (DBG < 8 && (EVEN) && (j) == ((2 * (PH) + 5) & 7)) piece((s) + 5, g, kill2_);                          \
  } while (0)
  // one sub-stage: MFMAs of stage s from buffer `cur`, the fragments of stage s+1 into buffer `nxt`; EVEN sub-stages issue the
  // pieces of stages s+4 and s+5 pairwise -- the two 64-byte halves of the same 128-byte lines back to back, so the second
  // request finds the line in (or on its way into) the vector L1 instead of fetching it from L2 a second time one sub-stage later
#define W4_ITER(cur, nxt, s, PH, EVEN)                                                                        \
  do {                                                                                                        \
    W4_STAMP(s, 0);                                                                                           \
    if (DBG < 8) W4_VMCNT(16);                          /* everything but the newest pair of stages has landed */ \
    W4_STAMP(s, 1);                                                                                           \
    A3V_BARRIER();                                                                                            \
    W4_STAMP(s, 2);                                                                                           \
    const unsigned nb_ = (unsigned)((((s) + 1) % NST) * STG);                                                 \
    const unsigned ra_ = nb_ + (unsigned)a_off, rw_ = nb_ + (unsigned)w_off;                                  \
    const unsigned kill_ = ((s) + 4 < ns && DBG != 10) ? 0u : 0x80000000u, kill2_ = ((s) + 5 < ns && DBG != 10) ? 0u : 0x80000000u; \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                           \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) W4_SLOT(cur, nxt, s, PH, g, j, EVEN);                     \
    }                                                                                                         \
    W4_STAMP(s, 3);                                                                                           \
    A3V_WAIT_LGKM0();                                                                                         \
    W4_STAMP(s, 4);                                                                                           \
  } while (0)
#define W4_KLOOP(PH)                                                                                          \
  for (int s = 0; s < ns; s += 2) {                     /* (the last sub-stage reads one stage past the end: in-ring, never used) */ \
    W4_ITER(0, 1, s, PH, true);                                                                               \
    W4_ITER(1, 0, s + 1, PH, false);                                                                          \
  }

  // clock probe (DBG >= 4 with a buffer in `bias`): shader-clock and 100-MHz real-time counters at kernel entry / exit of block 0
  unsigned long long* const probe = (DBG >= 4 && p.bias && blockIdx.x == 0 && tid == 0) ? (unsigned long long*)p.bias + 4 * 64 * 8 : nullptr;
  if (probe) { probe[0] = __builtin_amdgcn_s_memtime(); probe[1] = __builtin_amdgcn_s_memrealtime(); }
  for (int vb = blockIdx.x;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { accl[i][j] = f32x4{}; accr[i][j] = f32x4{}; }
    W4_TSTAMP(5);
    A3V_WAIT_VM0();
    A3V_BARRIER();
    W4_TSTAMP(6);
#pragma unroll
    for (int g = 0; g < 8; ++g) { W4_READ(af[0][g], (unsigned)a_off, g * 1024); W4_READ(wf[0][g], (unsigned)w_off, g * 1024); }
    A3V_WAIT_LGKM0();
    if (wave == 0) { W4_KLOOP(0) }
    else if (wave == 1) { W4_KLOOP(1) }
    else if (wave == 2) { W4_KLOOP(2) }
    else { W4_KLOOP(3) }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the last MFMAs retire before the epilogue's v_accvgpr_read (no interlock)
    W4_TSTAMP(7);
    A3V_BARRIER();                                       // the last reads of the ring are done: the next tile may land
    const int nb = vb + (int)gridDim.x;
    if (nb < ntiles) {
      tile_of(nb, sm0, sn0);
      prologue();
    }
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      char* const patch = lds + 4 * STG + wave * 4096;   // slot 4: the prologue fills slots 0..3
      gemm_epilogue<8, 4, false, SET>(accl, p, m0 + wr * 128, n0 + wc * 128, lane_e, patch);
      gemm_epilogue<8, 4, false, SET>(accr, p, m0 + wr * 128, n0 + wc * 128 + 64, lane_e, patch);
    }
    ++tile_no;
    if (nb >= ntiles) break;
    vb = nb; m0 = sm0; n0 = sn0;
  }
  if (probe) { probe[2] = __builtin_amdgcn_s_memtime(); probe[3] = __builtin_amdgcn_s_memrealtime(); }
#undef W4_TSTAMP
#undef W4_KLOOP
#undef W4_ITER
#undef W4_SLOT
#undef W4_READ
#undef W4_MFMA
#undef W4_VMCNT
#undef W4_STAMP
}
#endif  // A3V_EXPERIMENTS

// ------------------------------------------------------------------------------------
// "Overlapped" form: the 8 waves and 128 x 64 wave tiles of the ring kernel on the 32-k sub-stage ring of the kernel above, but
// no L / M phases: every wave runs ONE stream in which its own fragment reads (12 per sub-stage) and DMA pieces sit in the
// shadows of its MFMAs (32 per sub-stage), hand-ordered (volatile asm), accumulators pinned to AGPRs.  The two waves of a SIMD
// issue into the matrix pipe whenever they can; when one stalls on a load issue the other one's MFMAs go out -- nobody hands
// the pipe over.  One barrier per sub-stage; the two groups meet it half a sub-stage apart (group 0 at the top of its
// iteration, group 1 after its 16th MFMA), so while one wave of a SIMD waits, its partner still has MFMAs to issue.
//   group 0, iteration s:  vmcnt ; barrier(s) ; rows 0..7: 4 MFMAs + reads of stage s+1 (+ pieces on even s) ; lgkmcnt(0)
//   group 1, iteration s:  rows 0..3: 4 MFMAs (+ pieces on even s) ; vmcnt ; barrier(s) ; rows 4..7: 4 MFMAs + reads ; lgkmcnt(0)
//   pieces: 16 rows x 64 B; even sub-stages issue stages s+4 and s+5 pairwise (both halves of a 128-B line back to back).
// ------------------------------------------------------------------------------------
#ifdef A3V_EXPERIMENTS   // measured and not dispatched: built only with `make EXPERIMENTS=1`
template <int DBG, int SET = EPI_SET_COMMON>
__global__ __launch_bounds__(512) void gemm_nt_bf16_ov_kernel(GemmArgs p) {
  constexpr int TBM = 256, TBN = 256, KS = 32, NST = 5;
  constexpr int HALF = 256 * KS * 2;                    // 16 KiB: the A (or W) part of a sub-stage
  constexpr int STG = 2 * HALF;                         // 32 KiB
  __shared__ __attribute__((aligned(1024))) char lds[NST * STG];   // 163840 B
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;             // wr = group: waves 0..3 and 4..7 share the four SIMDs pairwise

  const int ntiles = p.tiles_m * p.tiles_n;
  auto tile_of = [&](int vb, int& tm0, int& tn0) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int per_group = GROUP_M * p.tiles_n;
    const int group = bid / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int in_g = bid - group * per_group;
    tm0 = (first_m + in_g % gsz) * TBM;
    tn0 = (in_g / gsz) * TBN;
  };
  int m0, n0, sm0, sn0;
  tile_of(blockIdx.x, m0, n0);
  sm0 = m0; sn0 = n0;
  if (gridDim.y > 1) {                                  // split-K slices as in the ring kernel
    const int nk_all = p.K / BK, z = blockIdx.y, S = gridDim.y;
    const int t0 = (int)(((int64_t)z * nk_all) / S), t1 = (int)(((int64_t)(z + 1) * nk_all) / S);
    p.A += (int64_t)t0 * BK;
    p.W += (int64_t)t0 * BK;
    p.K = (t1 - t0) * BK;
    p.C = reinterpret_cast<char*>(p.C) + (int64_t)z * p.c_split;
  }
  const int ns = p.K / KS;                              // even (K % 64 == 0)
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
  const unsigned lsl = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned voA = (unsigned)(((lane >> 2) * p.lda + lsl * 8) * 2);
  const unsigned voW = (unsigned)(((lane >> 2) * p.ldw + lsl * 8) * 2);
  // piece c (0, 1: A rows, 2, 3: W rows) of this wave's 4 one-KiB pieces of sub-stage s; `kill`: see the kernel above
  auto piece = [&](int s, int c, unsigned kill) {
    char* const dst = lds + (s % NST) * STG;
    const int ch = wave * 2 + (c & 1);
    if (c < 2) {
      const unsigned so = (unsigned)(((int64_t)(sm0 + ch * 16) * p.lda + (DBG == 11 ? (s & 3) : DBG == 12 ? (s & 7) : DBG == 13 ? (s & 31) : s) * KS) * 2) | kill;   // DBG 11 / 12 / 13: the same 4 / 8 / 32 sub-stages over and over (L1 / L2 hits)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + ch * 1024), 16, voA + so, 0, 0, 0);
    } else {
      const unsigned so = (unsigned)(((int64_t)(sn0 + ch * 16) * p.ldw + (DBG == 11 ? (s & 3) : DBG == 12 ? (s & 7) : DBG == 13 ? (s & 31) : s) * KS) * 2) | kill;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + HALF + ch * 1024), 16, voW + so, 0, 0, 0);
    }
  };
  auto prologue = [&]() {                               // stages 0..3, the two halves of a line back to back (ns is even)
#pragma unroll
    for (int s = 0; s < 4; s += 2)
      if (s < ns) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { piece(s, c, 0u); piece(s + 1, c, 0u); }
      }
  };
  prologue();

  const int fl = lane & 15;
  const int fq = ((lane >> 4) ^ ((fl >> 2) & 3)) << 4;
  const int a_off = (wr * 128 + fl) * 64 + fq;
  const int w_off = HALF + (wc * 64 + fl) * 64 + fq;
  bf16x8 af[2][8], wf[2][4];
  f32x4 acc[8][4];

#define OV_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define OV_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define OV_MFMA(c, w, a) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a))
  // the 12 reads of stage s+1 in read slots 0..15 (slot = 4 * row-in-window + MFMA index): A_0..A_7 in slots 0..7, W_0..W_3 in 8..11
#define OV_RSLOT(nxt, q)                                                                                      \
  do {                                                                                                        \
    if (DBG != 9 && (q) < 8) OV_READ(af[nxt][(q) & 7], ra_, ((q) & 7) * 1024);                                \
    if (DBG != 9 && (q) >= 8 && (q) < 12) OV_READ(wf[nxt][(q) & 3], rw_, ((q) & 3) * 1024);                   \
  } while (0)
  // row g of the MFMA stream.  RB = first row of the read window (group 0: rows 0..7 two MFMAs apart, group 1: rows 4..7 every MFMA)
#define OV_ROW(cur, nxt, s, g, GRP, EVEN)                                                                     \
  do {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
      OV_MFMA(acc[g][j], wf[cur][j], af[cur][g]);                                                             \
      if ((GRP) == 0) { if (((j) & 1) == 0) OV_RSLOT(nxt, (g) * 2 + ((j) >> 1)); }                            \
      else { if ((g) >= 4) OV_RSLOT(nxt, ((g) - 4) * 4 + (j)); }                                              \
      if (DBG < 8 && (EVEN) && (j) == 1 && ((GRP) == 0 || (g) < 4)) {                                         \
        if ((GRP) == 0) piece((s) + 4 + ((g) & 1), (g) >> 1, ((g) & 1) ? kill2_ : kill_);                     \
        else { piece((s) + 4, g, kill_); }                                                                    \
      }                                                                                                       \
      if (DBG < 8 && (EVEN) && (j) == 3 && (GRP) == 1 && (g) < 4) piece((s) + 5, g, kill2_);                  \
    }                                                                                                         \
  } while (0)
#define OV_ITER(cur, nxt, s, GRP, EVEN)                                                                       \
  do {                                                                                                        \
    const unsigned nb_ = (unsigned)((((s) + 1) % NST) * STG);                                                 \
    const unsigned ra_ = nb_ + (unsigned)a_off, rw_ = nb_ + (unsigned)w_off;                                  \
    const unsigned kill_ = ((s) + 4 < ns && DBG != 10) ? 0u : 0x80000000u, kill2_ = ((s) + 5 < ns && DBG != 10) ? 0u : 0x80000000u; \
    if ((GRP) == 0) {                                                                                         \
      if (DBG < 8) OV_VMCNT(8);                         /* all but the newest pair of stages (4 pieces each) */ \
      A3V_BARRIER();                                                                                          \
      _Pragma("unroll") for (int g = 0; g < 8; ++g) OV_ROW(cur, nxt, s, g, GRP, EVEN);                        \
    } else {                                                                                                  \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) OV_ROW(cur, nxt, s, g, GRP, EVEN);                        \
      if (DBG < 8) { if (EVEN) OV_VMCNT(16); else OV_VMCNT(8); }   /* EVEN: this iteration's pair is already out */ \
      A3V_BARRIER();                                                                                          \
      _Pragma("unroll") for (int g = 4; g < 8; ++g) OV_ROW(cur, nxt, s, g, GRP, EVEN);                        \
    }                                                                                                         \
    A3V_WAIT_LGKM0();                                                                                         \
  } while (0)
#define OV_KLOOP(GRP)                                                                                         \
  for (int s = 0; s < ns; s += 2) {                     /* (the last sub-stage reads one stage past the end: in-ring, never used) */ \
    OV_ITER(0, 1, s, GRP, true);                                                                              \
    OV_ITER(1, 0, s + 1, GRP, false);                                                                         \
  }

  for (int vb = blockIdx.x;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{};
    A3V_WAIT_VM0();
    A3V_BARRIER();
#pragma unroll
    for (int g = 0; g < 8; ++g) OV_READ(af[0][g], (unsigned)a_off, g * 1024);
#pragma unroll
    for (int g = 0; g < 4; ++g) OV_READ(wf[0][g], (unsigned)w_off, g * 1024);
    A3V_WAIT_LGKM0();
    if (wr == 0) { OV_KLOOP(0) }
    else { OV_KLOOP(1) }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the last MFMAs retire before the epilogue's v_accvgpr_read (no interlock)
    A3V_BARRIER();                                       // the last reads of the ring are done: the next tile may land
    const int nb = vb + (int)gridDim.x;
    if (nb < ntiles) {
      tile_of(nb, sm0, sn0);
      prologue();
    }
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      char* const patch = lds + 4 * STG + wave * 4096;   // slot 4: the prologue fills slots 0..3
      gemm_epilogue<8, 4, false, SET>(acc, p, m0 + wr * 128, n0 + wc * 64, lane_e, patch);
    }
    if (nb >= ntiles) break;
    vb = nb; m0 = sm0; n0 = sn0;
  }
#undef OV_KLOOP
#undef OV_ITER
#undef OV_ROW
#undef OV_RSLOT
#undef OV_MFMA
#undef OV_READ
#undef OV_VMCNT
}
#endif  // A3V_EXPERIMENTS

// ------------------------------------------------------------------------------------
// fp8 (OCP e4m3fn) form of the 256x256 ping-pong kernel: A [M][K] and W [N][K] are fp8 bytes, one k-tile is 128 elements =
// the same 128-byte LDS rows, DMA pieces and swizzle as the bf16 kernel's 64-element tile, and the two 16-byte fragment reads
// of a lane (16-B chunks g and 4 + g of its row) feed ONE v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) instead of
// two 16x16x32 bf16 MFMAs: the same cycles per k-tile for twice the k.  Which 32 of the 128 k a lane group holds does not
// matter as long as A and W agree (tools/ubench/mxfp8.hip).  Per-row dequantisation scales are applied in the epilogue.
// ------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
__global__ __launch_bounds__(512) void gemm_nt_fp8_pp_kernel(GemmArgs p) {
  constexpr int TBM = 256, TBN = 256, WTM = 128, WTN = 64, TM = 8, TN = 4, KT = 128;
  constexpr int STAGE = (TBM + TBN) * KT;   // 64 KiB
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

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
  const int m0 = tm * TBM, n0 = tn * TBN;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // split-K (gridDim.y slices -> raw fp32 planes, the tail rows of the hybrid dispatch): this block's k-tile range [kt0, nk)
  const int nk_all = p.K / KT;
  const int kt0 = (int)((int64_t)nk_all * blockIdx.y / gridDim.y), nk = (int)((int64_t)nk_all * (blockIdx.y + 1) / gridDim.y);
  if (gridDim.y > 1) p.C = reinterpret_cast<char*>(p.C) + (int64_t)blockIdx.y * p.c_split;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((int64_t)(p.M - 1) * p.lda + p.K), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((int64_t)(p.N - 1) * p.ldw + p.K), 0x00020000);
  const unsigned lr = lane >> 3;
  unsigned voA[2], voW[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const unsigned sl = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
    voA[par] = (unsigned)(lr * p.lda + sl * 16);
    voW[par] = (unsigned)(lr * p.ldw + sl * 16);
  }
  auto stage = [&](int t) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int ch = wave * 4 + (c & 3);
      char* dst = lds + (t & 1) * STAGE + (c >= 4 ? TBM * KT : 0) + ch * 1024;
      if (c < 4) {
        const unsigned so = (unsigned)((int64_t)(m0 + ch * 8) * p.lda + t * KT);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voA[c & 1] + so, 0, 0, 0);
      } else {
        const unsigned so = (unsigned)((int64_t)(n0 + ch * 8) * p.ldw + t * KT);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, voW[c & 1] + so, 0, 0, 0);
      }
    }
  };
  stage(kt0);
  if (nk > kt0 + 1) stage(kt0 + 1);
  A3V_WAIT_VM0();
  A3V_BARRIER();

  const int frow = lane & 15, fsw = (lane >> 1) & 7, fks = lane >> 4;
  const int off0 = ((0 * 4 + fks) ^ fsw) << 4, off1 = ((1 * 4 + fks) ^ fsw) << 4;
  const int a_base = (wr * WTM + frow) * 128;
  const int w_base = TBM * KT + (wc * WTN + frow) * 128;
  i32x4 alo[TM], ahi[TM], wlo[TN], whi[TN];
#define F8_READ_FRAGS(cur)                                                                           \
  do {                                                                                               \
    const char* At_ = (cur) + a_base;                                                                \
    const char* Wt_ = (cur) + w_base;                                                                \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                 \
      wlo[j] = *reinterpret_cast<const i32x4*>(Wt_ + j * 2048 + off0);                               \
      whi[j] = *reinterpret_cast<const i32x4*>(Wt_ + j * 2048 + off1);                               \
    }                                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                 \
      alo[i] = *reinterpret_cast<const i32x4*>(At_ + i * 2048 + off0);                               \
      ahi[i] = *reinterpret_cast<const i32x4*>(At_ + i * 2048 + off1);                               \
    }                                                                                                \
  } while (0)
#define F8_MFMA_ALL()                                                                                \
  do {                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(wlo[j], whi[j], 0, 1, 2, 3, 4, 5, 6, 7), \
            __builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7), acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);    \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  } while (0)
  if (wr == 0) {
    for (int t = kt0; t < nk; ++t) {
      F8_READ_FRAGS(lds + (t & 1) * STAGE);
      if (t >= kt0 + 1 && t + 1 < nk) stage(t + 1);
      A3V_WAIT_LGKM0();
      A3V_BARRIER();
      F8_MFMA_ALL();
      A3V_WAIT_VM0();
      A3V_BARRIER();
    }
    A3V_BARRIER();
  } else {
    A3V_BARRIER();
    for (int t = kt0; t < nk; ++t) {
      F8_READ_FRAGS(lds + (t & 1) * STAGE);
      A3V_WAIT_LGKM0();
      A3V_WAIT_VM0();
      A3V_BARRIER();
      if (t + 2 < nk) stage(t + 2);
      F8_MFMA_ALL();
      A3V_BARRIER();
    }
  }
#undef F8_READ_FRAGS
#undef F8_MFMA_ALL
  gemm_epilogue<TM, TN, true>(acc, p, m0 + wr * WTM, n0 + wc * WTN, lane);
}

// ------------------------------------------------------------------------------------
// "TN" form of the 256x256 ping-pong kernel: C[M,N] = At^T . Wt with BOTH operands K-major, At [K][lda] (m contiguous)
// and Wt [K][ldw] (n contiguous) -- the weight-gradient product dW = dY^T . X on the activations as they sit in memory
// (token-major), without materialising dY^T and X^T.  A stage tile is [64 k][256 m] (512-B rows, LDS-DMA of two k-rows
// per instruction); MFMA fragments come from ds_read_b64_tr_b16 (gfx950 transpose read: the 16 lanes of a group pass the
// addresses of a 4 x 16 block -- lane i: row i>>2, columns 4 (i&3).. -- and lane c receives column c), two reads per
// 8-k operand.  The 32-B column chunks of a row are XOR-swizzled on the DMA source side with key(k) = (k>>3 & 3)*4 + (k & 3):
// the 16 k-rows one read instruction touches (k = 8g + 4jj + r over lane groups g and r = 0..3) get 16 different keys, and
// the 8 rows of either wave half 8 different keys mod 8 -- 512 B over all 64 banks twice, conflict-free.  K need not be
// a tile multiple: rows past K are out of the buffer descriptor's range and read as zero.
// ------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <bool A_ROWS>
__global__ __launch_bounds__(512) void gemm_tn_bf16_pp_kernel(GemmArgs p) {
  constexpr int TBM = 256, TBN = 256, WTM = 128, WTN = 64, TM = 8, TN = 4;
  constexpr int STAGE = (TBM + TBN) * BK * 2;   // 64 KiB
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  if (p.xmap && bid < (nwg & ~255)) {
    // blocks are dispatched in index order, 256 (one per CU) at a time: within each such round XCD x takes the x-th run of 32
    // consecutive tiles, so all eight XCDs share the round's row-group panels through the Infinity Cache (see the ring kernel)
    bid = (bid & ~255) + (bid & 7) * 32 + ((bid & 255) >> 3);
  } else {
    const int base = p.xmap ? (nwg & ~255) : 0, R = nwg - base, v = bid - base;
    const int xcd = v & 7, q = R >> 3, r = R & 7;
    bid = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  }
  const int per_group = GROUP_M * p.tiles_n;
  const int group = bid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_g = bid - group * per_group;
  int tm = first_m + in_g % gsz;
  int tn = ((p.xmap & 2) && (group & 1)) ? p.tiles_n - 1 - in_g / gsz : in_g / gsz;   // serpentine over the row groups (see the ring kernel)
  if (p.xmap & 4) {                          // super-tile walk (see the ring kernel): 256 consecutive blocks = one 16 x 16 block of tiles
    const int b = blockIdx.x, rnd = b >> 8, x = b & 7, i = (b & 255) >> 3, nsc = p.tiles_n >> 4;
    tm = (rnd / nsc) * 16 + (x >> 2) * 8 + (i & 7);
    tn = (rnd % nsc) * 16 + (x & 3) * 4 + (i >> 3);
  }
  const int m0 = tm * TBM, n0 = tn * TBN;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // split-K: gridDim.y slices of the k-tile range, slice z -> output plane z (raw fp32 partial sums, a3v_splitk_reduce adds them)
  const int nk_all = (p.K + BK - 1) / BK;
  const int kt0 = (int)((int64_t)nk_all * blockIdx.y / gridDim.y), nk = (int)((int64_t)nk_all * (blockIdx.y + 1) / gridDim.y);
  if (gridDim.y > 1) p.C = reinterpret_cast<char*>(p.C) + (int64_t)blockIdx.y * p.c_split;
  // waves whose 128 x 64 part of the tile lies outside C (adapter-sized M or N) keep the barriers but skip reads and MFMAs
  const bool active = (m0 + wr * WTM < p.M) && (n0 + wc * WTN < p.N);
  // A_ROWS ("NN" form, C = A . Wt): A is [M][K] row-major and takes the NT kernel's staging (8-row chunks of 128-B rows, 16-B slot
  // swizzle) and ds_read_b128 fragments; only Wt [K][N] goes through the transpose reads.  K % 64 == 0 there (host-checked).
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, A_ROWS ? (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2)
                                                                           : (int)(((int64_t)(p.K - 1) * p.lda + p.M) * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)(((int64_t)(p.K - 1) * p.ldw + p.N) * 2), 0x00020000);
  // piece c of this wave: DMA instruction q = wave*4 + c of the tile (k-rows 2q, 2q+1); lane -> row 2q + (lane>>5),
  // 16-B position lane&31 of the 512-B row, which holds source chunk ((pos>>1) ^ key(k)) * 32 B + (pos&1) * 16 B
  unsigned voA[4], voW[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned kin = 2 * (wave * 4 + c) + (lane >> 5);
    const unsigned col = (((((unsigned)lane & 31) >> 1) ^ (((kin >> 3) & 3) * 4 + (kin & 3))) << 5) + (lane & 1) * 16;
    voA[c] = (unsigned)(kin * p.lda * 2) + col;
    voW[c] = (unsigned)(kin * p.ldw * 2) + col;
    if constexpr (A_ROWS) {   // row = lane/8 of the 8-row chunk, 16-B slot = (lane%8) ^ ((chunk*4 + lane/16) & 7); chunk parity = c & 1
      const unsigned sl = (lane & 7) ^ (((c & 1) * 4 + (lane >> 4)) & 7);
      voA[c] = (unsigned)(((lane >> 3) * p.lda + sl * 8) * 2);
    }
  }
  auto stage_piece = [&](int t, int c) {
    char* dst = lds + (t & 1) * STAGE + (c >= 4 ? TBM * BK * 2 : 0) + (wave * 4 + (c & 3)) * 1024;
    if (c < 4) {
      const unsigned so = A_ROWS ? (unsigned)(((int64_t)(m0 + (wave * 4 + c) * 8) * p.lda + t * BK) * 2)
                                 : (unsigned)(((int64_t)t * BK * p.lda + m0) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voA[c & 3] + so, 0, 0, 0);
    } else {
      const unsigned so = (unsigned)(((int64_t)t * BK * p.ldw + n0) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, voW[c & 3] + so, 0, 0, 0);
    }
  };
  auto stage = [&](int t) {
#pragma unroll
    for (int c = 0; c < 8; ++c) stage_piece(t, c);
  };
  stage(kt0);
  if (nk > kt0 + 1) stage(kt0 + 1);
  A3V_WAIT_VM0();
  A3V_BARRIER();

  // fragment addressing: lane (g = lane>>4, il = lane&15); k sub-block (ks, jj): rows ks*32 + 8g + 4jj + (il>>2)
  const int fg = lane >> 4, il = lane & 15;
  int kro[2][2], kxo[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int k = ks * 32 + 8 * fg + 4 * jj + (il >> 2);
      kro[ks][jj] = k * 512 + (il & 3) * 8;
      kxo[ks][jj] = ((k >> 3) & 3) * 4 + (k & 3);
    }
  const int off0 = ((0 * 4 + fg) ^ ((lane >> 1) & 7)) << 4, off1 = ((1 * 4 + fg) ^ ((lane >> 1) & 7)) << 4;   // A_ROWS fragment slots
  const int a_base = (wr * WTM + il) * 128;
  bf16x8 af0[TM], af1[TM], wf0[TN], wf1[TN];
  auto tr8 = [&](const char* tile, int blk16, int ks) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + kro[ks][0] + ((blk16 ^ kxo[ks][0]) << 5)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + kro[ks][1] + ((blk16 ^ kxo[ks][1]) << 5)));
    bf16x8 r;
    const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    __builtin_memcpy(&r, v, 16);
    return r;
  };
#define TN_READ_FRAGS(cur)                                                                           \
  do {                                                                                               \
    const char* At_ = (cur);                                                                         \
    const char* Wt_ = (cur) + TBM * BK * 2;                                                          \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                 \
      wf0[j] = tr8(Wt_, wc * 4 + j, 0);                                                              \
      wf1[j] = tr8(Wt_, wc * 4 + j, 1);                                                              \
    }                                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                 \
      if constexpr (A_ROWS) {                                                                        \
        af0[i] = *reinterpret_cast<const bf16x8*>(At_ + a_base + i * 2048 + off0);                   \
        af1[i] = *reinterpret_cast<const bf16x8*>(At_ + a_base + i * 2048 + off1);                   \
      } else {                                                                                       \
        af0[i] = tr8(At_, wr * 8 + i, 0);                                                            \
        af1[i] = tr8(At_, wr * 8 + i, 1);                                                            \
      }                                                                                              \
    }                                                                                                \
  } while (0)
#define TN_MFMA_ALL()                                                                                \
  do {                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[j], af0[i], acc[i][j], 0, 0, 0);     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[j], af1[i], acc[i][j], 0, 0, 0);     \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  } while (0)
  if (wr == 0) {
    for (int t = kt0; t < nk; ++t) {
      if (active) TN_READ_FRAGS(lds + (t & 1) * STAGE);
      if (t >= kt0 + 1 && t + 1 < nk) stage(t + 1);
      A3V_WAIT_LGKM0();
      A3V_BARRIER();
      if (active) TN_MFMA_ALL();
      A3V_WAIT_VM0();
      A3V_BARRIER();
    }
    A3V_BARRIER();
  } else {
    A3V_BARRIER();
    for (int t = kt0; t < nk; ++t) {
      if (active) TN_READ_FRAGS(lds + (t & 1) * STAGE);
      A3V_WAIT_LGKM0();
      A3V_WAIT_VM0();
      A3V_BARRIER();
      if (t + 2 < nk) stage(t + 2);
      if (active) TN_MFMA_ALL();
      A3V_BARRIER();
    }
  }
#undef TN_READ_FRAGS
#undef TN_MFMA_ALL
  // every read of the stage buffers is behind the last barrier: each wave takes a private 4 KiB of them for the row-contiguous stores
  // (sums of squares only in the TN form: the NN form computes input gradients)
  if (active) gemm_epilogue<TM, TN, false, (A_ROWS ? EPI_SET_COMMON : EPI_SET_COMMON | EPI_SET_SUMSQ)>(acc, p, m0 + wr * WTM, n0 + wc * WTN, lane, lds + wave * 4096);
}

// A ring form of this kernel (A_top / A_bot / W rings over all 160 KiB as in gemm_nt_bf16_ring_kernel; for TN the A halves as
// [64 k][128 m] tiles with 256-B rows and key(k) & 7) was built and measured in round 2: bit-identical, but NN +1..3 % and TN -7..9 %
// against this two-stage form (gpurun_out/tn_ring_b.log) -- the transpose-read L interval, not the DMA flight time, paces these
// loops -- so it was not kept.  One finding worth keeping: with ds_read_tr builtins in the loop, hipcc orders every fragment read
// behind ALL outstanding LDS-DMA builtins (a compiler-inserted s_waitcnt vmcnt(0)); LDS-DMA that must stay in flight across
// transpose reads has to be issued from inline asm (s_mov_b32 m0 / buffer_load_dwordx4 ... offen lds).
// Same schedule with v_mfma_f32_32x32x16_bf16 (8-pass, higher sustained rate than 16x16x32):
// wave tile 128x64 = 4x2 tiles of 32x32, 4 k-steps of 16 per K-tile, 32 MFMAs per interval.
#ifdef A3V_EXPERIMENTS   // measured and not dispatched: built only with `make EXPERIMENTS=1`
template <int DBG>
__global__ __launch_bounds__(512) void gemm_nt_bf16_pp32_kernel(GemmArgs p) {
  constexpr int TBM = 256, TBN = 256, NW = 8, WTM = 128, WTN = 64, TM = 4, TN = 2;
  constexpr int STAGE = (TBM + TBN) * BK * 2;   // 64 KiB
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

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
  const int m0 = tm * TBM, n0 = tn * TBN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  // LDS-DMA through buffer descriptors: rows past M / N are out of range and read as zero (no
  // clamping VALU), addresses are {SGPR descriptor, 32-bit VGPR offset, SGPR k-offset}.
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
  // per-lane byte offset inside an 8-row chunk: row = lane/8, 16-B slot = (lane%8) ^ ((chunk*4 + lane/16) & 7)
  const unsigned lr = lane >> 3;
  unsigned voA[2], voW[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const unsigned sl = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
    voA[par] = (unsigned)((lr * p.lda + sl * 8) * 2);
    voW[par] = (unsigned)((lr * p.ldw + sl * 8) * 2);
  }
  // one 1-KiB DMA piece: c in [0,8): 0..3 -> A chunks, 4..7 -> W chunks of this wave
  auto stage_piece = [&](int t, int c) {
    const int ch = wave * 4 + (c & 3);
    char* dst = lds + (t & 1) * STAGE + (c >= 4 ? TBM * BK * 2 : 0) + ch * 1024;
    if (c < 4) {
      // (row-chunk + k) offset is wave-uniform and changes with t: an SGPR sum added per piece, so
      // nothing per-piece stays live in VGPRs across the loop
      const unsigned so = (unsigned)(((int64_t)(m0 + ch * 8) * p.lda + t * BK) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voA[c & 1] + so, 0, 0, 0);
    } else {
      const unsigned so = (unsigned)(((int64_t)(n0 + ch * 8) * p.ldw + t * BK) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, voW[c & 1] + so, 0, 0, 0);
    }
  };
  auto stage = [&](int t) {
#pragma unroll
    for (int c = 0; c < 8; ++c) stage_piece(t, c);
  };
  stage(0);
  if (nk > 1) stage(1);
  A3V_WAIT_VM0();
  A3V_BARRIER();

  // fragments: lane -> row (lane&31) of a 32-row tile, 16-B slot kk*2 + (lane>>5) of the 64-k row
  const int frow = lane & 31, fsw = (lane >> 1) & 7, fhh = lane >> 5;
  int offk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) offk[kk] = ((kk * 2 + fhh) ^ fsw) << 4;
  const int a_base = (wr * WTM + frow) * 128;
  const int w_base = TBM * BK * 2 + (wc * WTN + frow) * 128;
  bf16x8 af[4][TM], wf[4][TN];

#define PP_READ_FRAGS(cur)                                                                           \
  do {                                                                                               \
    const char* At_ = (cur) + a_base;                                                                \
    const char* Wt_ = (cur) + w_base;                                                                \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                               \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                 \
        wf[kk][j] = *reinterpret_cast<const bf16x8*>(Wt_ + j * 4096 + offk[kk]);                     \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        af[kk][i] = *reinterpret_cast<const bf16x8*>(At_ + i * 4096 + offk[kk]);                     \
    }                                                                                                \
  } while (0)

#define PP_MFMA_ALL()                                                                                \
  do {                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                 \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                               \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                   \
  } while (0)

  constexpr bool do_dma = !(DBG & 1), do_rd = !(DBG & 2);
  if (wr == 0) {
    for (int t = 0; t < nk; ++t) {
      if (do_rd || t == 0) PP_READ_FRAGS(lds + (t & 1) * STAGE);
      if (do_dma && t >= 1 && t + 1 < nk) stage(t + 1);
      A3V_WAIT_LGKM0();
      A3V_BARRIER();
      PP_MFMA_ALL();
      A3V_WAIT_VM0();
      A3V_BARRIER();
    }
    A3V_BARRIER();
  } else {
    A3V_BARRIER();
    for (int t = 0; t < nk; ++t) {
      if (do_rd || t == 0) PP_READ_FRAGS(lds + (t & 1) * STAGE);
      A3V_WAIT_LGKM0();
      A3V_WAIT_VM0();
      A3V_BARRIER();
      if (do_dma && t + 2 < nk) stage(t + 2);
      PP_MFMA_ALL();
      A3V_BARRIER();
    }
  }
#undef PP_READ_FRAGS
#undef PP_MFMA_ALL
  gemm_epilogue32<TM, TN>(acc, p, m0 + wr * WTM, n0 + wc * WTN, lane);
}
#endif  // A3V_EXPERIMENTS

// ------------------------------------------------------------------------------------
// Skinny GEMM, single launch: one 8-wave block per 16 (or 32 with SwiGLU: gate block + up block)
// rows of W.  The block's waves split K, each streams its slice of the W rows straight into MFMA
// operand registers (8 independent 16-B non-temporal loads in flight per lane), the eight partial
// 16x16 accumulators are summed through LDS and wave 0 applies the epilogue.  W is read from HBM
// exactly once, nothing is written but C: algorithmic bytes = 2 N K (+ M K x re-reads from L2).
// ------------------------------------------------------------------------------------
struct Skinny1Args {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const void* res;
  int64_t lda, ldw, ldc, ldr;
  int M, N, K, epi, kslice;
};

template <int TILES>   // 1: 16 rows per block; 2: 32 rows (interleaved gate/up pair)
__global__ __launch_bounds__(512) void gemm_skinny1_bf16_kernel(Skinny1Args p) {
  __shared__ float red[8][TILES][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16 * TILES;
  const int row = lane & 15, kq = (lane >> 4) * 8;
  const int ar = row < p.M ? row : p.M - 1;
  const bf16_t* ap = p.A + (int64_t)ar * p.lda + wave * p.kslice + kq;
  const bf16_t* wp[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    int wr = n0 + t * 16 + row;
    wr = wr < p.N ? wr : p.N - 1;
    wp[t] = p.W + (int64_t)wr * p.ldw + wave * p.kslice + kq;
  }
  f32x4 acc[TILES][2];
#pragma unroll
  for (int t = 0; t < TILES; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int kend = min(p.kslice, p.K - wave * p.kslice);     // the last wave may own a shorter (or empty) slice
  constexpr int U = TILES == 1 ? 8 : 4;                      // 32-k steps per unrolled iteration
  int k = 0;
  for (; k + U * 32 <= kend; k += U * 32) {
    bf16x8 w[TILES][U], a[U];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int q = 0; q < U; ++q) w[t][q] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + k + q * 32));
#pragma unroll
    for (int q = 0; q < U; ++q) a[q] = *reinterpret_cast<const bf16x8*>(ap + k + q * 32);
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int q = 0; q < U; ++q) acc[t][q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][q], a[q], acc[t][q & 1], 0, 0, 0);
  }
  for (; k < kend; k += 32) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + k);
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const bf16x8 w = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + k));
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc[t][0], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][t][lane][r] = acc[t][0][r] + acc[t][1][r];
  __syncthreads();
  if (wave != 0) return;
  float v[TILES][4];
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) a += red[w8][t][lane][r];
      v[t][r] = a;
    }
  // D[n = (lane>>4)*4 + r][m = lane&15]
  const int m = lane & 15;
  if (m >= p.M) return;
  if (TILES == 2) {        // SwiGLU: tile 0 = gate rows, tile 1 = up rows of the same 16 output columns
    const int oc = (n0 >> 1) + (lane >> 4) * 4;
    if (n0 >= p.N) return;
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(silu(rbf(v[0][r]))) * rbf(v[TILES - 1][r]));
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + oc) = o;
    return;
  }
  const int n = n0 + (lane >> 4) * 4;
  if (n >= p.N) return;
  float o4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o4[r] = rbf(v[0][r]);
  if (p.epi & A3V_EPI_RESIDUAL) {
    const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] += bf2f(rr[r]);
  }
  if (p.epi & A3V_EPI_OUT_F32) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = o4[r];
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = o;
  } else {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(o4[r]);
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n) = o;
  }
}


// ------------------------------------------------------------------------------------
// Decode GEMV, M <= 16, K % 128 == 0: W streamed ONCE from HBM by LDS-DMA.
//   * block = 4 waves, all on K-slice `sl` of row group `tg` (64 W rows); wave w owns the 16-row tile tg*4 + w.
//   * the A slice [AROWS][slice] is DMA'd once per block into LDS and shared by the 4 waves (AROWS = 8 when
//     M <= 8: MFMA columns m >= 8 read row m & 7 and are never stored), so activations cost no VGPR traffic.
//   * each wave streams its 16 rows through a private 2-stage ring of 16 rows x 128 k (4 KB = 4 DMA instructions
//     of 4 rows x 256 contiguous bytes); MFMA fragments by ds_read_b128, slot XOR row swizzle applied on the DMA
//     source side (conflict-free).  No block barrier in the K loop (the ring is wave-private).
//   * split-K across blocks: partial accumulators go to the workspace; the WAVE that arrives last at its tile's
//     (agent-scope) counter sums the S partials in slice order (deterministic), runs the epilogue and leaves the
//     counter at zero -- no block barrier, no second launch.  Blocks of one row group share an XCD (same L2).
// Measured (tools/ubench/skinny.hip, weights rotating through 3 GB): 5.0-5.5 TB/s vs 3.5-4.3 for the direct-to-VGPR form.
// ------------------------------------------------------------------------------------
struct GemvArgs {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const void* res;
  float* part;
  int* counters;
  int64_t lda, ldw, ldc, ldr;
  int M, N, K, epi, S, nkb, tgs, maxkb;
  // fused decode-step forms (a3v_gemv_fused): RMSNorm prologue, RoPE + KV-cache epilogue, sum-of-squares side output
  const bf16_t* norm_w;     // PRO: A holds the un-normalised rows h; the block normalises its K slice while staging it
  const float* ssq_in;      // PRO: [ssq_tiles][16] per-16-column partial sums of squares of the rows of A
  float* ssq_out;           // GEMV_EPI_SSQ: the same quantity for the rows this GEMV writes (residual stream)
  const float* cos_sin;     // GEMV_EPI_ROPEKV: fp32 [pos][hd/2][2]
  bf16_t* k_cache;          //   [M, Hkv, Smax, hd]
  bf16_t* vt_cache;         //   [M, Hkv, hd, Smax]
  const float* wscale;      // W8: per-row dequantisation scales (W rows are fp8 e4m3fn bytes, ldw in BYTES)
  float eps;
  int ssq_tiles, H, Hkv, hd, Smax, pos;
};

constexpr int GEMV_EPI_ROPEKV = 1 << 24;
constexpr int GEMV_EPI_SSQ = 1 << 25;

// Epilogue of the decode GEMVs for one finished 16-row tile (SwiGLU: one gate / up tile pair): v[r] = D[n = nt0 + 4 (lane>>4) + r][m = lane & 15]
// (u: the matching up-projection rows).  Shared by the split-K-across-blocks kernel above and the split-K-inside-the-block kernel below.
__device__ __forceinline__ void gemv_finish(const GemvArgs& p, f32x4 v, f32x4 u, int nt0, int lane, bool swiglu) {
  const int m = lane & 15;
  if (swiglu) {
    if (m >= p.M || nt0 >= p.N) return;
    const int oc = (nt0 >> 1) + (lane >> 4) * 4;
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(silu(rbf(v[r]))) * rbf(u[r]));
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + oc) = o;
    return;
  }
  const int n = nt0 + (lane >> 4) * 4;
  float o4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o4[r] = rbf(v[r]);
  if (p.epi & GEMV_EPI_ROPEKV) {
    // fused apply_rotary_emb + KV-cache write of the decode step (llama_ens5.py:118,124-129; a3v_rope_kvcache at S == 1):
    // rows n are [q heads | k heads | v heads] x hd; a lane holds two interleaved pairs of one head, batch row m.
    if (m >= p.M || n >= p.N) return;
    const int slot = n / p.hd, d = n % p.hd, half = p.hd >> 1;
    if (slot < p.H + p.Hkv) {
      const float* cs = p.cos_sin + ((int64_t)p.pos * half + (d >> 1)) * 2;
      const f32x4 t = *reinterpret_cast<const f32x4*>(cs);          // (cos, sin) of the two pairs
      bf16x4 o;
      o[0] = f2bf(o4[0] * t[0] - o4[1] * t[1]);
      o[1] = f2bf(o4[0] * t[1] + o4[1] * t[0]);
      o[2] = f2bf(o4[2] * t[2] - o4[3] * t[3]);
      o[3] = f2bf(o4[2] * t[3] + o4[3] * t[2]);
      bf16_t* dst = slot < p.H ? reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n
                               : p.k_cache + (((int64_t)m * p.Hkv + (slot - p.H)) * p.Smax + p.pos) * p.hd + d;
      *reinterpret_cast<bf16x4*>(dst) = o;
    } else {
      bf16_t* dst = p.vt_cache + (((int64_t)m * p.Hkv + (slot - p.H - p.Hkv)) * p.hd + d) * (int64_t)p.Smax + p.pos;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(int64_t)r * p.Smax] = f2bf(o4[r]);
    }
    return;
  }
  const bool live = m < p.M && n < p.N;
  if (live && (p.epi & A3V_EPI_RESIDUAL)) {
    const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] += bf2f(rr[r]);
  }
  if (p.epi & GEMV_EPI_SSQ) {
    // sum of squares of the 16 bf16 values this tile contributes to row m (consumed by the next GEMV's RMSNorm prologue)
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float hb = rbf(o4[r]); sq = fmaf(hb, hb, sq); }
    if (!live) sq = 0.f;
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (lane < 16 && nt0 < p.N) p.ssq_out[(nt0 >> 4) * 16 + lane] = sq;
  }
  if (!live) return;
  if (p.epi & A3V_EPI_OUT_F32) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = o4[r];
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = o;
  } else {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(o4[r]);
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + n) = o;
  }
}

// W8: the weight rows are OCP fp8 e4m3fn (weight-only quantisation, one fp32 scale per row applied to the summed
// accumulator).  A ring stage is still 16 rows x 256 B, i.e. 256 k instead of 128; fragments are read 8 B per lane and
// widened fp8 -> f32 -> bf16 in registers (exact), so the arithmetic is the bf16 MFMA on dequantised weights.
template <int AROWS, bool PRO, bool W8, int WAUX = 0>   // WAUX: cache-policy bits of the weight-stream LDS-DMA (2 = nt: streamed once)
__global__ __launch_bounds__(256) void gemv_dma_bf16_kernel(GemvArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char gemv_lds[];
  __shared__ float rinv_s[16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // block -> (row group, slice): all slices of a row group on one XCD (blockIdx % 8)
  const int xq = blockIdx.x >> 3;
  const int sl = xq % p.S;
  const int tg = (xq / p.S) * 8 + (blockIdx.x & 7);
  if (tg >= p.tgs) return;
  constexpr int APB = W8 ? 2 : 1;                      // 128-k blocks of A per ring stage of W
  const int nst_all = p.nkb / APB;
  const int st0 = (int)(((int64_t)sl * nst_all) / p.S), st1 = (int)(((int64_t)(sl + 1) * nst_all) / p.S);
  const int nst = st1 - st0;                           // ring stages of this slice
  const int kb0 = st0 * APB, nkb = nst * APB;          // ... in 128-k blocks of A
  constexpr int ABLK = AROWS * 256;                    // bytes of A per 128-k block
  char* Alds = gemv_lds;
  char* Wring = gemv_lds + p.maxkb * ABLK + wave * 2 * 4096;
  const int n0 = (tg * 4 + wave) * 16;
  const int dr = lane >> 4, dslot = lane & 15;
  if (!PRO) {
    constexpr int IPB = AROWS / 4;                     // DMA instructions per k-block
    for (int j = wave; j < nkb * IPB; j += 4) {
      const int kb = j / IPB, i = j % IPB;
      const int row = 4 * i + dr;
      const int ar = row < p.M ? row : p.M - 1;
      const bf16_t* src = p.A + (int64_t)ar * p.lda + (int64_t)(kb0 + kb) * 128 + ((dslot ^ row) & 15) * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(Alds + kb * ABLK + i * 1024), 16, 0, 0);
    }
  }
  const char* wrow[4];                                 // a stage row is 256 B in both weight formats
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * i + dr;
    int wr = n0 + row;
    wr = wr < p.N ? wr : p.N - 1;
    wrow[i] = reinterpret_cast<const char*>(p.W) + (int64_t)wr * p.ldw * (W8 ? 1 : 2) + (int64_t)st0 * 256 + ((dslot ^ row) & 15) * 16;
  }
  auto dma_stage = [&](int st, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow[i] + st * 256),
                                       (__attribute__((address_space(3))) void*)(Wring + slot * 4096 + i * 1024), 16, 0, WAUX);
  };
  if (PRO) {
    // RMSNorm of the block's K slice of A (model/components.py:39,52-53 rounding: fp32 x*rinv -> bf16 -> * weight -> bf16),
    // 1/rms from the producer's per-tile sums of squares.  All prologue loads (L2 hits) are issued BEFORE the weight
    // ring's first DMAs: memory returns in order, so the normalisation runs while the first weight stages are in flight.
    __shared__ float ssq_w[4][16];
    constexpr int NQ = AROWS / 4;                      // float4 per tile row of the ssq table
    f32x4 sq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) sq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = threadIdx.x; t < p.ssq_tiles; t += 256) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p.ssq_in + t * 16 + q * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) sq[q][r] += x[r];
      }
    }
    const int per_row = nkb * 16, total = p.M * per_row;
    constexpr int CH = 4;                              // items (16-B chunks of A) per thread per pass
    bf16x8 xa[CH], ga[CH];
    auto issue = [&](int base) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        int it = base + j * 256 + threadIdx.x;
        it = it < total ? it : total - 1;
        const int m = it / per_row, cc = it % per_row;
        const int64_t k = (int64_t)(kb0 + (cc >> 4)) * 128 + (cc & 15) * 8;
        xa[j] = *reinterpret_cast<const bf16x8*>(p.A + (int64_t)m * p.lda + k);
        ga[j] = *reinterpret_cast<const bf16x8*>(p.norm_w + k);
      }
    };
    auto finish = [&](int base) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int it = base + j * 256 + threadIdx.x;
        if (it < total) {
          const int m = it / per_row, cc = it % per_row;
          const int kb = cc >> 4, c = cc & 15;
          const float ri = rinv_s[m];
          bf16x8 y;
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = f2bf(rbf((float)xa[j][e] * ri) * (float)ga[j][e]);
          *reinterpret_cast<bf16x8*>(Alds + kb * ABLK + m * 256 + ((c ^ m) & 15) * 16) = y;
        }
      }
    };
    issue(0);
    dma_stage(0, 0);
    if (nst > 1) dma_stage(1, 1);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = wave_sum(sq[q][r]);
        if (lane == 0) ssq_w[wave][q * 4 + r] = t;
      }
    __syncthreads();
    if (threadIdx.x < AROWS)
      rinv_s[threadIdx.x] = rsqrtf((ssq_w[0][threadIdx.x] + ssq_w[1][threadIdx.x] + ssq_w[2][threadIdx.x] + ssq_w[3][threadIdx.x]) / (float)p.K + p.eps);
    __syncthreads();
    finish(0);
    for (int base = CH * 256; base < total; base += CH * 256) {
      issue(base);
      finish(base);
    }
  } else {
    dma_stage(0, 0);
    if (nst > 1) {
      dma_stage(1, 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // in-order completion: the A pieces issued before the ring prologue
    } else {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
  }
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  const int arow = AROWS == 8 ? (fr & 7) : fr;
  int foff[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) foff[s4] = (((4 * s4 + fg) ^ fr) & 15) * 16;
  int aoff[4];
#pragma unroll
  // AROWS == 8: lanes fr >= 8 feed accumulator columns 8..15, which are never stored.  Pointing them at the SAME address as lane
  // fr - 8 made every A fragment read a 2-way bank conflict (b128 reads do not merge duplicates: PMC conflict cycles = 2x the
  // active LDS cycles); with the lane's own fr in the swizzle they read the other half-row of the same row instead (any finite
  // data will do) and the 16 lanes of a group cover 16 distinct slots.
  for (int s4 = 0; s4 < 4; ++s4) aoff[s4] = (((4 * s4 + fg) ^ fr) & 15) * 16;
  for (int st = 0; st < nst; ++st) {
    const int slot = st & 1;
    if (st + 2 <= nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* Ws = Wring + slot * 4096 + fr * 256;
    if (!W8) {
      const char* As = Alds + st * ABLK + arow * 256;
      bf16x8 wf[4], af[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        wf[s4] = *reinterpret_cast<const bf16x8*>(Ws + foff[s4]);
        af[s4] = *reinterpret_cast<const bf16x8*>(As + aoff[s4]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments are in registers: the slot may be overwritten
      if (st + 2 < nst) dma_stage(st + 2, slot);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4], acc0, 0, 0, 0);
      }
    } else {
      // 8 MFMA steps per stage: step s8 covers k = 32 s8 .. 32 s8 + 31; the lane's 8 bytes sit in 16-B chunk 2 s8 + (fg >> 1)
      u32x2 wq[8];
      bf16x8 af[8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        wq[s8] = *reinterpret_cast<const u32x2*>(Ws + (((2 * s8 + (fg >> 1)) ^ fr) & 15) * 16 + (fg & 1) * 8);
        af[s8] = *reinterpret_cast<const bf16x8*>(Alds + (st * 2 + (s8 >> 2)) * ABLK + arow * 256 + aoff[s8 & 3]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (st + 2 < nst) dma_stage(st + 2, slot);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        bf16x8 wf;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          // (round 5) v_cvt_scalef32_pk_bf16_fp8 (gfx950): two e4m3 bytes -> two bf16 in ONE VALU op (scale 1: exact, as the fp8 -> f32
          // -> bf16 pair of ops it replaces; A3V_W8_CVT_F32 builds keep those for the A/B)
#ifdef A3V_W8_CVT_F32
          const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)wq[s8][h2], false);
          const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)wq[s8][h2], true);
          wf[4 * h2 + 0] = f2bf(lo[0]); wf[4 * h2 + 1] = f2bf(lo[1]); wf[4 * h2 + 2] = f2bf(hi[0]); wf[4 * h2 + 3] = f2bf(hi[1]);
#else
          const bf16x2 lo = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wq[s8][h2], 1.0f, false);
          const bf16x2 hi = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wq[s8][h2], 1.0f, true);
          wf[4 * h2 + 0] = lo[0]; wf[4 * h2 + 1] = lo[1]; wf[4 * h2 + 2] = hi[0]; wf[4 * h2 + 3] = hi[1];
#endif
        }
        if (s8 & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[s8], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[s8], acc0, 0, 0, 0);
      }
    }
  }
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc0[r] + acc1[r];
  const bool swiglu = (p.epi & A3V_EPI_SWIGLU) != 0;     // launcher guarantees S > 1 with SwiGLU
  f32x4 u = {0.f, 0.f, 0.f, 0.f};
  int nt0 = n0;                                          // first W row of the tile this wave finishes
  if (p.S > 1) {
    // Wave-granular split-K fix-up, no block barrier: the wave writes its partial accumulator (sc0 sc1 = agent-coherent
    // access, no cache-wide write-back / invalidate), waits for the acknowledge, then bumps the arrival counter of its
    // tile (of its gate/up tile PAIR with SwiGLU).  The wave that arrives last reloads all partials in one round trip,
    // sums them in slice order (bit-identical whichever wave is last), resets the counter and runs the epilogue.
#ifdef GEMV_NO_FIXUP   // timing build (wrong results): slice 0 finishes on its own partial, nobody stores or waits: 3.85 -> 3.43 ms per decode step
    if (sl != 0) return;
#else
    float* mine = p.part + (((int64_t)(tg * p.S + sl) * 4 + wave) * 64 + lane) * 4;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(mine), "v"(v) : "memory");   // (s_nop: the store's data registers, see the V^T store of the fused-qkv epilogue)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int* ctr = p.counters + (swiglu ? tg * 2 + (wave >> 1) : tg * 4 + wave);
    const int expect = swiglu ? 2 * p.S : p.S;
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != expect - 1) return;
    if (lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int w0 = swiglu ? (wave & ~1) : wave;          // gate tile (or own tile)
    nt0 = (tg * 4 + w0) * 16;
    const float* base = p.part + (((int64_t)tg * p.S * 4 + w0) * 64 + lane) * 4;
    f32x4 x[8], y[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const float* src = base + (int64_t)(s8 < p.S ? s8 : p.S - 1) * 4 * 64 * 4;
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(x[s8]) : "v"(src) : "memory");
    }
    if (swiglu) {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const float* src = base + 64 * 4 + (int64_t)(s8 < p.S ? s8 : p.S - 1) * 4 * 64 * 4;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(y[s8]) : "v"(src) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])::"memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"memory");
    v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
      if (s8 < p.S) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += x[s8][r];
        if (swiglu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] += y[s8][r];
        }
      }
#endif
  }
  if (W8) {                 // per-row dequantisation scale on the summed accumulator (rows clamp: the stores are masked)
    const int nr = nt0 + (lane >> 4) * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + (nr + 4 <= p.N ? nr : 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= sc[r];
    if (swiglu) {
      const f32x4 su = *reinterpret_cast<const f32x4*>(p.wscale + (nr + 20 <= p.N ? nr + 16 : 0));
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] *= su[r];
    }
  }
  gemv_finish(p, v, u, nt0, lane, swiglu);
}

// ------------------------------------------------------------------------------------
// Decode GEMV with the K split INSIDE the block (round 4; M <= 8, bf16 weights): the block owns ONE 16-row tile of W (with SwiGLU a
// gate / up tile pair) and its waves are the K slices -- the same number of waves streaming the same 16 rows x (K / slices) through
// the same private 2-stage rings as gemv_dma_bf16_kernel, but the partial accumulators meet in LDS behind one block barrier instead
// of in HBM behind store -> acknowledge -> arrival counter -> reload (three dependent memory round trips per launch: 0.42 ms of the
// 3.85-ms decode step, profiles/r04g_decode_fixup_and_rope_epilogue.txt).  What the shared LDS slice of A gave up for that: a wave
// reads ITS K slice of the 8 activation rows straight from L2 into registers in MFMA operand layout (4 x 16 B per lane and stage,
// two stages ahead; with the RMSNorm prologue the norm weights the same way and the normalisation in registers), so a block needs
// only its rings + a 1-KiB reduce patch per wave and 5-6 blocks fit a CU.
// ------------------------------------------------------------------------------------
template <bool PRO>
__global__ __launch_bounds__(768) void gemv_kq_bf16_kernel(GemvArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char kq_lds[];
  __shared__ float rinv_s[8];
  __shared__ float ssq_w[12][8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
  const bool swiglu = (p.epi & A3V_EPI_SWIGLU) != 0;
  // tile and K slice of this wave
  const int tiles_pb = swiglu ? 2 : 1;
  const int slices = nw / tiles_pb;
  const int tile = blockIdx.x * tiles_pb + (swiglu ? (wave & 1) : 0);
  const int sl = swiglu ? (wave >> 1) : wave;
  const int n0 = tile * 16;
  const int st0 = (int)(((int64_t)sl * p.nkb) / slices), st1 = (int)(((int64_t)(sl + 1) * p.nkb) / slices);
  const int nst = st1 - st0;
  char* Wring = kq_lds + wave * 2 * 4096;
  float* red = reinterpret_cast<float*>(kq_lds + nw * 2 * 4096);          // [nw][64 lanes][4] partial accumulators
  const int dr = lane >> 4, dslot = lane & 15;
  const char* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * i + dr;
    int wr = n0 + row;
    wr = wr < p.N ? wr : p.N - 1;
    wrow[i] = reinterpret_cast<const char*>(p.W) + (int64_t)wr * p.ldw * 2 + (int64_t)st0 * 256 + ((dslot ^ row) & 15) * 16;
  }
  auto dma_stage = [&](int st, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow[i] + st * 256),
                                       (__attribute__((address_space(3))) void*)(Wring + slot * 4096 + i * 1024), 16, 0, 2);
  };
  // A fragments (B operand): lane (fr = lane & 15 -> activation row fr & 7, fg = lane >> 4): k = 128 stage + 32 s4 + 8 fg + 0..7
  const int fr = lane & 15, fg = lane >> 4;
  const int arow = (fr & 7) < p.M ? (fr & 7) : p.M - 1;
  const bf16_t* arp = p.A + (int64_t)arow * p.lda + (int64_t)st0 * 128 + fg * 8;
  const bf16_t* gp = PRO ? p.norm_w + (int64_t)st0 * 128 + fg * 8 : nullptr;
  bf16x8 ax[2][4], ag[2][4];
  auto load_a = [&](int st, int set) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      ax[set][s4] = *reinterpret_cast<const bf16x8*>(arp + st * 128 + s4 * 32);
      if (PRO) ag[set][s4] = *reinterpret_cast<const bf16x8*>(gp + st * 128 + s4 * 32);
    }
  };
  if (nst > 0) { load_a(0, 0); dma_stage(0, 0); }
  if (nst > 1) { load_a(1, 1); dma_stage(1, 1); }
  float ri = 1.f;
  if (PRO) {
    // 1/rms of the 8 rows from the producer's per-16-column sums of squares (model/components.py:39,52-53); issued behind the first
    // stages, so the reduction runs while they are in flight
    f32x4 sq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int t = threadIdx.x; t < p.ssq_tiles; t += blockDim.x) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p.ssq_in + t * 16 + q * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) sq[q][r] += x[r];
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = wave_sum(sq[q][r]);
        if (lane == 0) ssq_w[wave][q * 4 + r] = t;
      }
    __syncthreads();
    if (threadIdx.x < 8) {
      float t = 0.f;
      for (int w = 0; w < nw; ++w) t += ssq_w[w][threadIdx.x];
      rinv_s[threadIdx.x] = rsqrtf(t / (float)p.K + p.eps);
    }
    __syncthreads();
    ri = rinv_s[fr & 7];
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int foff[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) foff[s4] = (((4 * s4 + fg) ^ fr) & 15) * 16;
  auto stage_body = [&](int st, auto setc) {
    constexpr int SET = decltype(setc)::value;
    // W(st) and A(st) have landed when at most the next stage's pieces are outstanding (in-order completion)
    if (st + 1 < nst) { if (PRO) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* Ws = Wring + SET * 4096 + fr * 256;
    bf16x8 wf[4], af[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) wf[s4] = *reinterpret_cast<const bf16x8*>(Ws + foff[s4]);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if (PRO) {
#pragma unroll
        for (int e = 0; e < 8; ++e) af[s4][e] = f2bf(rbf((float)ax[SET][s4][e] * ri) * (float)ag[SET][s4][e]);
      } else {
        af[s4] = ax[SET][s4];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments are in registers: the slot may be overwritten
    if (st + 2 < nst) { load_a(st + 2, SET); dma_stage(st + 2, SET); }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if (s4 & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4], acc0, 0, 0, 0);
    }
  };
  {
    int st = 0;
    for (; st + 1 < nst; st += 2) {
      stage_body(st, std::integral_constant<int, 0>{});
      stage_body(st + 1, std::integral_constant<int, 1>{});
    }
    if (st < nst) stage_body(st, std::integral_constant<int, 0>{});
  }
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = acc0[r] + acc1[r];
  f32x4 u = {0.f, 0.f, 0.f, 0.f};
  if (nw > 1) {
    // the K slices meet in LDS: the tile's first wave sums them in slice order (deterministic) and finishes the tile
    *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = v;
    __syncthreads();
    if (wave != 0) return;
    v = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s8 = 0; s8 < slices; ++s8) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(red + ((s8 * tiles_pb) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += x[r];
      if (swiglu) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(red + ((s8 * 2 + 1) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] += y[r];
      }
    }
  }
  gemv_finish(p, v, u, blockIdx.x * tiles_pb * 16, lane, swiglu);
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

// environment switches: see A3V_ENV_INT (a3v_common.h)
static int g_env_generation = 0;
int a3v_env_generation() { return g_env_generation; }
extern "C" int a3v_reload_env(void) { return ++g_env_generation; }
// bit 0: built with -DA3V_EXPERIMENTS (the measured-and-not-dispatched GEMM kernels and their switches are present)
extern "C" int a3v_build_flags(void) {
#ifdef A3V_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

// Optional scratch for the split-K forms of the hybrid dispatch (tail rows, few-tile problems): the library never allocates, so without
// it those rows run as plain launches.  Registrations are keyed by (device, stream): two streams that run GEMMs concurrently must not
// share split-K planes (round 3 kept ONE process-global pointer: a second stream, model or device raced through it silently).
//   a3v_gemm_set_workspace_for(stream, ptr, bytes)   the scratch of GEMM calls issued on `stream` of the current device
//   a3v_gemm_set_workspace(ptr, bytes)               legacy form: a scratch for callers that never name a stream; it is BOUND to the
//                                                    first (device, stream) that uses it -- any other stream without a registration
//                                                    of its own gets none (plain launches), never somebody else's planes
#include <mutex>
#include <vector>
namespace {
struct GemmWs { float* p; int64_t bytes; };
struct GemmWsEntry { int dev; hipStream_t st; float* p; int64_t bytes; };
std::mutex g_ws_mu;
std::vector<GemmWsEntry> g_ws_tab;
GemmWsEntry g_ws_legacy = {-1, nullptr, nullptr, 0};
bool g_ws_legacy_bound = false;
GemmWs gemm_ws_for(hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (const GemmWsEntry& e : g_ws_tab)
    if (e.dev == dev && e.st == st) return {e.p, e.bytes};
  if (g_ws_legacy.p) {
    if (!g_ws_legacy_bound) { g_ws_legacy.dev = dev; g_ws_legacy.st = st; g_ws_legacy_bound = true; }
    if (g_ws_legacy.dev == dev && g_ws_legacy.st == st) return {g_ws_legacy.p, g_ws_legacy.bytes};
  }
  return {nullptr, 0};
}
}  // namespace
extern "C" int a3v_gemm_set_workspace_for(void* stream, void* ptr, int64_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (size_t i = 0; i < g_ws_tab.size(); ++i)
    if (g_ws_tab[i].dev == dev && g_ws_tab[i].st == (hipStream_t)stream) {
      if (ptr) { g_ws_tab[i].p = (float*)ptr; g_ws_tab[i].bytes = bytes; } else g_ws_tab.erase(g_ws_tab.begin() + i);
      return A3V_OK;
    }
  if (ptr) g_ws_tab.push_back({dev, (hipStream_t)stream, (float*)ptr, bytes});
  return A3V_OK;
}
extern "C" int a3v_gemm_set_workspace(void* ptr, int64_t bytes) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  g_ws_legacy = {-1, nullptr, (float*)ptr, ptr ? bytes : 0};
  g_ws_legacy_bound = false;
  return A3V_OK;
}

namespace {
// sum of S raw fp32 planes [M][N] -> rounded once to bf16 ("the value F.linear returns") -> residual / output forms of gemm_epilogue
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part, int S, int64_t plane, int M, int N, void* __restrict__ C,
                                                              int64_t ldc, const void* __restrict__ res, int64_t ldr, int epi,
                                                              const bf16_t* __restrict__ bias = nullptr, float* __restrict__ sumsq = nullptr) {
  const int64_t n4 = (int64_t)M * (N / 4);
  float ss = 0.f;                                        // sumsq: squares of the fp32 values this block stores -> slot blockIdx.x
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / (N / 4)), c = (int)(i % (N / 4)) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(part + (int64_t)s * plane + (int64_t)r * N + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += x[e];
    }
    if (bias) {
      const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += bf2f(b4[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = rbf(a[e]);
    if (epi & A3V_EPI_RES_F32) {
      const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(res) + (int64_t)r * ldr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += rr[e];
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + (int64_t)r * ldc + c) = a;
#pragma unroll
      for (int e = 0; e < 4; ++e) ss = fmaf(a[e], a[e], ss);
      continue;
    }
    if (epi & A3V_EPI_RESIDUAL) {
      const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(res) + (int64_t)r * ldr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += bf2f(rr[e]);
    }
    if (epi & A3V_EPI_OUT_F32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + (int64_t)r * ldc + c) = a;
#pragma unroll
      for (int e = 0; e < 4; ++e) ss = fmaf(a[e], a[e], ss);
    } else {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(a[e]);
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(C) + (int64_t)r * ldc + c) = o;
    }
  }
  if (sumsq) {
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) sumsq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}
}  // namespace

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v & ~7;
    if (n <= 0) n = 256;
  }
  return n;
}
static bool pp_persistent() { return A3V_ENV_INT("A3V_GEMM_PERSISTENT", 1) != 0; }   // 0: one block per tile (A/B runs)
static int nt_store_env() { return A3V_ENV_INT("A3V_GEMM_NT_STORE", 0); }
static int slow_epi_env() { return A3V_ENV_INT("A3V_GEMM_FAST_EPI", 1) == 0 ? 1 : 0; }   // 0: every tile through the general epilogue

#ifdef A3V_EXPERIMENTS
static bool pp_ring() { return A3V_ENV_INT("A3V_GEMM_RING", 1) != 0; }       // 0: the two-stage ping-pong kernel
static unsigned* xsync_buffer(hipStream_t st) {   // A3V_GEMM_LOCKSTEP=1: eight counters, zeroed on the launch stream before every launch
  if (!A3V_ENV_INT("A3V_GEMM_LOCKSTEP", 0)) return nullptr;
  static unsigned* buf = nullptr;
  if (!buf && hipMalloc(&buf, 64) != hipSuccess) { buf = nullptr; return nullptr; }
  if (hipMemsetAsync(buf, 0, 64, st) != hipSuccess) return nullptr;
  return buf;
}
static int w4_env() { return A3V_ENV_INT("A3V_GEMM_W4", 0); }   // 1: one-wave-per-SIMD kernel, 20: overlapped form, ... (tools/gemm_w4_ab.py)
#else
static bool pp_ring() { return true; }
#endif

template <bool A_ROWS>
static void launch_tn(dim3 grid, hipStream_t st, const GemmArgs& q0) {
  GemmArgs q = q0;
  q.slow_epi = slow_epi_env(); q.nt_store = nt_store_env();
  q.xmap = A3V_ENV_INT("A3V_GEMM_XMAP_TN", 1);   // =0: one contiguous run of tiles per XCD (A/B); 1: round-major; +2: serpentine; +4: 16 x 16 super-tiles where they fit
  if (!((q.xmap & 4) && (q.tiles_m & 15) == 0 && (q.tiles_n & 15) == 0)) q.xmap &= ~4;
  hipLaunchKernelGGL(gemm_tn_bf16_pp_kernel<A_ROWS>, grid, dim3(512), 0, st, q);
}

static int gemm_nt_impl(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                        int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                        int epilogue, int dtype, void* stream, const RopeKvArgs* rk) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !C) return A3V_ERR_ARG;
  if ((epilogue & A3V_EPI_BIAS) && !bias) return A3V_ERR_ARG;
  if ((epilogue & (A3V_EPI_RESIDUAL | A3V_EPI_RES_F32)) && !residual) return A3V_ERR_ARG;
  if (epilogue & A3V_EPI_SWIGLU_BWD) {     // alone (no bias / activation / residual kinds), bf16, gate / up rows in `residual`
    if (!residual || dtype != A3V_BF16 || (epilogue & 0xffff & ~A3V_EPI_SWIGLU_BWD)) return A3V_ERR_ARG;
    if ((N % 8) || (ldr % 4) || ldr < 2 * (int64_t)N || ldc < 2 * (int64_t)N) return A3V_ERR_SHAPE;
  }
  hipStream_t st = (hipStream_t)stream;
  const GemmWs gws = gemm_ws_for(st);
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
  GemmArgs p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C; p.bias = bias; p.res = residual;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.epi = epilogue & 0xffff;
  p.dbg = (epilogue >> 24) & 0xf;
  p.c_split = 0;
  p.rk = RopeKvArgs{};
  if (rk) { p.rk = *rk; p.epi |= GEMM_EPI_ROPEKV; }
  // Tile choice.  256x256 ping-pong (8 waves, 1 block/CU) for the rows that fill whole 256-row
  // tiles when its grid keeps the 256 CUs busy (>= 75 % of its last round); the remaining (< 256)
  // rows, and every problem the big tile would quantise badly, go to the 128x128 kernel (4 waves,
  // 2 blocks/CU).  A3V_EPI_TILE_* force one configuration for the whole problem (tuning / tests).
  auto launch = [&](int cfg, GemmArgs q) {
    q.slow_epi = slow_epi_env(); q.nt_store = nt_store_env();
    if (cfg == 128) {
      q.tiles_m = (q.M + 127) / 128; q.tiles_n = (q.N + 127) / 128;
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128, 2, 2>), dim3(q.tiles_m * q.tiles_n), dim3(256), 0, st, q);
      return;
    }
    const int tbm = cfg == 259 ? 192 : 256;
    q.tiles_m = (q.M + tbm - 1) / tbm; q.tiles_n = (q.N + 255) / 256;
    const int nt = q.tiles_m * q.tiles_n;
    // the ping-pong kernel is persistent: one block per CU walks its tiles (A3V_GEMM_PERSISTENT=0: one block per tile, for A/B runs)
    const dim3 g((cfg == 257 || cfg == 259) && pp_persistent() ? std::min(nt, cu_count()) : nt), b(512);
    if (cfg == 256) { hipLaunchKernelGGL((gemm_nt_bf16_kernel<256, 256, 2, 4>), g, b, 0, st, q); return; }
    q.xmap = (g.x & 63) ? 0 : A3V_ENV_INT("A3V_GEMM_XMAP", 1);   // =0: one contiguous run of tiles per XCD (A/B)
    if (!((q.xmap & 4) && g.x == 256 && (q.tiles_m & 15) == 0 && (q.tiles_n & 15) == 0)) q.xmap &= ~4;
    q.skew = 0; q.xsync = nullptr;
#ifdef A3V_EXPERIMENTS
    if (cfg == 258) { hipLaunchKernelGGL(gemm_nt_bf16_pp32_kernel<0>, g, b, 0, st, q); return; }
    {
      int dbg = q.dbg;
      if (dbg == 0 && pp_ring()) dbg = 5;
      q.skew = A3V_ENV_INT("A3V_GEMM_SKEW", 0);
      q.xsync = (dbg == 5 && g.y == 1) ? xsync_buffer(st) : nullptr;
      if (dbg != 5 || w4_env() != 0) {
        switch (dbg) {
          case 0: hipLaunchKernelGGL((gemm_nt_bf16_pp_kernel<0, 0>), g, b, 0, st, q); break;
          case 5:
            if (q.epi & GEMM_EPI_ROPEKV) hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_ROPE>), g, b, 0, st, q);
            else if (q.epi & (A3V_EPI_BIAS | A3V_EPI_GELU | A3V_EPI_QUICKGELU))
              hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON | EPI_SET_PRE>), g, b, 0, st, q);
            else if (w4_env() == 1) hipLaunchKernelGGL((gemm_nt_bf16_w4_kernel<0>), g, dim3(256), 0, st, q);
            else if (w4_env() == 2) hipLaunchKernelGGL((gemm_nt_bf16_w4_kernel<4>), g, dim3(256), 0, st, q);   // cycle stamps into `bias`
            else if (w4_env() == 8) hipLaunchKernelGGL((gemm_nt_bf16_w4_kernel<8>), g, dim3(256), 0, st, q);
            else if (w4_env() == 9) hipLaunchKernelGGL((gemm_nt_bf16_w4_kernel<9>), g, dim3(256), 0, st, q);
            else if (w4_env() == 10) hipLaunchKernelGGL((gemm_nt_bf16_w4_kernel<10>), g, dim3(256), 0, st, q);   // every k-loop piece out of bounds
            else if (w4_env() == 7) hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<7, false, 0>), g, b, 0, st, q);   // ring kernel, cache-resident operands (timing experiment, wrong results)
            else if (w4_env() == 20) hipLaunchKernelGGL((gemm_nt_bf16_ov_kernel<0>), g, b, 0, st, q);            // overlapped 8-wave form
            else if (w4_env() == 28) hipLaunchKernelGGL((gemm_nt_bf16_ov_kernel<8>), g, b, 0, st, q);
            else if (w4_env() == 29) hipLaunchKernelGGL((gemm_nt_bf16_ov_kernel<9>), g, b, 0, st, q);
            else if (w4_env() == 30) hipLaunchKernelGGL((gemm_nt_bf16_ov_kernel<10>), g, b, 0, st, q);
            else if (w4_env() == 31) hipLaunchKernelGGL((gemm_nt_bf16_ov_kernel<11>), g, b, 0, st, q);
            else if (w4_env() == 32) hipLaunchKernelGGL((gemm_nt_bf16_ov_kernel<12>), g, b, 0, st, q);
            else if (w4_env() == 33) hipLaunchKernelGGL((gemm_nt_bf16_ov_kernel<13>), g, b, 0, st, q);
            else hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0>), g, b, 0, st, q);
            break;
          case 11: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, false>), g, b, 0, st, q); break;   // ring, direct (unstaged) epilogue stores
          case 12: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON, true>), g, b, 0, st, q); break;   // ring, group 0 waits for its W half at the end of L
          case 13: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<4, false, 0, true, EPI_SET_COMMON, true>), g, b, 0, st, q); break;   // ... with cycle stamps
          case 14: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 1, true, EPI_SET_COMMON, true>), g, b, 0, st, q); break;   // ... + barrier 1 tile row before the last MFMA
          case 15: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 2, true, EPI_SET_COMMON, true>), g, b, 0, st, q); break;   // ... 2 tile rows
          case 8: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 4, true, EPI_SET_COMMON, true>), g, b, 0, st, q); break;    // ... 4 tile rows
          case 7: hipLaunchKernelGGL((gemm_nt_bf16_pp_kernel<0, 0>), g, b, 0, st, q); break;   // two-stage kernel, for A/B runs
          case 9: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, true, 0>), g, b, 0, st, q); break;   // ring, 32x32x16 MFMA
          case 10: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<4, true, 0>), g, b, 0, st, q); break;   // 32x32x16, stamps
          case 6: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<4, false, 0>), g, b, 0, st, q); break;   // cycle stamps (tools/ring_stamps.py)
          case 1: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON, false, false>), g, b, 0, st, q); break;   // ablation: the wait after the MFMAs (round-2 form)
          case 2: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<4, false, 0, true, EPI_SET_COMMON, false, false>), g, b, 0, st, q); break;   // same, stamps
          case 3: hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON, false, true, false>), g, b, 0, st, q); break;   // ablation: prologue burst + block barrier per tile (no cross-tile DMA stream)
          default: break;
        }
        return;
      }
    }
#endif
    // the product path: the ring kernel, instantiated per set of fast epilogue forms
    if (cfg == 259) {                                    // 192 x 256 tiles (the fused-qkv form has no instantiation: general epilogue there)
      if (q.epi & (A3V_EPI_BIAS | A3V_EPI_GELU | A3V_EPI_QUICKGELU))
        hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON | EPI_SET_PRE, false, true, true, 192>), g, b, 0, st, q);
      else hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON, false, true, true, 192>), g, b, 0, st, q);
      return;
    }
    if (q.epi & GEMM_EPI_ROPEKV) hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_ROPE>), g, b, 0, st, q);
    else if (q.epi & (A3V_EPI_BIAS | A3V_EPI_GELU | A3V_EPI_QUICKGELU))
      hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON | EPI_SET_PRE>), g, b, 0, st, q);
    else hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0>), g, b, 0, st, q);
  };
  const int64_t bytesA = ((int64_t)(M - 1) * lda + K) * 2, bytesW = ((int64_t)(N - 1) * ldw + K) * 2;
  const bool desc_ok = bytesA < (1LL << 31) && bytesW < (1LL << 31);   // buffer descriptors: 32-bit offsets
  if (epilogue & (A3V_EPI_TILE_256PP32 | A3V_EPI_TILE_256PP | A3V_EPI_TILE_256 | A3V_EPI_TILE_128 | A3V_EPI_TILE_192PP)) {
    int cfg = 128;
    if (epilogue & A3V_EPI_TILE_192PP) cfg = 259;
    else if (epilogue & A3V_EPI_TILE_256PP32) cfg = 258;
    else if (epilogue & A3V_EPI_TILE_256PP) cfg = 257;
    else if (epilogue & A3V_EPI_TILE_256) cfg = 256;
    if (cfg > 256 && !desc_ok) return A3V_ERR_SHAPE;
    launch(cfg, p);
  } else {
    // Cost model in units of one 256x256 tile's time on one CU (T256): a "round" is 256 concurrent big tiles or
    // 512 concurrent 128x128 tiles (2 blocks/CU, each ~0.65 T256 at the small kernel's lower rate); a second launch
    // costs ~0.1.  Candidates: small kernel for everything, big kernel for everything, or big kernel on the M-tile rows
    // that fill whole rounds + small kernel on the remaining rows.
    const long tn256 = (N + 255) / 256, tn128 = (N + 127) / 128;
    const bool eligible = desc_ok && M >= 512 && N >= 512;
    // (round 2 recalibration, tools/hybrid_vs_ring.py: with the ring kernel at 1.35-1.45 PF and the small kernel at ~0.6 PF a round
    // of 512 small tiles costs ~1.15 T256, and a split-K tail adds its reduce pass: the 7B qkv shape (6.56 rounds) went to the
    // hybrid under the old 0.65 and lost 11 % to the all-ring launch)
    auto small_cost = [&](long rows) { return rows <= 0 ? 0.0 : 1.15 * std::max(0.5, (double)((rows + 127) / 128) * tn128 / 512.0); };
    const double round_us = 12.0 + 0.0206 * K;          // one round of 256 tiles of 256 x 256 on the ring kernel (M = 8192, N = 4096: 98 us at K = 4096, 476 at 22016)
    const double c_small = small_cost(M);
    const double c_big = eligible ? (double)((((long)(M + 255) / 256) * tn256 + 255) / 256) : 1e30;
    long mt_h = ((long)(M / 256) * tn256 / 256) * 256 / tn256;          // M-tile rows that make whole rounds
    double c_hyb = 1e30;
    if (eligible && mt_h >= 1 && mt_h * 256 < M) {
      // tail rows: on the ring kernel split over K when that fills the CUs (1/S of a tile time + the reduce pass), else small tiles
      const long tail_rows = M - mt_h * 256, tail_tiles = ((tail_rows + 255) / 256) * tn256;
      int S2 = tail_tiles > 0 ? (int)(cu_count() / tail_tiles) : 0;
      if (S2 > 8) S2 = 8;
      while (S2 > 1 && K / 64 < 8 * S2) --S2;
      const bool simple_epi = !(p.epi & ~(A3V_EPI_RESIDUAL | A3V_EPI_RES_F32 | A3V_EPI_OUT_F32)) && gws.p && N % 4 == 0;
      // (round 5 recalibration, tools/ring192_ab.py) the split-K tail costs its 1 / S of a tile time plus ~32 us that do not depend on K
      // (plane traffic + two launches): 0.30 of a round at K = 4096, 0.07 at K = 22016, where a round of 256 tiles takes ~12 + 0.0206 K us
      const double tail = (S2 >= 3 && simple_epi && pp_ring() && pp_persistent()) ? 1.0 / S2 + 32.0 / round_us : small_cost(tail_rows) + 0.25;
      c_hyb = (double)((mt_h * tn256 + 255) / 256) + tail;
    }
    // (round 5) the whole problem on 192 x 256 ring tiles: a round of them takes 0.79 of a 256 x 256 round (48 instead of 64 MFMAs per
    // wave and K-tile on 7 / 8 of the LDS-DMA pieces); 8728 x 4096 is 2.875 rounds of these against 2 rounds + a split-K tail
    double c_192 = 1e30;
    if (eligible && !rk && pp_ring() && pp_persistent() && A3V_ENV_INT("A3V_GEMM_RING_192", 1) != 0 && !(p.epi & A3V_EPI_SWIGLU))
      c_192 = 0.79 * (double)((((long)(M + 191) / 192) * tn256 + 255) / 256) + 0.02;
    // few big tiles (small N or M: the ViT's output projections, 76 tiles): the whole problem on the ring kernel split over K
    double c_spl = 1e30;
    int S3 = 0;
    {
      const long tiles_all = ((long)(M + 255) / 256) * tn256;
      S3 = tiles_all > 0 ? (int)(cu_count() / tiles_all) : 0;
      if (S3 > 8) S3 = 8;
      while (S3 > 1 && K / 64 < 8 * S3) --S3;
      const int okbits = A3V_EPI_BIAS | A3V_EPI_RESIDUAL | A3V_EPI_RES_F32 | A3V_EPI_OUT_F32;
      const bool on = A3V_ENV_INT("A3V_GEMM_RING_SPLIT", 1) != 0;
      if (on && eligible && S3 >= 3 && !(p.epi & ~okbits) && pp_ring() && pp_persistent() && gws.p && N % 4 == 0 &&
          (int64_t)S3 * M * N * 4 <= gws.bytes && (!(p.epi & A3V_EPI_BIAS) || !(reinterpret_cast<uintptr_t>(bias) & 7)))
        c_spl = 1.0 / S3 + 0.2;
    }
    if (c_192 < c_spl && c_192 < c_small && c_192 < c_big && c_192 < c_hyb) {
      launch(259, p);
    } else if (c_spl < c_small && c_spl < c_big && c_spl < c_hyb) {
      GemmArgs t = p;
      t.C = gws.p; t.ldc = N; t.res = nullptr; t.bias = nullptr;
      t.epi = A3V_EPI_OUT_F32 | GEMM_EPI_RAW;
      t.tiles_m = (M + 255) / 256; t.tiles_n = (int)tn256;
      t.c_split = (int64_t)M * N * 4;
      t.slow_epi = slow_epi_env(); t.nt_store = nt_store_env();
      hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0>), dim3(t.tiles_m * t.tiles_n, S3), dim3(512), 0, st, t);
      const int64_t n4 = (int64_t)M * (N / 4);
      const int rb = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(rb), dim3(256), 0, st, gws.p, S3, (int64_t)M * N, M, N, p.C, p.ldc, p.res, p.ldr,
                         p.epi & ~A3V_EPI_BIAS, (p.epi & A3V_EPI_BIAS) ? (const bf16_t*)p.bias : nullptr);
    } else if (c_big <= c_small && c_big <= c_hyb) {
      launch(257, p);
    } else if (c_hyb < c_small) {
      const int m_big = (int)(mt_h * 256);
      GemmArgs q = p;
      q.M = m_big;
      launch(257, q);
      GemmArgs r = p;
      r.M = M - m_big;
      r.A = p.A + (int64_t)m_big * lda;
      r.rk.m_off = m_big;
      const int esz = (p.epi & (A3V_EPI_OUT_F32 | A3V_EPI_RES_F32)) ? 4 : 2;
      r.C = (char*)p.C + (int64_t)m_big * ldc * esz;
      if (p.res) r.res = (const char*)p.res + (int64_t)m_big * ldr * ((p.epi & A3V_EPI_RES_F32) ? 4 : 2);
      // tail rows: a few hundred rows x N on 128x128 tiles = ~160 blocks with a serial K loop (50 us at K = 4096, 135 us at
      // K = 11008).  With a registered workspace the K loop is split into S planes (more blocks, 1/S the latency) and a
      // reduce pass applies the epilogue; only the plain / residual / fp32 forms are handled there.
      const int simple = A3V_EPI_RESIDUAL | A3V_EPI_RES_F32 | A3V_EPI_OUT_F32;
      int S = 1;
      const int tblocks = ((r.M + 127) / 128) * ((N + 127) / 128);
      while (tblocks * S < 512 && S < 8 && K / 64 >= 16 * S) S *= 2;
      // (round 2) the same tail on the ring kernel: its ceil(rows / 256) x tn256 big tiles split S ways over K so that they fill the
      // CUs once -- a fraction 1/S of a tile time instead of ~0.6 on the small kernel (wo: 257 -> ~235 us)
      const int big_tiles = (int)(((r.M + 255) / 256) * tn256);
      int S2 = big_tiles > 0 ? cu_count() / big_tiles : 0;
      if (S2 > 8) S2 = 8;
      while (S2 > 1 && K / 64 < 8 * S2) --S2;
      { const int e = A3V_ENV_INT("A3V_GEMM_TAIL_SLICES", 0); if (e >= 3 && e <= 16 && K / 64 >= 2 * e) S2 = e; }      // sweeps (tools/tail_cost.py)
      const bool ring_tail = A3V_ENV_INT("A3V_GEMM_RING_TAIL", 1) != 0;
      if (ring_tail && pp_ring() && pp_persistent() && S2 >= 3 && !(p.epi & ~simple) && gws.p && (int64_t)S2 * r.M * N * 4 <= gws.bytes && N % 4 == 0) {
        GemmArgs t = r;
        t.C = gws.p; t.ldc = N; t.res = nullptr; t.bias = nullptr;
        t.epi = A3V_EPI_OUT_F32 | GEMM_EPI_RAW;
        t.tiles_m = (t.M + 255) / 256; t.tiles_n = (int)tn256;
        t.c_split = (int64_t)t.M * N * 4;
        t.slow_epi = slow_epi_env(); t.nt_store = nt_store_env();
        t.skew = 0;
        hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0>), dim3(big_tiles, S2), dim3(512), 0, st, t);
        const int64_t n4 = (int64_t)r.M * (N / 4);
        const int rb = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(rb), dim3(256), 0, st, gws.p, S2, (int64_t)r.M * N, r.M, N, r.C, r.ldc, r.res, r.ldr, p.epi);
      } else if (S > 1 && !(p.epi & ~simple) && gws.p && (int64_t)S * r.M * N * 4 <= gws.bytes && N % 4 == 0) {
        GemmArgs t = r;
        t.C = gws.p; t.ldc = N; t.res = nullptr; t.bias = nullptr;
        t.epi = A3V_EPI_OUT_F32 | GEMM_EPI_RAW;
        t.tiles_m = (t.M + 127) / 128; t.tiles_n = (N + 127) / 128;
        t.c_split = (int64_t)t.M * N * 4;
        t.slow_epi = slow_epi_env(); t.nt_store = nt_store_env();
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128, 2, 2>), dim3(t.tiles_m * t.tiles_n, S), dim3(256), 0, st, t);
        const int64_t n4 = (int64_t)r.M * (N / 4);
        const int rb = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(rb), dim3(256), 0, st, gws.p, S, (int64_t)r.M * N, r.M, N, r.C, r.ldc, r.res, r.ldr, p.epi);
      } else {
        launch(128, r);
      }
    } else {
      launch(128, p);
    }
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                           int M, int N, int K, const void* bias, const void* residual, int64_t ldr,
                           int epilogue, int dtype, void* stream) {
  return gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, epilogue, dtype, stream, nullptr);
}

// qkv = A . Wqkv^T with the rotary embedding and the KV-cache write in the GEMM epilogue (prefill: LLM/llama_ens5.py:155-176
// xq, xk, xv = wq(x), wk(x), wv(x); apply_rotary_emb; cache_k / cache_v[:bsz, start_pos:start_pos+seqlen] = xk / xv).
// Same values as a3v_gemm_nt followed by a3v_rope_kvcache (the accumulator is rounded to the bf16 qkv value first).
extern "C" int a3v_gemm_qkv_rope(const void* A, int64_t lda, const void* W, int64_t ldw, int K, void* q_out, int64_t ldq,
                                 void* k_cache, void* vt_cache, void* v_rows, int64_t ldv, const void* delta, int64_t ldd,
                                 const float* cos_sin, int B, int S, int H, int Hkv, int hd, int Smax, int start_pos, int rope_pos0,
                                 void* stream) {
  if (!q_out || !k_cache || !vt_cache || !cos_sin || B <= 0 || S <= 0 || H <= 0 || Hkv <= 0) return A3V_ERR_ARG;
  if ((hd != 64 && hd != 128) || ldq % 4 || (v_rows && ldv % 4) || (delta && ldd % 4) || start_pos < 0 || start_pos + S > Smax) return A3V_ERR_SHAPE;
  RopeKvArgs rk;
  rk.q_out = (bf16_t*)q_out; rk.k_cache = (bf16_t*)k_cache; rk.vt_cache = (bf16_t*)vt_cache; rk.cos_sin = cos_sin;
  rk.v_rows = (bf16_t*)v_rows; rk.ldv = ldv;
  rk.ldq = ldq; rk.S = S; rk.H = H; rk.Hkv = Hkv; rk.hd_shift = hd == 128 ? 7 : 6; rk.Smax = Smax;
  rk.start_pos = start_pos; rk.rope_pos0 = rope_pos0; rk.m_off = 0;
  return gemm_nt_impl(A, lda, W, ldw, q_out, ldq, B * S, (H + 2 * Hkv) * hd, K, nullptr, delta, ldd, delta ? A3V_EPI_RESIDUAL : 0, A3V_BF16,
                      stream, &rk);
}

// split-K factor of the DMA GEMV: enough blocks (row groups x S >= A3V_GEMV_BLOCKS) for 256 CUs, S <= 8, S <= number of ring stages
static int gemv_split(int N, int K, bool w8) {
  const int tgs = (N + 63) / 64, nst = K / (w8 ? 256 : 128);
// block target of the split: 768 measured best in the decode bench of the 7B geometry (qkv: 4 slices instead of 8, LM head 2 instead
// of 4; +2-3 % decode tok/s, +6 % with fp8 weights; 640 / 896 / 1024 / 1536 all slower) -- sweep with -DA3V_GEMV_BLOCKS=n
#ifndef A3V_GEMV_BLOCKS
#define A3V_GEMV_BLOCKS 768
#endif
#ifndef A3V_GEMV_MAXS
#define A3V_GEMV_MAXS 8
#endif
  int S = 1;
  while (S < A3V_GEMV_MAXS && tgs * S < A3V_GEMV_BLOCKS && S * 2 <= nst) S *= 2;
  return S;
}

// can the LDS-DMA GEMV take this problem (else: the direct-to-VGPR kernel, which has no fused decode / fp8 forms)
static bool gemv_fits(int M, int N, int K, int epi, bool w8) {
  const int kst = w8 ? 256 : 128;
  if (K % kst || K < kst || N > 65536 || M > 16) return false;
  const int S = gemv_split(N, K, w8), nst = K / kst;
  const size_t ldsb = (size_t)((nst + S - 1) / S) * (kst / 128) * (M <= 8 ? 8 : 16) * 256 + 4 * 2 * 4096;
  return ldsb <= 150 * 1024 && !((epi & A3V_EPI_SWIGLU) && S == 1);
}
bool a3v_gemv_supported(int M, int N, int K, int epilogue, int w8) { return gemv_fits(M, N, K, epilogue, w8 != 0); }

// fills the split-K plan + workspace pointers of `g` and launches; false when the shape needs the direct-to-VGPR kernel
static bool gemv_launch(GemvArgs& g, void* ws, hipStream_t st) {
  const bool w8 = g.wscale != nullptr;
  if (!gemv_fits(g.M, g.N, g.K, g.epi, w8)) return false;
  g.counters = (int*)ws;
  g.part = (float*)((char*)ws + A3V_WS_PARTIALS);
  const int kst = w8 ? 256 : 128;
  g.S = gemv_split(g.N, g.K, w8); g.nkb = g.K / 128; g.tgs = (g.N + 63) / 64;
  g.maxkb = ((g.K / kst + g.S - 1) / g.S) * (kst / 128);
  const int arows = g.M <= 8 ? 8 : 16;
  const size_t ldsb = (size_t)g.maxkb * arows * 256 + 4 * 2 * 4096;
  const int blocks = ((g.tgs + 7) / 8) * 8 * g.S;
  const bool pro = g.norm_w != nullptr;
  // M <= 8, bf16 weights: the K slices as the waves of ONE block per 16-row tile (gemv_kq_bf16_kernel: the partials meet in LDS,
  // no split-K fix-up through HBM); same slice count as the across-blocks plan.  A3V_GEMV_KQ=0: the across-blocks kernel (A/B runs)
  // A3V_GEMV_KQ: 1 (default) = where it was measured to win: no RMSNorm prologue, no SwiGLU, and the 16-row tiles spread EVENLY over the
  // CUs (one or two blocks each: 7B wo / w2 = 256 tiles: 11.5 -> 10.4 us, 19.4 -> 19.6) -- a block of 8 waves is a coarse unit, 320
  // tiles (13B wo / w2) leave a quarter of the CUs with twice the work (15.1 -> 18.0 us, 32.0 -> 38.8: tools/ab_gemv_kq.py);
  // 2 = every bf16 GEMV with M <= 8 (A/B runs); 0 = none
  const int kq_mode = A3V_ENV_INT("A3V_GEMV_KQ", 1);
  const int kq_tiles = g.N / 16, kq_cus = cu_count();
  const bool kq_even = !pro && !(g.epi & A3V_EPI_SWIGLU) && kq_tiles % kq_cus == 0 && kq_tiles <= 2 * kq_cus;
  if (!w8 && arows == 8 && (kq_mode == 2 || (kq_mode == 1 && kq_even)) && g.N % 16 == 0 && (reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && g.lda % 8 == 0 &&
      (!pro || (reinterpret_cast<uintptr_t>(g.norm_w) & 15) == 0)) {
    const bool sw = (g.epi & A3V_EPI_SWIGLU) != 0;
    int slices = g.S;
    if (sw && slices > 6) slices = 6;
#ifdef A3V_ABLATION
    { const int e = A3V_ENV_INT("A3V_GEMV_KQ_SLICES", 0); if (e > 0) slices = sw && e > 6 ? 6 : (e > 12 ? 12 : e); }     // sweeps (tools/ab_gemv_kq.py)
#endif
    while (slices > 1 && g.nkb < 2 * slices) --slices;
    const int nw = slices * (sw ? 2 : 1);
    const int tiles = g.N / 16;
    if (!sw || tiles % 2 == 0) {
      const int kq_blocks = sw ? tiles / 2 : tiles;
      const size_t kq_lds = (size_t)nw * (2 * 4096 + 1024);
      void (*kq)(GemvArgs) = pro ? gemv_kq_bf16_kernel<true> : gemv_kq_bf16_kernel<false>;
      static bool kq_attr[A3V_MAX_DEV][2] = {};
      if (a3v_dyn_lds_once(kq_attr, pro ? 1 : 0, (const void*)kq, 150 * 1024) != 0) return false;
      hipLaunchKernelGGL(kq, dim3(kq_blocks), dim3(nw * 64), kq_lds, st, g);
      return true;
    }
  }
  void (*kern)(GemvArgs);
  // weights are streamed once per step by ONE CU each: non-temporal policy on their LDS-DMA (A3V_GEMV_NT=0: default policy, A/B)
  const bool nt = A3V_ENV_INT("A3V_GEMV_NT", 1) != 0;
#define GEMV_PICK(AUX)                                                                                                      \
  (w8 ? (arows == 8 ? (pro ? gemv_dma_bf16_kernel<8, true, true, AUX> : gemv_dma_bf16_kernel<8, false, true, AUX>)          \
                    : (pro ? gemv_dma_bf16_kernel<16, true, true, AUX> : gemv_dma_bf16_kernel<16, false, true, AUX>))       \
      : (arows == 8 ? (pro ? gemv_dma_bf16_kernel<8, true, false, AUX> : gemv_dma_bf16_kernel<8, false, false, AUX>)        \
                    : (pro ? gemv_dma_bf16_kernel<16, true, false, AUX> : gemv_dma_bf16_kernel<16, false, false, AUX>)))
  kern = nt ? GEMV_PICK(2) : GEMV_PICK(0);
#undef GEMV_PICK
  static bool attr_done[A3V_MAX_DEV][16] = {};
  const int ki = (nt ? 8 : 0) + (w8 ? 4 : 0) + (arows == 16 ? 2 : 0) + (pro ? 1 : 0);
  if (a3v_dyn_lds_once(attr_done, ki, (const void*)kern, 150 * 1024) != 0) return false;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), ldsb, st, g);
  return true;
}

extern "C" int a3v_gemm_skinny_split(int M, int N, int K) {
  (void)M;
  return (K % 128 == 0 && K >= 128) ? gemv_split(N, K, false) : 1;
}

extern "C" int64_t a3v_gemm_skinny_ws_bytes(int M, int N, int K) {
  (void)M;
  if (K % 128 || K < 128) return A3V_WS_PARTIALS;
  const int64_t tgs = (N + 63) / 64;
  return A3V_WS_PARTIALS + tgs * 8 * 4 * 1024;     // sized for the largest split of either weight format
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
  if (K % 128 == 0 && N <= 65536) {
    GemvArgs g{};
    g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = C; g.res = residual;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
    g.M = M; g.N = N; g.K = K; g.epi = epilogue;
    if (gemv_launch(g, partial, st)) return A3V_OK;
  }
  // direct-to-VGPR form (any K % 32 == 0): 8 waves split K in 32-element granules, LDS reduction (workspace unused)
  Skinny1Args q;
  q.A = (const bf16_t*)A; q.W = (const bf16_t*)W; q.C = C; q.res = residual;
  q.lda = lda; q.ldw = ldw; q.ldc = ldc; q.ldr = ldr;
  q.M = M; q.N = N; q.K = K; q.epi = epilogue;
  q.kslice = ((K / 32 + 7) / 8) * 32;
  if (epilogue & A3V_EPI_SWIGLU) hipLaunchKernelGGL(gemm_skinny1_bf16_kernel<2>, dim3((N + 31) / 32), dim3(512), 0, st, q);
  else hipLaunchKernelGGL(gemm_skinny1_bf16_kernel<1>, dim3((N + 15) / 16), dim3(512), 0, st, q);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// Decode-step forms of the GEMV (internal to the library, used by a3v_llama_decode_step):
//   norm_w != NULL : A is the un-normalised residual rows h; RMSNorm(h) is applied while the A slice is staged (ssq_in)
//   rope != 0      : [q|k|v] rows get RoPE and go to C (q) / the KV cache at `pos` (no separate rope kernel)
//   ssq_out != NULL: with a residual epilogue, also emit the per-tile sums of squares of the new rows
int a3v_gemv_fused(const void* A, int64_t lda, const void* W, int64_t ldw, const float* wscale, void* C, int64_t ldc, int M, int N,
                   int K, const void* residual, int64_t ldr, int epilogue, const void* norm_w, const float* ssq_in, float eps,
                   float* ssq_out, int rope, const float* cos_sin, void* k_cache, void* vt_cache, int H, int Hkv, int hd,
                   int Smax, int pos, void* ws, void* stream) {
  if (M <= 0 || M > 16 || K % 128 || N % 16 || N > 65536 || !ws) return A3V_ERR_SHAPE;
  GemvArgs g{};
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = C; g.res = residual;
  g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
  g.M = M; g.N = N; g.K = K;
  g.epi = epilogue | (rope ? GEMV_EPI_ROPEKV : 0) | (ssq_out ? GEMV_EPI_SSQ : 0);
  g.wscale = wscale;
  g.norm_w = (const bf16_t*)norm_w; g.ssq_in = ssq_in; g.eps = eps; g.ssq_tiles = K / 16; g.ssq_out = ssq_out;
  g.cos_sin = cos_sin; g.k_cache = (bf16_t*)k_cache; g.vt_cache = (bf16_t*)vt_cache;
  g.H = H; g.Hkv = Hkv; g.hd = hd; g.Smax = Smax; g.pos = pos;
  if (rope && (hd % 16 || !cos_sin || !k_cache || !vt_cache)) return A3V_ERR_ARG;
  if (!gemv_launch(g, ws, (hipStream_t)stream)) return A3V_ERR_SHAPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// Weight-only fp8 (OCP e4m3fn) form of a3v_gemm_skinny: Wq [N, K] bytes (row stride ldw BYTES), wscale [N] fp32;
// C = epilogue((A . dequant(Wq)^T) * wscale).  K % 256 == 0.  BASELINE config 5 / SURVEY 8(a) row Q: no reference oracle
// exists for this path (the reference's quantised path is bitsandbytes NF4, util/quant.py); parity is stated against
// the bf16 GEMV on the dequantised weights.
extern "C" int a3v_gemm_skinny_fp8(const void* A, int64_t lda, const void* Wq, int64_t ldw, const float* wscale, void* C, int64_t ldc,
                                   int M, int N, int K, const void* residual, int64_t ldr, int epilogue, void* workspace, void* stream) {
  if (M <= 0 || M > 16 || N <= 0 || K <= 0 || !A || !Wq || !wscale || !C || !workspace) return A3V_ERR_ARG;
  if (K % 256 || lda % 8 || ldw % 16 || N % 4 || ldc % 4) return A3V_ERR_SHAPE;
  if ((epilogue & A3V_EPI_SWIGLU) && (N % 32)) return A3V_ERR_SHAPE;
  if (epilogue & ~(A3V_EPI_RESIDUAL | A3V_EPI_SWIGLU | A3V_EPI_OUT_F32)) return A3V_ERR_ARG;
  if ((epilogue & A3V_EPI_RESIDUAL) && (!residual || (ldr % 4))) return A3V_ERR_ARG;
  GemvArgs g{};
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)Wq; g.C = C; g.res = residual; g.wscale = wscale;
  g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldr = ldr;
  g.M = M; g.N = N; g.K = K; g.epi = epilogue;
  if (!gemv_launch(g, workspace, (hipStream_t)stream)) return A3V_ERR_SHAPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// ------------------------------------------------------------------------------------
// Split-K skinny GEMM for the LoRA adapter products (N or M = 64 with K in the thousands): S slices of the K loop run
// as gridDim.y planes of the 128x128 kernel writing fp32 partial outputs; a3v_splitk_reduce sums the planes in slice
// order (deterministic), optionally accumulates into the destination, and rounds once -- the same fp32-accumulate,
// round-once arithmetic as the un-split kernel.
// ------------------------------------------------------------------------------------
namespace {
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int64_t plane, int M, int N, TO* __restrict__ out,
                                                            int64_t ldo, int accumulate) {
  const int64_t n4 = (int64_t)M * (N / 4);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / (N / 4)), c = (int)(i % (N / 4)) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(part + (int64_t)s * plane + (int64_t)r * N + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += x[e];
    }
    TO* o = out + (int64_t)r * ldo + c;
    // the product is rounded to bf16 exactly once, like the un-split kernel's epilogue (the value autocast's bf16 matmul
    // returns), also when it is then accumulated into an fp32 gradient
#pragma unroll
    for (int e = 0; e < 4; ++e) Cvt<TO>::st(o + e, accumulate ? Cvt<TO>::ld(o + e) + rbf(a[e]) : rbf(a[e]));
  }
}
}  // namespace

extern "C" int a3v_gemm_nt_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, float* partial, int M, int N, int K, int S,
                                  void* stream) {
  if (!A || !W || !partial || M <= 0 || N <= 0 || K <= 0 || S < 1 || S > 64) return A3V_ERR_ARG;
  if (K % 64 || lda % 8 || ldw % 8 || N % 4 || S > K / 64) return A3V_ERR_SHAPE;
  GemmArgs p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = partial; p.bias = nullptr; p.res = nullptr;
  p.lda = lda; p.ldw = ldw; p.ldc = N; p.ldr = 0;
  p.M = M; p.N = N; p.K = K; p.epi = A3V_EPI_OUT_F32 | GEMM_EPI_RAW; p.dbg = 0;
  p.c_split = (int64_t)M * N * 4;
  // Rows of the streamed operand per block and LDS stages (tools/skinny_stages_bench.py, operands rotating through 700 MB, us incl.
  // the reduce pass at 8728 x 64 x K = 4096 / 11008 / 12288 / 22016):
  //   64 rows, 2 stages, S = 4 (rounds 2-3)   24.0  54.5  60.0  112.1     half of the LDS-DMA traffic is the 64-row second operand
  //   64 rows, 4 stages, S = 3                23.1  52.8  56.4   96.9     k-tiles in flight across the barrier: +2..14 %
  //  256 rows, 3 stages, S = 7                23.5  46.3  49.0   82.9     35 row tiles x 7 slices = 245 blocks: ONE resident block per CU
  //  256 rows, 3 stages, S = 8                31.2  63.0  66.1  114.0     280 blocks: the 24 that wait for a CU double the time
  // so the 256-row form is taken when its blocks fill between half and all of the CUs (the caller picks S for that: train._skinny),
  // the 64-row form otherwise.  A3V_SKINNY_NARROW = 1 / 2 / 3 forces 256 / 128 / 64 rows, A3V_SKINNY_STAGES = 2 the two-stage kernels.
  const int narrow_env = A3V_ENV_INT("A3V_SKINNY_NARROW", 0);
  const int nst_env = A3V_ENV_INT("A3V_SKINNY_STAGES", 0);
  const int blocks256 = ((M + 255) / 256) * S;
  const bool wide = narrow_env ? narrow_env == 1 : (blocks256 <= cu_count() && 2 * blocks256 > cu_count());
  const int narrow = narrow_env ? narrow_env : (wide ? 1 : 3);
  if (N <= 64 && M >= 512 && narrow) {
    p.tiles_n = 1;
    const int nst = nst_env ? nst_env : (narrow == 1 ? 3 : 4);
    const int rows = narrow == 3 ? 64 : narrow == 2 ? 128 : 256;
    p.tiles_m = (M + rows - 1) / rows;
    static bool attr[A3V_MAX_DEV][9] = {};
    int attr_rc = 0;
    auto go = [&](auto kern, int bytes, int ai) {
      attr_rc = a3v_dyn_lds_once(attr, ai, (const void*)kern, bytes);
      if (attr_rc == 0) hipLaunchKernelGGL(kern, dim3(p.tiles_m, S), dim3(256), (size_t)bytes, (hipStream_t)stream, p);
    };
    // the product library carries the two forms the rule above picks; the other rows / stages of the sweep only in A3V_ABLATION builds
    const int bytes = (rows + 64) * 128 * nst;
    if (rows == 64 && nst == 4) go(gemm_nt_skinny_kernel<64, 4>, bytes, 1);
    else if (rows == 256 && nst == 3) go(gemm_nt_skinny_kernel<256, 3>, bytes, 6);
#ifdef A3V_ABLATION
    else if (rows == 64 && nst == 3) go(gemm_nt_skinny_kernel<64, 3>, bytes, 0);
    else if (rows == 64 && nst == 5) go(gemm_nt_skinny_kernel<64, 5>, bytes, 2);
    else if (rows == 128 && nst == 3) go(gemm_nt_skinny_kernel<128, 3>, bytes, 3);
    else if (rows == 128 && nst == 4) go(gemm_nt_skinny_kernel<128, 4>, bytes, 4);
    else if (rows == 128 && nst == 5) go(gemm_nt_skinny_kernel<128, 5>, bytes, 5);
    else if (rows == 256 && nst == 4) go(gemm_nt_skinny_kernel<256, 4>, bytes, 7);
#endif
    else if (nst != 2) return A3V_ERR_ARG;               // a rows / stages pair this build does not carry (sweeps only)
    else if (narrow == 3) {
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<64, 64, 4, 1>), dim3(p.tiles_m, S), dim3(256), 0, (hipStream_t)stream, p);
    } else if (narrow == 2) {
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 64, 4, 1>), dim3(p.tiles_m, S), dim3(256), 0, (hipStream_t)stream, p);
    } else {
      hipLaunchKernelGGL((gemm_nt_bf16_kernel<256, 64, 4, 1>), dim3(p.tiles_m, S), dim3(256), 0, (hipStream_t)stream, p);
    }
    if (attr_rc != 0) return attr_rc;
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  p.tiles_m = (M + 127) / 128; p.tiles_n = (N + 127) / 128;
  p.slow_epi = slow_epi_env(); p.nt_store = nt_store_env();
  hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128, 2, 2>), dim3(p.tiles_m * p.tiles_n, S), dim3(256), 0, (hipStream_t)stream, p);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_splitk_reduce(const float* partial, int S, int M, int N, void* out, int64_t ldo, int out_dtype, int accumulate, void* stream) {
  if (!partial || !out || S < 1 || M <= 0 || N <= 0) return A3V_ERR_ARG;
  if (N % 4) return A3V_ERR_SHAPE;
  const int64_t n4 = (int64_t)M * (N / 4);
  const int blocks = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
  if (out_dtype == A3V_BF16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, S, (int64_t)M * N, M, N, (bf16_t*)out, ldo, accumulate);
  else if (out_dtype == A3V_F32) hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, S, (int64_t)M * N, M, N, (float*)out, ldo, accumulate);
  else return A3V_ERR_DTYPE;
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// C[M,N] = epilogue(At^T . Wt): At [K, M] (row stride lda), Wt [K, N] (row stride ldw), both bf16 with the CONTRACTED index
// as the row index -- dW = dY^T . X straight from the token-major activations (no a3v_transpose of either operand).
// 256x256 tiles only (M, N >= 256 recommended); plain / residual / fp32 epilogues as a3v_gemm_nt; any K >= 1.
extern "C" int64_t a3v_gemm_tn_sumsq_slots(int M, int N) {
  return (int64_t)((M + 255) / 256) * ((N + 255) / 256) * 8 + 2048;      // 8 waves per 256 x 256 tile + the split-K reduce pass' blocks
}

static int gemm_tn_impl(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                        const void* residual, int64_t ldr, int epilogue, void* stream, float* sumsq, int64_t sumsq_cap) {
  if (!At || !Wt || !C || M <= 0 || N <= 0 || K <= 0) return A3V_ERR_ARG;
  if (sumsq && (!(epilogue & (A3V_EPI_OUT_F32 | A3V_EPI_RES_F32)) || sumsq_cap < a3v_gemm_tn_sumsq_slots(M, N))) return A3V_ERR_ARG;
  if (lda % 8 || ldw % 8 || N % 4 || ldc % 4 || M % 8) return A3V_ERR_SHAPE;
  if (epilogue & ~(A3V_EPI_RESIDUAL | A3V_EPI_RES_F32 | A3V_EPI_OUT_F32)) return A3V_ERR_ARG;
  if ((epilogue & (A3V_EPI_RESIDUAL | A3V_EPI_RES_F32)) && !residual) return A3V_ERR_ARG;
  if (((int64_t)(K - 1) * lda + M) * 2 >= (1LL << 31) || ((int64_t)(K - 1) * ldw + N) * 2 >= (1LL << 31)) return A3V_ERR_SHAPE;
  GemmArgs p{};
  p.A = (const bf16_t*)At; p.W = (const bf16_t*)Wt; p.C = C; p.bias = nullptr; p.res = residual;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.epi = epilogue; p.dbg = 0;
  p.tiles_n = (N + 255) / 256;
  p.sumsq = sumsq;
  hipStream_t st = (hipStream_t)stream;
  const GemmWs gws = gemm_ws_for(st);
  // rows of C beyond whole tile rounds (e.g. dW of w1|w3: 86 x 16 tiles = 5.4 rounds): split over the contracted index into fp32
  // planes + the reduce epilogue, as in a3v_gemm_nn / a3v_gemm_nt (needs the registered workspace; otherwise one plain launch)
  const int ncu = cu_count();
  const int tm_all = (M + 255) / 256;
  const long total = (long)tm_all * p.tiles_n;
  const long mt_h = (total / ncu) * ncu / p.tiles_n;
  const long rem_tiles = total - mt_h * p.tiles_n;
  int S = 1;
  // as many K-slices as fill the CUs once (round 2: was the largest power of two with 2 S rem_tiles <= CUs, i.e. 96 blocks for the 48
  // tail tiles of a [8728, 4096] output; 5 slices = 240 blocks finish the tail in 1/5 of a tile time instead of 1/2)
  S = rem_tiles > 0 ? (int)std::min<long>(8, ncu / rem_tiles) : 1;
  while (S > 1 && ((K + 63) / 64) < 16 * S) --S;
  if (S < 1) S = 1;
  const int m_big = (int)(mt_h * 256);
  const bool tail_on = A3V_ENV_INT("A3V_TN_TAIL", 1) != 0;
  if (tail_on && mt_h >= 1 && m_big < M && S > 1 && rem_tiles * 4 < 3 * ncu && gws.p && (int64_t)S * (M - m_big) * N * 4 <= gws.bytes) {
    GemmArgs q = p;
    q.M = m_big; q.tiles_m = (int)mt_h;
    launch_tn<false>(dim3(q.tiles_m * q.tiles_n), st, q);
    GemmArgs t = p;
    t.M = M - m_big;
    t.A = p.A + m_big;                      // At is [K][lda] with the C-row index contiguous: the tail rows of C are columns m_big.. of At
    t.C = gws.p; t.ldc = N; t.res = nullptr; t.sumsq = nullptr;      // (raw planes: the reduce pass below adds up the final values)
    t.epi = A3V_EPI_OUT_F32 | GEMM_EPI_RAW;
    t.tiles_m = (t.M + 255) / 256;
    t.c_split = (int64_t)t.M * N * 4;
    launch_tn<false>(dim3(t.tiles_m * t.tiles_n, S), st, t);
    const int esz = (epilogue & (A3V_EPI_OUT_F32 | A3V_EPI_RES_F32)) ? 4 : 2;
    void* Ct = (char*)C + (int64_t)m_big * ldc * esz;
    const void* Rt = residual ? (const char*)residual + (int64_t)m_big * ldr * ((epilogue & A3V_EPI_RES_F32) ? 4 : 2) : nullptr;
    const int64_t n4 = (int64_t)t.M * (N / 4);
    const int rb = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(rb), dim3(256), 0, st, gws.p, S, (int64_t)t.M * N, t.M, N, Ct, ldc, Rt, ldr, epilogue,
                       (const bf16_t*)nullptr, sumsq ? sumsq + (int64_t)tm_all * p.tiles_n * 8 : nullptr);
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  p.tiles_m = tm_all;
  launch_tn<false>(dim3(p.tiles_m * p.tiles_n), st, p);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_gemm_tn(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                           const void* residual, int64_t ldr, int epilogue, void* stream) {
  return gemm_tn_impl(At, lda, Wt, ldw, C, ldc, M, N, K, residual, ldr, epilogue, stream, nullptr, 0);
}

// a3v_gemm_tn with fp32 output that also leaves the sum of squares of what it stored, as a3v_gemm_tn_sumsq_slots(M, N) partial
// sums in `sumsq` (every slot it owns is written or was zeroed by the caller; slots of tiles outside C stay untouched = 0).
extern "C" int a3v_gemm_tn_sumsq(const void* At, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                                 const void* residual, int64_t ldr, int epilogue, float* sumsq, int64_t sumsq_cap, void* stream) {
  if (!sumsq) return A3V_ERR_ARG;
  return gemm_tn_impl(At, lda, Wt, ldw, C, ldc, M, N, K, residual, ldr, epilogue, stream, sumsq, sumsq_cap);
}

// split-K form of a3v_gemm_tn for adapter-sized outputs (M or N of a few dozen, long K: the LoRA weight gradients
// dB = dY^T . t and dA = dt^T . X): S slices of the contracted index write raw fp32 planes partial[s][M][N]; a3v_splitk_reduce
// sums them in slice order.  Waves of the 256 x 256 tile that fall outside C idle.
extern "C" int a3v_gemm_tn_splitk(const void* At, int64_t lda, const void* Wt, int64_t ldw, float* partial, int M, int N, int K, int S,
                                  void* stream) {
  if (!At || !Wt || !partial || M <= 0 || N <= 0 || K <= 0 || S < 1 || S > 64) return A3V_ERR_ARG;
  if (lda % 8 || ldw % 8 || N % 4 || M % 8 || S > (K + 63) / 64) return A3V_ERR_SHAPE;
  if (((int64_t)(K - 1) * lda + M) * 2 >= (1LL << 31) || ((int64_t)(K - 1) * ldw + N) * 2 >= (1LL << 31)) return A3V_ERR_SHAPE;
  GemmArgs p{};
  p.A = (const bf16_t*)At; p.W = (const bf16_t*)Wt; p.C = partial; p.bias = nullptr; p.res = nullptr;
  p.lda = lda; p.ldw = ldw; p.ldc = N; p.ldr = 0;
  p.M = M; p.N = N; p.K = K; p.epi = A3V_EPI_OUT_F32 | GEMM_EPI_RAW; p.dbg = 0;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  p.c_split = (int64_t)M * N * 4;
  launch_tn<false>(dim3(p.tiles_m * p.tiles_n, S), (hipStream_t)stream, p);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// (round 5) the fp8 product on the ring kernel: persistent tile walk, three LDS rings, staged epilogues.  `grid_y` > 1: split-K planes.
// Bias / activation kinds keep the two-stage kernel (no fp8 instantiation of that epilogue set).  A3V_GEMM_FP8_RING=0: the two-stage
// kernel for everything (A/B runs, equality tests).
static bool launch_ring_fp8(GemmArgs q, int grid_y, hipStream_t st, int tbm = 256) {
  if (A3V_ENV_INT("A3V_GEMM_FP8_RING", 1) == 0 || !pp_persistent()) return false;
  if (q.epi & (A3V_EPI_BIAS | A3V_EPI_GELU | A3V_EPI_QUICKGELU)) return false;
  q.epi &= ~GEMM_EPI_SCALE;                              // the kernel scales its accumulators itself, in front of the staged epilogues
  q.slow_epi = slow_epi_env(); q.nt_store = nt_store_env();
  q.skew = 0; q.xsync = nullptr;
  const int nt = q.tiles_m * q.tiles_n;
  const dim3 g(grid_y > 1 ? nt : std::min(nt, cu_count()), grid_y), b(512);
  q.xmap = (g.x & 63) || grid_y > 1 ? 0 : A3V_ENV_INT("A3V_GEMM_XMAP", 1);
  if (!((q.xmap & 4) && g.x == 256 && (q.tiles_m & 15) == 0 && (q.tiles_n & 15) == 0)) q.xmap &= ~4;
  if (tbm == 192) hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON, false, true, true, 192, true>), g, b, 0, st, q);
  else if (q.epi & GEMM_EPI_ROPEKV) hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_ROPE, false, true, true, 256, true>), g, b, 0, st, q);
  else hipLaunchKernelGGL((gemm_nt_bf16_ring_kernel<0, false, 0, true, EPI_SET_COMMON, false, true, true, 256, true>), g, b, 0, st, q);
  return true;
}

// C = epilogue((Aq . Wq^T) * sa[m] * sw[n]): fp8 (OCP e4m3fn) activations and weights with per-row fp32 scales, MX-scaled
// MFMA at twice the bf16 rate.  K % 128 == 0; 256 x 256 tiles for the whole problem; epilogues as a3v_gemm_nt (bf16 C).
static int gemm_nt_fp8_impl(const void* Aq, int64_t lda, const float* sa, const void* Wq, int64_t ldw, const float* sw, void* C,
                            int64_t ldc, int M, int N, int K, const void* bias, const void* residual, int64_t ldr, int epilogue,
                            void* stream, const RopeKvArgs* rk) {
  if (!Aq || !sa || !Wq || !sw || !C || M <= 0 || N <= 0 || K <= 0) return A3V_ERR_ARG;
  if (K % 128 || lda % 16 || ldw % 16 || N % 4 || ldc % 4) return A3V_ERR_SHAPE;
  if (epilogue & ~(A3V_EPI_BIAS | A3V_EPI_GELU | A3V_EPI_QUICKGELU | A3V_EPI_RESIDUAL | A3V_EPI_SWIGLU | A3V_EPI_OUT_F32 | A3V_EPI_RES_F32)) return A3V_ERR_ARG;
  if ((epilogue & A3V_EPI_BIAS) && !bias) return A3V_ERR_ARG;
  if ((epilogue & (A3V_EPI_RESIDUAL | A3V_EPI_RES_F32)) && (!residual || ldr % 4)) return A3V_ERR_ARG;
  if ((epilogue & A3V_EPI_SWIGLU) && (N % 32)) return A3V_ERR_SHAPE;
  if ((int64_t)(M - 1) * lda + K >= (1LL << 31) || (int64_t)(N - 1) * ldw + K >= (1LL << 31)) return A3V_ERR_SHAPE;
  GemmArgs p{};
  p.A = (const bf16_t*)Aq; p.W = (const bf16_t*)Wq; p.C = C; p.bias = bias; p.res = residual;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.epi = epilogue | GEMM_EPI_SCALE; p.dbg = 0;
  p.sa = sa; p.sw = sw;
  if (rk) { p.rk = *rk; p.epi |= GEMM_EPI_ROPEKV; }
  p.tiles_n = (N + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  const GemmWs gws = gemm_ws_for(st);
  // Tile rounds: tiles_m x tiles_n blocks over the CUs.  When the last round would be mostly empty (e.g. 35 x 16 = 560 tiles on 256
  // CUs: a third round at 19 %), the tile rows that fill whole rounds run as usual and the remaining rows (< one round of tiles)
  // are split over K into raw fp32 planes -- S times the blocks at 1/S the length -- which a reduce pass rounds and stores with
  // the residual forms of the epilogue (needs the registered workspace of a3v_gemm_set_workspace; otherwise one plain launch).
  const int ncu = cu_count();
  const int tm_all = (M + 255) / 256;
  const long total = (long)tm_all * p.tiles_n;
  const int simple = A3V_EPI_RESIDUAL | A3V_EPI_RES_F32 | A3V_EPI_OUT_F32;
  const long mt_h = (total / ncu) * ncu / p.tiles_n;                 // tile rows that make whole rounds
  const long rem_tiles = total - mt_h * p.tiles_n;
  int S = 1;
  while (rem_tiles * S * 2 <= ncu && S < 8 && (K / 128) >= 8 * S) S *= 2;   // measured on wo / w2 of 7B: S = 2 (96 blocks) beat S = 8 (384 blocks)
  const int m_big = (int)(mt_h * 256);
  // (round 5) 192 x 256 ring tiles where they cost fewer rounds than whole 256-row rounds + a split-K tail (a round of them is 0.79 of a
  // 256-row round; the tail's ~32 us do not shrink with the fp8 round time ~12 + 0.0103 K us): A3V_GEMM_FP8_192 = 0 never, 2 always
  {
    const int e192 = A3V_ENV_INT("A3V_GEMM_FP8_192", 1);
    const bool tail_form = !rk && !(epilogue & ~simple) && mt_h >= 1 && m_big < M && S > 1 && rem_tiles * 4 < 3 * ncu;
    const double round_us = 12.0 + 0.0103 * K;
    const double c_now = tail_form ? (double)(mt_h * p.tiles_n / ncu) + 1.0 / S + 32.0 / round_us : (double)((total + ncu - 1) / ncu);
    const long t192 = (long)((M + 191) / 192) * p.tiles_n;
    const double c_192 = 0.79 * (double)((t192 + ncu - 1) / ncu) + 0.02;
    if (e192 && !rk && !(epilogue & (A3V_EPI_SWIGLU | A3V_EPI_BIAS | A3V_EPI_GELU | A3V_EPI_QUICKGELU)) && M >= 512 && N >= 512 && (e192 == 2 || c_192 < c_now)) {
      GemmArgs q = p;
      q.tiles_m = (M + 191) / 192;
      if (launch_ring_fp8(q, 1, st, 192)) { A3V_LAUNCH_CHECK(); return A3V_OK; }
    }
  }
  if (!rk && !(epilogue & ~simple) && mt_h >= 1 && m_big < M && S > 1 && rem_tiles * 4 < 3 * ncu && gws.p &&
      (int64_t)S * (M - m_big) * N * 4 <= gws.bytes) {
    GemmArgs q = p;
    q.M = m_big; q.tiles_m = (int)mt_h;
    if (!launch_ring_fp8(q, 1, st)) hipLaunchKernelGGL(gemm_nt_fp8_pp_kernel, dim3(q.tiles_m * q.tiles_n), dim3(512), 0, st, q);
    GemmArgs t = p;
    t.M = M - m_big;
    t.A = (const bf16_t*)((const char*)Aq + (int64_t)m_big * lda);
    t.sa = sa + m_big;
    t.C = gws.p; t.ldc = N; t.res = nullptr; t.bias = nullptr;
    t.epi = A3V_EPI_OUT_F32 | GEMM_EPI_RAW | GEMM_EPI_SCALE;
    t.tiles_m = (t.M + 255) / 256;
    t.c_split = (int64_t)t.M * N * 4;
    if (!launch_ring_fp8(t, S, st)) hipLaunchKernelGGL(gemm_nt_fp8_pp_kernel, dim3(t.tiles_m * t.tiles_n, S), dim3(512), 0, st, t);
    const int esz = (epilogue & (A3V_EPI_OUT_F32 | A3V_EPI_RES_F32)) ? 4 : 2;
    void* Ct = (char*)C + (int64_t)m_big * ldc * esz;
    const void* Rt = residual ? (const char*)residual + (int64_t)m_big * ldr * ((epilogue & A3V_EPI_RES_F32) ? 4 : 2) : nullptr;
    const int64_t n4 = (int64_t)t.M * (N / 4);
    const int rb = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(rb), dim3(256), 0, st, gws.p, S, (int64_t)t.M * N, t.M, N, Ct, ldc, Rt, ldr, epilogue);
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  p.tiles_m = tm_all;
  if (!launch_ring_fp8(p, 1, st)) hipLaunchKernelGGL(gemm_nt_fp8_pp_kernel, dim3(p.tiles_m * p.tiles_n), dim3(512), 0, st, p);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_gemm_nt_fp8(const void* Aq, int64_t lda, const float* sa, const void* Wq, int64_t ldw, const float* sw, void* C,
                               int64_t ldc, int M, int N, int K, const void* bias, const void* residual, int64_t ldr, int epilogue,
                               void* stream) {
  return gemm_nt_fp8_impl(Aq, lda, sa, Wq, ldw, sw, C, ldc, M, N, K, bias, residual, ldr, epilogue, stream, nullptr);
}

extern "C" int a3v_gemm_qkv_rope_fp8(const void* Aq, int64_t lda, const float* sa, const void* Wq, int64_t ldw, const float* sw, int K,
                                     void* q_out, int64_t ldq, void* k_cache, void* vt_cache, const float* cos_sin, int B, int S,
                                     int H, int Hkv, int hd, int Smax, int start_pos, int rope_pos0, void* stream) {
  if (!q_out || !k_cache || !vt_cache || !cos_sin || B <= 0 || S <= 0 || H <= 0 || Hkv <= 0) return A3V_ERR_ARG;
  if ((hd != 64 && hd != 128) || ldq % 4 || start_pos < 0 || start_pos + S > Smax) return A3V_ERR_SHAPE;
  RopeKvArgs rk{};
  rk.q_out = (bf16_t*)q_out; rk.k_cache = (bf16_t*)k_cache; rk.vt_cache = (bf16_t*)vt_cache; rk.cos_sin = cos_sin;
  rk.ldq = ldq; rk.S = S; rk.H = H; rk.Hkv = Hkv; rk.hd_shift = hd == 128 ? 7 : 6; rk.Smax = Smax;
  rk.start_pos = start_pos; rk.rope_pos0 = rope_pos0; rk.m_off = 0;
  return gemm_nt_fp8_impl(Aq, lda, sa, Wq, ldw, sw, q_out, ldq, B * S, (H + 2 * Hkv) * hd, K, nullptr, nullptr, 0, 0, stream, &rk);
}

// "NN" GEMM: C[M,N] = epilogue(A . Wt) with A [M, K] row-major and Wt [K, N] row-major (the contracted index is Wt's ROW index) --
// the input gradient dX = dY . W on the weight image the forward pass uses, without a transposed copy of W.  K % 64 == 0.
// Tile rows that fill whole rounds of the CUs run as one launch; the remaining rows are split over K into fp32 planes + the
// reduce epilogue when a workspace is registered (a3v_gemm_set_workspace), exactly as in a3v_gemm_nt / a3v_gemm_nt_fp8.
extern "C" int a3v_gemm_nn(const void* A, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                           const void* residual, int64_t ldr, int epilogue, void* stream) {
  if (!A || !Wt || !C || M <= 0 || N <= 0 || K <= 0) return A3V_ERR_ARG;
  if (K % 64 || lda % 8 || ldw % 8 || N % 8 || ldc % 4) return A3V_ERR_SHAPE;
  const int simple = A3V_EPI_RESIDUAL | A3V_EPI_RES_F32 | A3V_EPI_OUT_F32;
  const bool swb = epilogue == A3V_EPI_SWIGLU_BWD;     // the product is d(act): `residual` = the forward's [gate | up] rows, C = [d gate | d up] (see a3v_gemm_nt)
  if (swb) {
    if (!residual || (ldr % 4) || ldr < 2 * (int64_t)N || ldc < 2 * (int64_t)N) return A3V_ERR_ARG;
  } else if (epilogue & ~simple) {
    return A3V_ERR_ARG;
  }
  if ((epilogue & (A3V_EPI_RESIDUAL | A3V_EPI_RES_F32)) && (!residual || ldr % 4)) return A3V_ERR_ARG;
  if (((int64_t)(M - 1) * lda + K) * 2 >= (1LL << 31) || ((int64_t)(K - 1) * ldw + N) * 2 >= (1LL << 31)) return A3V_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const GemmWs gws = gemm_ws_for(st);
  GemmArgs p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)Wt; p.C = C; p.bias = nullptr; p.res = residual;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.epi = epilogue; p.dbg = 0;
  p.tiles_n = (N + 255) / 256;
  const int ncu = cu_count();
  const int tm_all = (M + 255) / 256;
  const long total = (long)tm_all * p.tiles_n;
  const long mt_h = (total / ncu) * ncu / p.tiles_n;
  const long rem_tiles = total - mt_h * p.tiles_n;
  int S = 1;
  S = rem_tiles > 0 ? (int)std::min<long>(8, ncu / rem_tiles) : 1;      // as many K-slices as fill the CUs once (see a3v_gemm_tn)
  while (S > 1 && (K / 64) < 16 * S) --S;
  if (S < 1) S = 1;
  const int m_big = (int)(mt_h * 256);
  if (!swb && mt_h >= 1 && m_big < M && S > 1 && rem_tiles * 4 < 3 * ncu && gws.p && (int64_t)S * (M - m_big) * N * 4 <= gws.bytes) {
    GemmArgs q = p;
    q.M = m_big; q.tiles_m = (int)mt_h;
    launch_tn<true>(dim3(q.tiles_m * q.tiles_n), st, q);
    GemmArgs t = p;
    t.M = M - m_big;
    t.A = p.A + (int64_t)m_big * lda;
    t.C = gws.p; t.ldc = N; t.res = nullptr;
    t.epi = A3V_EPI_OUT_F32 | GEMM_EPI_RAW;
    t.tiles_m = (t.M + 255) / 256;
    t.c_split = (int64_t)t.M * N * 4;
    launch_tn<true>(dim3(t.tiles_m * t.tiles_n, S), st, t);
    const int esz = (epilogue & (A3V_EPI_OUT_F32 | A3V_EPI_RES_F32)) ? 4 : 2;
    void* Ct = (char*)C + (int64_t)m_big * ldc * esz;
    const void* Rt = residual ? (const char*)residual + (int64_t)m_big * ldr * ((epilogue & A3V_EPI_RES_F32) ? 4 : 2) : nullptr;
    const int64_t n4 = (int64_t)t.M * (N / 4);
    const int rb = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(rb), dim3(256), 0, st, gws.p, S, (int64_t)t.M * N, t.M, N, Ct, ldc, Rt, ldr, epilogue);
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  p.tiles_m = tm_all;
  launch_tn<true>(dim3(p.tiles_m * p.tiles_n), st, p);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

