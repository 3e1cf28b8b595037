// EXPERIMENT (round 4; built only with `make EXPERIMENTS=1`, not part of the C-ABI of include/a3vlm_hip.h; driver: tools/lora_stream_bench.py).
// Outcome (profiles/r04d_stream_experiments.txt): correct on the first run, NOT faster than the kernels it was meant to replace --
// the NT form 2.2 - 3.3 TB/s against 3.4 - 5.0 (a wave load in the MFMA operand layout = 16 rows x 64 B costs ~90 TA cycles instead of
// 16, whether it hits L2 or not), the TN form 4.5 - 5.0 against 4.1 - 4.8 at its best slice count; ring depth (3 or 4 stages) and panel
// width (256 or 512 B) change nothing.  A bare LDS-DMA stream with the same panel walk (tools/ubench/dmaread.hip) does 5.1 TB/s with the
// ~500 blocks these grids have and 6 - 6.5 only with >= 2000 small blocks, i.e. with slice counts whose partial planes cost more than
// they win.  Kept as the record of what bounds these products; the product path is a3v_gemm_nt_splitk / a3v_gemm_tn_strip.
//
// Rank-64 products of a LoRA step as HBM STREAMS (model/peft.py:58-159: lora_a / lora_b of every decoder linear, and what autograd
// forms for their weight gradients, engine_finetune.py:55-57), gfx950:
//
//   a3v_gemm_nt_skinny64   P[s][M][64]  = X[M, K-slice s] . W[64, K-slice s]^T      t = x A^T (forward), dt = dy B (backward)
//   a3v_skinny_reduce      out[M][64] (bf16, row stride ldo) and out^T[64][ldt] (bf16, tokens >= M zero up to the next multiple of 64)
//   a3v_gemm_tn_strip2     P[s][R][N]   = (T^T)[R, token-slice s] . X[token-slice s, N]     dB^T = t^T dy, dA = dt^T x
//
// 2 * 64 * (bytes of X) / 2 FLOPs are nothing; each launch reads X (71 .. 384 MB at the 7B shapes) exactly once and what decides its
// speed is how many bytes of X each CU has in flight.  The general kernels these replace stage BOTH operands through two LDS stages
// per block (gemm_nt_bf16_kernel<64, 64>: 8 KiB of X per 32 KiB of LDS; gemm_tn_strip_kernel: 16 KiB per 48): with the ~2 blocks
// per CU the grids give, 17 - 48 KiB of X in flight per CU = 3.5 - 4 TB/s.  Here
//
//  * NT (K contiguous in both operands, so a lane's 16-byte global load IS its MFMA operand): X goes HBM -> AGPRs with
//    `buffer_load_dwordx4 a[..]`, four k-steps per chunk, a ring of four chunks per wave; the 64-row W chunk (re-read from L2 by every
//    block) is staged through a three-stage LDS ring by LDS-DMA in fragment order (one 1-KiB piece = one (k-step, n-tile) fragment of
//    the wave), read back with ds_read_b128 and used as the A operand (M = the 64 outputs) against X from the AGPRs as B (N = 16
//    rows): the lane ends up with 4 consecutive outputs of one row = 16-byte stores of the fp32 partial plane.
//  * TN (contraction over the token rows): X needs the transposing LDS read (ds_read_b64_tr_b16), so X goes HBM -> LDS ring (four
//    16-KiB stages, three in flight) by LDS-DMA with the bank-spreading chunk permutation of a3v_strip.hip; the small operand is taken
//    TRANSPOSED ([R][tokens], written by a3v_skinny_reduce next to the row-major result), so a lane's 16-byte global load of it is
//    again the MFMA operand as it stands: it is loaded into AGPRs (L2 hits) and never touches the LDS -- every LDS byte is X.
//
// Both kernels own their AGPRs by hand (literal registers in asm, the range as clobbers of every asm statement; the compiler never
// sees them) and count their own `s_waitcnt vmcnt`: a compiler-managed wait in front of a fragment read would drain the whole ring
// (loads retire in order), which is exactly the in-flight depth these kernels exist for.  tests/test_kernel_budget_cpu.py audits that
// no compiler-generated AGPR traffic, scratch or vmcnt appears in them.
#include "a3v_common.h"
#include <type_traits>

namespace {
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define SK_C10(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
#define SK_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", SK_C10(1), SK_C10(2), SK_C10(3), SK_C10(4), SK_C10(5), "a60", "a61", "a62", "a63"

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, int64_t bytes) {
  const uint64_t a = (uint64_t)base;
  i32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)(bytes < 0 ? 0 : (bytes < 0x7fffffff ? bytes : 0x7fffffff)));
  r[3] = 0x00020000;
  return r;
}
// 1 KiB of global memory -> LDS at lds_addr + 16 * lane (M0 = the wave's LDS destination, written in the statement that reads it)
__device__ __forceinline__ void dma16(i32x4_t rs, unsigned lds_addr, unsigned voff, int soff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
               : "memory", SK_AGPRS);
}
// 16 bytes per lane -> a[R:R+3]
template <int R, int IMM>
__device__ __forceinline__ void load_agpr(i32x4_t rs, unsigned voff, int soff) {
  asm volatile("buffer_load_dwordx4 a[%c3:%c4], %0, %1, %2 offen offset:%c5" ::"v"(voff), "s"(rs), "s"(soff), "i"(R), "i"(R + 3), "i"(IMM)
               : "memory", SK_AGPRS);
}
// acc (VGPRs) += A (VGPRs, the M index) x a[R:R+3] (the N index)
template <int R>
__device__ __forceinline__ void mfma16_va(f32x4& acc, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "i"(R), "i"(R + 3) : SK_AGPRS);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(N) : "memory", SK_AGPRS);
}
__device__ __forceinline__ void barrier_asm() { asm volatile("s_barrier" ::: "memory", SK_AGPRS); }
// the accumulators were written by asm MFMAs the compiler knows nothing about: pad the XDL-write -> VALU / store-read distance by hand
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory", SK_AGPRS); }

// ------------------------------------------------------------------------------------------------------------------- NT, 64 outputs
struct SkArgs {
  const bf16_t* X;   // [M][ldx]
  const bf16_t* W;   // [64][ldw]
  float* part;       // [S][M][64]
  int64_t ldx, ldw;
  int M, K;          // K % 128 == 0
  int abl;           // ablation bits of tools/lora_stream_bench.py (0 in the product): 1 no W loads, 2 no X loads, 4 no LDS reads / MFMA
};
constexpr int SK_CH = 128;             // k per chunk = 4 MFMA k-steps
constexpr int SK_WST = 64 * SK_CH * 2; // one W stage: 16 KiB
constexpr int SK_NSW = 3;              // W stages (W runs two chunks ahead)
constexpr int SK_NX = 4;               // X chunk slots of a wave in AGPRs (X runs three chunks ahead): a[16 slot + 4 kstep ..]

__global__ __launch_bounds__(256, 2) void skinny_nt64_kernel(SkArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[SK_NSW * SK_WST];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 64;
  const int nch_all = p.K / SK_CH;
  const int c0 = (int)((int64_t)nch_all * blockIdx.y / gridDim.y), c1 = (int)((int64_t)nch_all * (blockIdx.y + 1) / gridDim.y);
  const int n = c1 - c0;
  const int rows = p.M - m0 < 64 ? p.M - m0 : 64;
  // rows past M read as zeros (their offset is past num_records because ldx >= K)
  const i32x4_t rsX = make_rsrc(p.X + (int64_t)m0 * p.ldx, ((int64_t)(rows - 1) * p.ldx + p.K) * 2);
  const i32x4_t rsW = make_rsrc(p.W, ((int64_t)63 * p.ldw + p.K) * 2);
  const unsigned voX = (unsigned)(((16 * wave + (lane & 15)) * p.ldx + 8 * (lane >> 4)) * 2);
  unsigned voW[4];   // this wave stages k-step `wave` of every chunk: fragment (wave, j) = rows 16 j + (lane & 15), k 32 wave + 8 (lane >> 4) ..
#pragma unroll
  for (int j = 0; j < 4; ++j) voW[j] = (unsigned)(((16 * j + (lane & 15)) * p.ldw + 32 * wave + 8 * (lane >> 4)) * 2);
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  auto issue_w = [&](int c) __attribute__((always_inline)) {
    if (p.abl & 1) return;
    const unsigned dst = lds0 + (c % SK_NSW) * SK_WST + wave * 4096;
    const int so = (c0 + c) * (SK_CH * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rsW, dst + j * 1024, voW[j], so);
  };
  auto issue_x = [&](auto slot_, int c) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_)::value;
    if (p.abl & 2) return;
    const int so = (c0 + c) * (SK_CH * 2);
    load_agpr<16 * SL + 0, 0>(rsX, voX, so);
    load_agpr<16 * SL + 4, 64>(rsX, voX, so);
    load_agpr<16 * SL + 8, 128>(rsX, voX, so);
    load_agpr<16 * SL + 12, 192>(rsX, voX, so);
  };
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // queue order = order of need: X(0) | W(0) X(1) | W(1) X(2) | then iteration c issues W(c+2) X(c+3)
  if (0 < n) issue_x(std::integral_constant<int, 0>{}, 0);
  if (0 < n) issue_w(0);
  if (1 < n) issue_x(std::integral_constant<int, 1>{}, 1);
  if (1 < n) issue_w(1);
  if (2 < n) issue_x(std::integral_constant<int, 2>{}, 2);
  for (int cb = 0; cb < n; cb += SK_NX) {
    sfor<0, SK_NX>([&](auto u_) __attribute__((always_inline)) {
      constexpr int U = decltype(u_)::value;
      const int c = cb + U;
      if (c < n) {
        // W(c) and everything older (X(c) among it) has landed once at most X(c+1), W(c+1), X(c+2) are outstanding; near the end of
        // the slice fewer loads follow W(c) and the count would not cover it: drain
        if (c + 2 < n) wait_vm<12>(); else wait_vm<0>();
        barrier_asm();          // everybody's pieces of W(c) are in the LDS; everybody is done reading stage (c - 1) % 3 = (c + 2) % 3
        if (c + 2 < n) issue_w(c + 2);
        if (c + 3 < n) issue_x(std::integral_constant<int, (U + 3) % SK_NX>{}, c + 3);
        const char* st = lds + (c % SK_NSW) * SK_WST + lane * 16;
        if (!(p.abl & 4)) sfor<0, 4>([&](auto ks_) __attribute__((always_inline)) {
          constexpr int KS = decltype(ks_)::value;
          bf16x8 wf[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(st + (KS * 4 + j) * 1024);
#pragma unroll
          for (int j = 0; j < 4; ++j) mfma16_va<16 * U + 4 * KS>(acc[j], wf[j]);
        });
      }
    });
  }
  mfma_drain();
  // D = W_frag x X_frag: the lane holds P[row m0 + 16 wave + (lane & 15)][16 j + 4 (lane >> 4) + 0..3]
  const int row = m0 + 16 * wave + (lane & 15);
  if (row < p.M) {
    float* dst = p.part + ((int64_t)blockIdx.y * p.M + row) * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + 16 * j) = acc[j];
  }
}

// out[m][0..63] = sum_s P[s][m][..] rounded to bf16 once; out^T[n][m] the same values, and zeros for M <= m < Mpad
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* __restrict__ part, int S, int M, int Mpad, bf16_t* __restrict__ out,
                                                            int64_t ldo, bf16_t* __restrict__ outT, int64_t ldt) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int m = t >> 4, nq = t & 15;
  if (m >= Mpad) return;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (m < M) {
    const float* src = part + (int64_t)m * 64 + 4 * nq;
    for (int i = 0; i < S; ++i) s += *reinterpret_cast<const f32x4*>(src + (int64_t)i * M * 64);
  }
  bf16x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = (bf16_t)s[q];
  if (m < M && out) *reinterpret_cast<bf16x4*>(out + (int64_t)m * ldo + 4 * nq) = v;
  if (outT) {
#pragma unroll
    for (int q = 0; q < 4; ++q) outT[(int64_t)(4 * nq + q) * ldt + m] = v[q];
  }
}

// ------------------------------------------------------------------------------------------------------------------- TN strip, R <= 64
struct Strip2Args {
  const bf16_t* Tt;  // [64][ldt]: the small operand transposed; tokens Kt .. next multiple of 64 are zeros
  const bf16_t* X;   // [Kt][ldx], N columns
  float* part;       // [S][R][N]
  int64_t ldt, ldx;
  int R, N, Kt;
  int abl;
};
constexpr int S2_TILE = 16384;          // one X stage: [BK tokens][BN columns] bf16 with BK * BN = 8192
__device__ __forceinline__ int key_w(int k) { return (((k >> 3) & 1) << 2) | (k & 3); }   // as a3v_strip.hip: XOR key of a row's 32-byte chunks (low 3 bits)

// NST LDS stages of X (NST - 1 in flight); BN output columns per block: 128 (64-token tiles, a DMA instruction = 4 rows x 256 B) or
// 256 (32-token tiles, 2 rows x 512 B: longer runs per HBM page).  The T^T fragments of a tile sit in a[(BK / 8) (tile % NST) ..]
template <int NST, int BN>
__global__ __launch_bounds__(256, 2) void gemm_tn_strip2_kernel(Strip2Args p) {
  constexpr int BK = 8192 / BN, KSN = BK / 32, NJ = BN / 16, ROWB = BN * 2, RPP = 1024 / ROWB, LPR = ROWB / 16, TREG = 4 * KSN;
  constexpr int NLD = 4 + KSN;           // VMEM instructions of a wave per tile
  __shared__ __attribute__((aligned(1024))) char lds[NST * S2_TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * BN;
  const int nk_all = (p.Kt + BK - 1) / BK;
  const int kt0 = (int)((int64_t)nk_all * blockIdx.y / gridDim.y), kt1 = (int)((int64_t)nk_all * (blockIdx.y + 1) / gridDim.y);
  const int n = kt1 - kt0;
  const i32x4_t rsX = make_rsrc(p.X + n0, ((int64_t)(p.Kt - 1) * p.ldx + (p.N - n0)) * 2);   // tokens past Kt read as zeros
  const i32x4_t rsT = make_rsrc(p.Tt, ((int64_t)63 * p.ldt + (int64_t)(p.Kt + 63) / 64 * 64) * 2);
  unsigned voX[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int row = (wave * 4 + c) * RPP + lane / LPR, pos = lane % LPR;
    voX[c] = (unsigned)((row * p.ldx) * 2 + ((((pos >> 1) ^ key_w(row)) << 5) + (pos & 1) * 16));
  }
  // this wave's outputs: r = 16 wave + (lane & 15) (the MFMA's N index), all BN columns of the block
  const unsigned voT = (unsigned)(((16 * wave + (lane & 15)) * p.ldt + 8 * (lane >> 4)) * 2);
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  auto issue = [&](auto slot_, int t) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_)::value;
    const unsigned dst = lds0 + SL * S2_TILE + wave * 4096;
    const int64_t tok = (int64_t)(kt0 + t) * BK;
    const int soX = (int)(tok * p.ldx * 2), soT = (int)(tok * 2);
    if (!(p.abl & 2)) {
#pragma unroll
      for (int c = 0; c < 4; ++c) dma16(rsX, dst + c * 1024, voX[c], soX);
    }
    if (!(p.abl & 1)) {
      load_agpr<TREG * SL + 0, 0>(rsT, voT, soT);
      if constexpr (KSN == 2) load_agpr<TREG * SL + 4, 64>(rsT, voT, soT);
    }
  };
  // transposing fragment reads, as gemm_tn_strip_kernel: sub-block (ks, jj) = token rows 32 ks + 8 fg + 4 jj + (il >> 2)
  const int fg = lane >> 4, il = lane & 15;
  int roW[KSN][2], kW[KSN][2];
#pragma unroll
  for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int k = ks * 32 + 8 * fg + 4 * jj + (il >> 2);
      roW[ks][jj] = k * ROWB + (il & 3) * 8;
      kW[ks][jj] = key_w(k);
    }
  auto tr8 = [&](const char* tile, int ro0, int ro1, int x0, int x1) __attribute__((always_inline)) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + ro0 + (x0 << 5)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + ro1 + (x1 << 5)));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };
  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  sfor<0, NST - 1>([&](auto u_) __attribute__((always_inline)) {
    if (decltype(u_)::value < n) issue(u_, decltype(u_)::value);
  });
  for (int tb = 0; tb < n; tb += NST) {
    sfor<0, NST>([&](auto u_) __attribute__((always_inline)) {
      constexpr int U = decltype(u_)::value;
      const int t = tb + U;
      if (t < n) {
        // tile t has landed once at most the NST - 2 tiles behind it (NLD loads each) are outstanding
        if (t + NST - 2 < n) wait_vm<NLD * (NST - 2)>(); else wait_vm<0>();
        barrier_asm();
        if (t + NST - 1 < n) issue(std::integral_constant<int, (U + NST - 1) % NST>{}, t + NST - 1);
        const char* Xt = lds + U * S2_TILE;
        if (!(p.abl & 4)) sfor<0, KSN>([&](auto ks_) __attribute__((always_inline)) {
          constexpr int KS = decltype(ks_)::value;
          sfor<0, NJ / 8>([&](auto h_) __attribute__((always_inline)) {
            constexpr int H = decltype(h_)::value;
            bf16x8 xf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = tr8(Xt, roW[KS][0], roW[KS][1], (8 * H + j) ^ kW[KS][0], (8 * H + j) ^ kW[KS][1]);
#pragma unroll
            for (int j = 0; j < 8; ++j) mfma16_va<TREG * U + 4 * KS>(acc[8 * H + j], xf[j]);
          });
        });
      }
    });
  }
  mfma_drain();
  // D = X_frag x T_frag: the lane holds C[r = 16 wave + (lane & 15)][n = n0 + 16 j + 4 (lane >> 4) + 0..3]
  const int r = 16 * wave + (lane & 15);
  if (r < p.R) {
    float* dst = p.part + ((int64_t)blockIdx.y * p.R + r) * p.N;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nn = n0 + 16 * j + 4 * (lane >> 4);
      if (nn + 4 <= p.N) *reinterpret_cast<f32x4*>(dst + nn) = acc[j];
    }
  }
}
}  // namespace

extern "C" int a3v_gemm_nt_skinny64(const void* X, int64_t ldx, const void* W, int64_t ldw, float* partial, int M, int K, int S, void* stream);
extern "C" int a3v_skinny_reduce(const float* partial, int S, int M, void* out, int64_t ldo, void* outT, int64_t ldt, void* stream);
extern "C" int a3v_gemm_tn_strip2(const void* Tt, int64_t ldt, const void* X, int64_t ldx, float* partial, int R, int N, int Kt, int S, void* stream);

// P[s] = X[:, K-slice s] . W[:, K-slice s]^T for s < S (a3v_skinny_reduce sums the planes).  Replaces F.linear(x, lora_a.weight) of
// model/peft.py:95-101 for the fused adapter group of a decoder linear, and dy @ lora_b.weight of its autograd.
extern "C" int a3v_gemm_nt_skinny64(const void* X, int64_t ldx, const void* W, int64_t ldw, float* partial, int M, int K, int S, void* stream) {
  if (!X || !W || !partial || M <= 0 || K <= 0 || S < 1 || S > 64) return A3V_ERR_ARG;
  if (K % SK_CH || ldx % 8 || ldw % 8 || ldx < K || ldw < K || S > K / SK_CH) return A3V_ERR_SHAPE;
  if ((64 * ldx + K) * 2 >= (1LL << 31) || (64 * ldw + K) * 2 >= (1LL << 31)) return A3V_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(partial)) & 15) return A3V_ERR_SHAPE;
  SkArgs p{(const bf16_t*)X, (const bf16_t*)W, partial, ldx, ldw, M, K, A3V_ENV_INT("A3V_SK_ABL", 0)};
  hipLaunchKernelGGL(skinny_nt64_kernel, dim3((M + 63) / 64, S), dim3(256), 0, (hipStream_t)stream, p);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// out[M][64] (row stride ldo; may be null) and out^T[64][ldt] (may be null; columns M .. roundup(M, 64) - 1 are written as zeros:
// a3v_gemm_tn_strip2 reads whole 64-token tiles of it) from S fp32 planes, each value rounded to bf16 once
extern "C" int a3v_skinny_reduce(const float* partial, int S, int M, void* out, int64_t ldo, void* outT, int64_t ldt, void* stream) {
  if (!partial || S < 1 || M <= 0 || (!out && !outT)) return A3V_ERR_ARG;
  const int Mpad = outT ? (M + 63) / 64 * 64 : M;
  if ((out && (ldo < 64 || ldo % 4)) || (outT && ldt < Mpad)) return A3V_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(partial) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return A3V_ERR_SHAPE;
  hipLaunchKernelGGL(skinny_reduce_kernel, dim3((Mpad * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, S, M, Mpad, (bf16_t*)out, ldo,
                     (bf16_t*)outT, ldt);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// P[s][R][N] = Tt[:R, token-slice s] . X[token-slice s, :N]; Tt = the [tokens, 64] operand transposed ([64][ldt], zero past Kt up to a
// multiple of 64).  Replaces the autograd products lora_b.weight.grad^T = t^T dy and lora_a.weight.grad = dt^T x (model/peft.py:58-159).
extern "C" int a3v_gemm_tn_strip2(const void* Tt, int64_t ldt, const void* X, int64_t ldx, float* partial, int R, int N, int Kt, int S,
                                  void* stream) {
  if (!Tt || !X || !partial || R <= 0 || N <= 0 || Kt <= 0 || S < 1 || S > 64) return A3V_ERR_ARG;
  const int64_t kpad = (int64_t)(Kt + 63) / 64 * 64;
  if (R > 64 || N % 4 || ldt % 8 || ldx % 8 || ldt < kpad || ldx < N || S > kpad / 64) return A3V_ERR_SHAPE;
  if ((63 * ldt + kpad) * 2 >= (1LL << 31) || ((int64_t)(Kt - 1) * ldx + N) * 2 >= (1LL << 31)) return A3V_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(Tt) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(partial)) & 15) return A3V_ERR_SHAPE;
  Strip2Args p{(const bf16_t*)Tt, (const bf16_t*)X, partial, ldt, ldx, R, N, Kt, A3V_ENV_INT("A3V_SK_ABL", 0)};
  const int wide = A3V_ENV_INT("A3V_STRIP2_WIDE", 1), st = A3V_ENV_INT("A3V_STRIP2_STAGES", 4);
  if (wide) {
    if (S > (Kt + 31) / 32) return A3V_ERR_SHAPE;
    const dim3 grid((N + 255) / 256, S);
    if (st == 3) hipLaunchKernelGGL((gemm_tn_strip2_kernel<3, 256>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_tn_strip2_kernel<4, 256>), grid, dim3(256), 0, (hipStream_t)stream, p);
  } else {
    const dim3 grid((N + 127) / 128, S);
    if (st == 3) hipLaunchKernelGGL((gemm_tn_strip2_kernel<3, 128>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_tn_strip2_kernel<4, 128>), grid, dim3(256), 0, (hipStream_t)stream, p);
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}
