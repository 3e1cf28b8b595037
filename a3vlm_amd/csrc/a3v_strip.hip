// Adapter-sized weight gradients ("strips") for gfx950: out[R, N] = T[Kt, R]^T . X[Kt, N] with R <= 64 (the padded LoRA rank of a
// fused adapter group) and the contraction over the Kt token rows -- dB^T = t^T . dy and dA = dt^T . x of model/peft.py:58-159 as
// autograd forms them for lora_b / lora_a (engine_finetune.py:55-57), on the token-major tensors the step holds.
//
// These products are pure HBM streams of X (71 .. 384 MB per call at the 7B shapes; 2 R N Kt FLOPs are nothing), and what bounds a
// stream on this part is how many bytes each CU has in flight.  The 256 x 256 ping-pong TN kernel (a3v_gemm_tn_splitk) keeps two 64-KiB
// stages per block = ONE block per CU with one 40-KiB useful stage in flight: 2.4 - 3.5 TB/s.  Here a block is small -- 4 waves, a
// 64 x 128 output tile, two stages of 24 KiB ([64 k][64 m] of T in 128-byte rows + [64 k][128 n] of X in 256-byte rows) -- so that
// three blocks share a CU (72 KiB in flight) and the grid is (N / 128) x S blocks over S token slices.
//
//  * staging: LDS-DMA (buffer_load ... lds, 16 B per lane, 1 KiB per wave instruction = 8 rows of T or 4 rows of X); tokens past Kt
//    read as zeros through the buffer descriptor (ragged last k-tile);
//  * MFMA operands need 8 CONSECUTIVE k (tokens) of one output row / column per lane, i.e. a column of the row-major tile:
//    ds_read_b64_tr_b16 (16 lanes pass the addresses of a 4 x 16 block, lane c receives column c; semantics pinned by
//    tools/ubench/trread.hip and used the same way by gemm_tn_bf16_pp_kernel).  A half-wave reads two such blocks per LDS cycle
//    (k-rows r..r+3 and r+8..r+11 of one 32-byte chunk): the chunk index is XOR-ed with a key of the row so that the eight 32-byte
//    pieces hit eight different bank groups -- 256-byte rows: key = 4 (k>>3 & 1) + (k & 3); 128-byte rows: key = 2 (k>>3 & 1) + (k>>1 & 1)
//    (a 128-byte row covers half of the 64 banks, so rows of equal parity must differ in the chunk).  The permutation is applied on
//    the DMA SOURCE address (DMA destinations are lane-linear) and again on the read address.
//  * output: raw fp32 partial planes [S][R][N] (16-byte stores, a lane owns 4 consecutive n of one m), summed, rounded to bf16 once
//    and stored / accumulated by a3v_splitk_reduce exactly like the planes of a3v_gemm_tn_splitk.
#include "a3v_common.h"

namespace {
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int SBK = 64;                  // tokens per k-tile
constexpr int SBN = 128;                 // output columns per block
constexpr int A_TILE = SBK * 128;        // [64 k][64 m] bf16, 128-byte rows: 8 KiB
constexpr int W_TILE = SBK * 256;        // [64 k][128 n] bf16, 256-byte rows: 16 KiB
constexpr int S_STAGE = A_TILE + W_TILE; // 24 KiB

struct StripArgs {
  const bf16_t* T;     // [Kt][ldt], R valid columns
  const bf16_t* X;     // [Kt][ldx], N columns
  float* part;         // [S][R][N]
  int64_t ldt, ldx;
  int R, N, Kt;
};

__device__ __forceinline__ int key_a(int k) { return (((k >> 3) & 1) << 1) | ((k >> 1) & 1); }   // 128-byte rows, 4 chunks
__device__ __forceinline__ int key_w(int k) { return (((k >> 3) & 1) << 2) | (k & 3); }          // 256-byte rows, 8 chunks

template <int NST>     // LDS stages: 2 (48 KiB, three blocks per CU, one tile of lead) or 3 (72 KiB, two blocks per CU, two tiles of lead)
__global__ __launch_bounds__(256) void gemm_tn_strip_kernel(StripArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[NST * S_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * SBN;
  const int nk_all = (p.Kt + SBK - 1) / SBK;
  const int kt0 = (int)((int64_t)nk_all * blockIdx.y / gridDim.y), kt1 = (int)((int64_t)nk_all * (blockIdx.y + 1) / gridDim.y);
  const auto rsT = __builtin_amdgcn_make_buffer_rsrc((void*)p.T, 0, (int)(((int64_t)(p.Kt - 1) * p.ldt + 64) * 2), 0x00020000);
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)(((int64_t)(p.Kt - 1) * p.ldx + p.N) * 2), 0x00020000);
  // DMA pieces of this wave: T tile = 8 pieces of 8 rows (2 per wave), X tile = 16 pieces of 4 rows (4 per wave)
  unsigned voT[2], voX[4];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int row = (wave * 2 + c) * 8 + (lane >> 3), pos = lane & 7;
    voT[c] = (unsigned)((row * p.ldt) * 2 + ((((pos >> 1) ^ key_a(row)) << 5) + (pos & 1) * 16));
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int row = (wave * 4 + c) * 4 + (lane >> 4), pos = lane & 15;
    voX[c] = (unsigned)((row * p.ldx) * 2 + ((((pos >> 1) ^ key_w(row)) << 5) + (pos & 1) * 16));
  }
  auto stage = [&](int t) {
    char* base = lds + ((t - kt0) % NST) * S_STAGE;
    const unsigned soT = (unsigned)(((int64_t)t * SBK * p.ldt) * 2);
    const unsigned soX = (unsigned)(((int64_t)t * SBK * p.ldx + n0) * 2);
#pragma unroll
    for (int c = 0; c < 2; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, (__attribute__((address_space(3))) void*)(base + (wave * 2 + c) * 1024), 16, voT[c] + soT, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(base + A_TILE + (wave * 4 + c) * 1024), 16, voX[c] + soX, 0, 0, 0);
  };
  // fragment addressing (as gemm_tn_bf16_pp_kernel): lane (fg = lane >> 4, il = lane & 15); sub-block (ks, jj) = k-rows
  // 32 ks + 8 fg + 4 jj + (il >> 2); the lane points at the 8-byte piece (il & 3) of its row's 32-byte chunk
  const int fg = lane >> 4, il = lane & 15;
  int roA[2][2], kA[2][2], roW[2][2], kW[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int k = ks * 32 + 8 * fg + 4 * jj + (il >> 2);
      roA[ks][jj] = k * 128 + (il & 3) * 8;
      kA[ks][jj] = key_a(k);
      roW[ks][jj] = k * 256 + (il & 3) * 8;
      kW[ks][jj] = key_w(k);
    }
  auto tr8 = [&](const char* tile, int ro0, int ro1, int x0, int x1) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + ro0 + (x0 << 5)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + ro1 + (x1 << 5)));
    bf16x8 r;
    const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    __builtin_memcpy(&r, v, 16);
    return r;
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) stage(kt0);
  if (NST == 3 && kt0 + 1 < kt1) stage(kt0 + 1);
  for (int t = kt0; t < kt1; ++t) {
    // this wave's pieces of tile t have landed (loads retire in order; with three stages tile t+1's six pieces may stay in flight) ...
    if (NST == 3 && t + 1 < kt1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // ... and everybody's; all reads of tile t-1 (the buffer the next stage goes to) are done
    if (t + NST - 1 < kt1) stage(t + NST - 1);
    const char* At = lds + ((t - kt0) % NST) * S_STAGE;
    const char* Wt = At + A_TILE;
    bf16x8 af[2][4], wf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = tr8(At, roA[ks][0], roA[ks][1], i ^ kA[ks][0], i ^ kA[ks][1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[ks][j] = tr8(Wt, roW[ks][0], roW[ks][1], (wave * 2 + j) ^ kW[ks][0], (wave * 2 + j) ^ kW[ks][1]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
  }
  // D = W_frag x A_frag: the lane holds C[m = 16 i + (lane & 15)][n = n0 + 32 wave + 16 j + 4 (lane >> 4) + 0..3]
  float* plane = p.part + (int64_t)blockIdx.y * p.R * p.N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = 16 * i + (lane & 15);
    if (m >= p.R) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 32 * wave + 16 * j + 4 * (lane >> 4);
      if (n + 4 <= p.N) *reinterpret_cast<f32x4*>(plane + (int64_t)m * p.N + n) = acc[i][j];
    }
  }
}
}  // namespace

extern "C" int a3v_gemm_tn_strip(const void* T, int64_t ldt, const void* X, int64_t ldx, float* partial, int R, int N, int Kt, int S,
                                 void* stream) {
  if (!T || !X || !partial || R <= 0 || N <= 0 || Kt <= 0 || S < 1 || S > 64) return A3V_ERR_ARG;
  if (R > 64 || R % 4 || N % 4 || ldt % 8 || ldx % 8 || ldt < 64 || S > (Kt + SBK - 1) / SBK) return A3V_ERR_SHAPE;
  if (((int64_t)(Kt - 1) * ldt + 64) * 2 >= (1LL << 31) || ((int64_t)(Kt - 1) * ldx + N) * 2 >= (1LL << 31)) return A3V_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(partial)) & 15) return A3V_ERR_SHAPE;
  StripArgs p{(const bf16_t*)T, (const bf16_t*)X, partial, ldt, ldx, R, N, Kt};
  if (A3V_ENV_INT("A3V_STRIP_STAGES", 2) == 3) hipLaunchKernelGGL(gemm_tn_strip_kernel<3>, dim3((N + SBN - 1) / SBN, S), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_tn_strip_kernel<2>, dim3((N + SBN - 1) / SBN, S), dim3(256), 0, (hipStream_t)stream, p);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}
