// Shared device helpers for the gfx950 kernels (wave64, MFMA, bf16 <-> f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a3vlm_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define A3V_WAVE 64

// ---- A3V_* environment switches (A/B runs, tuning scripts and equality tests; the defaults are the product path and none of
// them selects a CPU path).  A switch is read ONCE per call site and cached: no launch calls getenv in steady state.  A process
// that flips a switch after the first launch calls a3v_reload_env() (C-ABI; a3vlm_amd.lib.env(...) does it) to have them re-read.
int a3v_env_generation();
#include <stdlib.h>
#define A3V_ENV_INT(name, dflt)                                                        \
  ([]() -> int {                                                                       \
    static int gen_ = -1, v_ = 0;                                                      \
    const int g_ = a3v_env_generation();                                               \
    if (gen_ != g_) { const char* e_ = getenv(name); v_ = e_ ? atoi(e_) : (dflt); gen_ = g_; } \
    return v_;                                                                         \
  }())

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }  // RNE (v_cvt_pk_bf16_f32)
// d(gate), d(up) of act = silu(gate) * up for one element (a3v_swiglu_bwd and the A3V_EPI_SWIGLU_BWD GEMM epilogue share this
// expression so that the fused and the un-fused path round identically)
__device__ __forceinline__ void swiglu_bwd_pair(float g, float u, float da, float& dg, float& du) {
  const float sig = 1.f / (1.f + __expf(-g));
  dg = da * u * (sig * (1.f + g * (1.f - sig)));
  du = da * (g * sig);
}
// round-trip: the value a bf16 store of v would hold
__device__ __forceinline__ float rbf(float v) { return (float)((bf16_t)v); }

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
  static __device__ __forceinline__ float rnd(float v) { return rbf(v); }
};

// 8 contiguous elements <-> 8 floats
__device__ __forceinline__ void load8(const bf16_t* p, float* o) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
__device__ __forceinline__ void load8(const float* p, float* o) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
}
// non-temporal forms for streams that are touched once (NT is a compile-time flag of the calling kernel)
template <bool NT>
__device__ __forceinline__ void load8_s(const bf16_t* p, float* o) {
  const bf16x8 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)) : *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <bool NT>
__device__ __forceinline__ void load8_s(const float* p, float* o) {
  const f32x4 a = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4)) : *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
}
template <bool NT>
__device__ __forceinline__ void store8_s(bf16_t* p, const float* v) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  if (NT) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p)); else *reinterpret_cast<bf16x8*>(p) = o;
}
template <bool NT>
__device__ __forceinline__ void store8_s(float* p, const float* v) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  if (NT) { __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p)); __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(p + 4)); }
  else { *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b; }
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ void store8(float* p, const float* v) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

// Wave-wide butterfly reductions WITHOUT the LDS pipe (round 4).  __shfl_xor compiles to ds_bpermute_b32: six LDS round trips of ~100+
// cycles each per reduction, in front of every row's second pass and in the prologue of every decode GEMV.  The same butterfly
// (partner lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 in this order -- the same pairs, hence the same bits as the __shfl_xor form) on the VALU:
//   ^ 32 / ^ 16 : v_permlane32_swap / v_permlane16_swap (asm: with one value in both operands of the builtins hipcc keeps ONE of the
//                 two results for both and the exchange disappears);
//   ^ 8  / ^ 4  : DPP row_ror:8 / row_ror:4 (after the ^ 8 level the lanes i and i ^ 8 of a row hold the same value, so lane i + 4
//                 mod 16 holds what lane i ^ 4 holds);   ^ 2 / ^ 1 : DPP quad_perm [2,3,0,1] / [1,0,3,2].
__device__ __forceinline__ void lane_xchg32(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_xchg16(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#ifdef A3V_WAVE_SHFL      // A/B builds: the __shfl_xor (ds_bpermute) forms of rounds 1-3
#define wave_sum wave_sum_shfl
#define wave_max wave_max_shfl
#else
__device__ __forceinline__ float wave_sum(float v) {
  float a, b;
  lane_xchg32(v, a, b); v = a + b;
  lane_xchg16(v, a, b); v = a + b;
  v += dpp_f<0x128>(v);
  v += dpp_f<0x124>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0xB1>(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  float a, b;
  lane_xchg32(v, a, b); v = fmaxf(a, b);
  lane_xchg16(v, a, b); v = fmaxf(a, b);
  v = fmaxf(v, dpp_f<0x128>(v));
  v = fmaxf(v, dpp_f<0x124>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0xB1>(v));
  return v;
}
#endif
// (the LDS-pipe forms, kept for the equality test of the two: a3v_probe_wave_reduce)
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x = NT (multiple of 64); red: NT/64 floats of LDS
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if (NT == 64) return v;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// Opt a kernel into more than 64 KiB of dynamic LDS ONCE PER DEVICE (the attribute is per device; a process-wide "done" flag left a
// second GPU of the process without it: launch failure).  `done`: a per-call-site table [A3V_MAX_DEV][slots]; the set is idempotent, so
// two host threads racing through the flag at worst both make the call.  Returns the hipError_t of the set (0 = fine).
constexpr int A3V_MAX_DEV = 16;
template <int SLOTS>
inline int a3v_dyn_lds_once(bool (&done)[A3V_MAX_DEV][SLOTS], int slot, const void* kern, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= A3V_MAX_DEV) dev = 0, done[0][slot] = false;
  if (done[dev][slot]) return 0;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done[dev][slot] = true;
  return 0;
}

#define A3V_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e_ = hipGetLastError();             \
    if (e_ != hipSuccess) return (int)e_;          \
  } while (0)

// Agent-coherent (sc0 sc1) scalar accesses for in-kernel hand-offs between workgroups that may sit on different XCDs
// (L2 is per XCD): the store is written through, the load bypasses stale lines; no cache-wide write-back / invalidate.
// ld_agent_issue only ISSUES the load: the caller must `s_waitcnt vmcnt(0)` (agent_wait) before using the value.
__device__ __forceinline__ void st_agent(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ float ld_agent_issue(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void agent_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ float ld_agent(const float* p) {          // issue + wait (the value is tied to the wait)
  float v = ld_agent_issue(p);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory");
  return v;
}

// Layout of the decode workspace (a3v_gemm_skinny `partial` / a3v_llama_decode_step `skinny_ws`): zero-filled once by the
// caller; every kernel leaves the counter areas zero.
//   [0, 16 KB)      GEMV split-K arrival counters (one per 16-row tile, N <= 65536)
//   [16 KB, 32 KB)  decode-attention arrival counters (one per (batch, head))
//   [32 KB, 64 KB)  per-tile sums of squares of the residual rows (RMSNorm fused into the consuming GEMV): dim / 16 tiles x 16 rows x 4 B,
//                   i.e. dim <= 8192 (round 4: was 16 KB = dim <= 4096, which kept the 13B geometry, dim 5120, out of the fused decode step)
//   [64 KB, ...)    split-K partial accumulators
constexpr int A3V_WS_ATTN_COUNTERS = 16384;
constexpr int A3V_WS_SSQ = 32768;
constexpr int A3V_WS_PARTIALS = 65536;
int a3v_gemv_fused(const void* A, int64_t lda, const void* W, int64_t ldw, const float* wscale, void* C, int64_t ldc, int M, int N,
                   int K, const void* residual, int64_t ldr, int epilogue, const void* norm_w, const float* ssq_in, float eps,
                   float* ssq_out, int rope, const float* cos_sin, void* k_cache, void* vt_cache, int H, int Hkv, int hd,
                   int Smax, int pos, void* ws, void* stream);
bool a3v_gemv_supported(int M, int N, int K, int epilogue, int w8);
int a3v_attention_decode_fused(const void* q, const void* k, const void* vt, void* out, int B, int Sk, int H, int Hkv, int hd,
                               const int64_t* strides, float* scratch, int* counters, void* stream);
// internal (a3v_train.hip): bf16 transposes of n_out x n_in matrices [R, C] -> [C, Rpad] in one launch
int a3v_transpose_2level(const bf16_t* src, int64_t ld_src, int64_t bs_in, int64_t bs_out, bf16_t* dst, int64_t ld_dst, int64_t bsd_in,
                         int64_t bsd_out, int R, int C, int Rpad, int n_in, int n_out, void* stream);
