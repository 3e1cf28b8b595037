// Decode step, the GEMV chain of one layer as ONE persistent launch (round 5; reference: LLM/llama_ens5.py:220-249 at seqlen == 1,
// the four projections around the attention of a3v_llama_decode_step):
//
//     h += att . Wo^T                      (residual; per-tile sums of squares of the new h)
//     act = SwiGLU(RMSNorm(h) . W13^T)     (interleaved gate / up image)
//     h += act . W2^T                      (residual; sums of squares)
//     qkv' = RMSNorm'(h) . Wqkv'^T         (the NEXT layer's projection: RoPE on q / k, k / v into its KV cache)        [not after the last layer]
//
// One block per CU (12 waves), co-resident for the whole launch.  Inside a phase a block owns the 16-row W tiles b, b + G, ... (SwiGLU:
// gate / up tile pairs); its waves are the K slices of a tile -- every wave streams 16 rows x its K slice through a private two-slot
// LDS ring by LDS-DMA (nt policy: each weight byte is read once per step), the partial accumulators meet in LDS and the tile's first
// wave finishes it (the arithmetic of gemv_kq_bf16_kernel / gemv_finish: same rounding points as the five-launch step).  A wave's
// stream runs on ACROSS the tiles of a phase (the next tile's first stages are in flight while the current one is reduced).
// Between phases every CU needs every CU's outputs: a grid barrier (ONE monotonic arrival counter) -- outputs leave
// through write-through (sc0 sc1) stores acknowledged before the arrival, the consumers invalidate their caches behind the barrier
// (agent-scope acquire) and read the activations through the L2 as usual.  The first two weight stages of a wave's first tile of
// the NEXT phase are issued BEFORE it waits at the barrier: weights do not depend on activations, so HBM keeps streaming through
// the hand-off.  A bounded spin (never a hang): a block that does not see its peers within ~0.5 s raises *err and goes on.
#include "a3v_common.h"
#include <type_traits>

namespace {

constexpr int CH_WAVES = 12;
constexpr int CH_RING = 2 * 4096;              // per wave: two slots of 16 rows x 256 B
constexpr int CH_RED = 1024;                   // per wave and parity: 64 lanes x 16 B of partial accumulators

struct ChainPhase {
  const bf16_t* A; int64_t lda;                // input rows [M][K]
  const bf16_t* W; int64_t ldw;                // [N][K] bf16
  bf16_t* C; int64_t ldc;
  const bf16_t* res; int64_t ldr;              // bf16 residual rows (kind 0)
  const bf16_t* norm_w;                        // RMSNorm weights of the prologue (nullptr: none)
  const float* ssq_in;                         // [K / 16][16] sums of squares of the rows of A (prologue)
  float* ssq_out;                              // kind 0: [N / 16][16] sums of squares of the rows written
  const float* cos_sin; bf16_t* k_cache; bf16_t* vt_cache;   // kind 2
  int N, K, kind;                              // 0: residual + ssq, 1: SwiGLU pairs, 2: qkv + RoPE + KV write
  int H, Hkv, hd, Smax, pos;
};

struct ChainArgs {
  ChainPhase ph[4];
  int nph, M;
  float eps;
  unsigned long long* bar;                     // monotonic arrival counter of the grid barriers
  int* err;
  int dbg;                                     // timing experiments (A3V_CHAIN_DBG): 1 no acquire fence, 2 fence by one wave per block, 4 no cross-phase prefetch, 8 plain stores
};

__device__ __forceinline__ void st_coh_b64(void* p, bf16x4 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_coh_b32f(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_coh_b16(bf16_t* p, bf16_t v) {
  const unsigned short u = __builtin_bit_cast(unsigned short, v);
  const unsigned w = u;
  asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ float ch_silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// epilogue of one finished tile (gemv_finish of a3v_gemm.hip with coherent stores): v[r] = D[n = nt0 + 4 (lane >> 4) + r][m = lane & 15]
__device__ __forceinline__ void chain_finish(const ChainPhase& p, int M, f32x4 v, f32x4 u, int nt0, int lane) {
  const int m = lane & 15;
  if (p.kind == 1) {
    if (m >= M || nt0 >= p.N) return;
    const int oc = (nt0 >> 1) + (lane >> 4) * 4;
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = f2bf(rbf(ch_silu(rbf(v[r]))) * rbf(u[r]));
    st_coh_b64(p.C + (int64_t)m * p.ldc + oc, o);
    return;
  }
  const int n = nt0 + (lane >> 4) * 4;
  float o4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o4[r] = rbf(v[r]);
  if (p.kind == 2) {
    if (m >= M || n >= p.N) return;
    const int slot = n / p.hd, d = n % p.hd, half = p.hd >> 1;
    if (slot < p.H + p.Hkv) {
      const float* cs = p.cos_sin + ((int64_t)p.pos * half + (d >> 1)) * 2;
      const f32x4 t = *reinterpret_cast<const f32x4*>(cs);
      bf16x4 o;
      o[0] = f2bf(o4[0] * t[0] - o4[1] * t[1]);
      o[1] = f2bf(o4[0] * t[1] + o4[1] * t[0]);
      o[2] = f2bf(o4[2] * t[2] - o4[3] * t[3]);
      o[3] = f2bf(o4[2] * t[3] + o4[3] * t[2]);
      bf16_t* dst = slot < p.H ? p.C + (int64_t)m * p.ldc + n : p.k_cache + (((int64_t)m * p.Hkv + (slot - p.H)) * p.Smax + p.pos) * p.hd + d;
      st_coh_b64(dst, o);
    } else {
      bf16_t* dst = p.vt_cache + (((int64_t)m * p.Hkv + (slot - p.H - p.Hkv)) * p.hd + d) * (int64_t)p.Smax + p.pos;
#pragma unroll
      for (int r = 0; r < 4; ++r) st_coh_b16(dst + (int64_t)r * p.Smax, f2bf(o4[r]));
    }
    return;
  }
  const bool live = m < M && n < p.N;
  if (live && p.res) {
    const bf16x4 rr = *reinterpret_cast<const bf16x4*>(p.res + (int64_t)m * p.ldr + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] += bf2f(rr[r]);
  }
  if (p.ssq_out) {
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float hb = rbf(o4[r]); sq = fmaf(hb, hb, sq); }
    if (!live) sq = 0.f;
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (lane < 16 && nt0 < p.N) st_coh_b32f(p.ssq_out + (nt0 >> 4) * 16 + lane, sq);
  }
  if (!live) return;
  bf16x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = f2bf(o4[r]);
  st_coh_b64(p.C + (int64_t)m * p.ldc + n, o);
}

// grid barrier on ONE monotonic 64-bit arrival counter (never reset; zero at first use, every launch uses the same grid size G, so it
// is a multiple of G between launches): the arrival that returns `old` belongs to barrier old / G, which is complete when the counter
// reaches (old / G + 1) G.  No second word, no reset, nothing to order besides the counter itself.
__device__ __forceinline__ unsigned long long chain_arrive(const ChainArgs& a) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's write-through stores are acknowledged
  __syncthreads();
  unsigned long long target = 0;
  if (threadIdx.x == 0) {
    const unsigned long long old = __hip_atomic_fetch_add(a.bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    target = (old / gridDim.x + 1) * gridDim.x;
  }
  return target;                                         // (meaningful in thread 0 only)
}
__device__ __forceinline__ void chain_wait(const ChainArgs& a, unsigned long long target) {
  if (threadIdx.x == 0) {
    int it = 0;
    while (__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++it > 400000) { *a.err = 1; break; }          // ~0.5 s: a peer never arrived (never a hang)
    }
  }
  if (a.dbg & 2) {
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    return;
  }
  __syncthreads();
  if (!(a.dbg & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the peers' outputs: nothing stale in this CU's L1 / this XCD's L2
}

template <bool PRO>
__device__ __forceinline__ void chain_phase(const ChainArgs& a, const ChainPhase& p, const ChainPhase& q, bool has_next, char* lds, float* rinv_s,
                                            float (*ssq_w)[8], bool prefetched) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int G = (int)gridDim.x;
  const bool swiglu = p.kind == 1;
  const int tpb = swiglu ? 2 : 1;
  const int nkb = p.K / 128;
  int slices = CH_WAVES / tpb;
  while (slices > 1 && nkb < 2 * slices) --slices;
  const int units = (p.N / 16) / tpb;
  const int my_units = (int)blockIdx.x < units ? (units - 1 - (int)blockIdx.x) / G + 1 : 0;
  const bool active = wave < slices * tpb;
  const int sl = swiglu ? (wave >> 1) : wave, tsel = swiglu ? (wave & 1) : 0;
  const int st0 = active ? (int)(((int64_t)sl * nkb) / slices) : 0, st1 = active ? (int)(((int64_t)(sl + 1) * nkb) / slices) : 0;
  const int nst = st1 - st0;
  char* Wring = lds + wave * CH_RING;
  float* red = reinterpret_cast<float*>(lds + CH_WAVES * CH_RING);         // [2 parities][CH_WAVES][64 lanes][4]
  const int dr = lane >> 4, dslot = lane & 15;
  const int fr = lane & 15, fg = lane >> 4;
  const int arow = (fr & 7) < a.M ? (fr & 7) : a.M - 1;
  const bf16_t* arp = p.A + (int64_t)arow * p.lda + (int64_t)st0 * 128 + fg * 8;
  const bf16_t* gp = PRO ? p.norm_w + (int64_t)st0 * 128 + fg * 8 : nullptr;
  // flat stream of this wave: item i = (unit i / nst, stage i % nst)
  const int total = active ? my_units * nst : 0;
  auto w_src = [&](int unit, int i, int st) -> const char* {
    const int row = 4 * i + dr;
    int wr = ((int)blockIdx.x + unit * G) * tpb * 16 + tsel * 16 + row;
    wr = wr < p.N ? wr : p.N - 1;
    return reinterpret_cast<const char*>(p.W) + (int64_t)wr * p.ldw * 2 + (int64_t)(st0 + st) * 256 + ((dslot ^ row) & 15) * 16;
  };
  auto dma_item = [&](int item) {
    const int unit = item / nst, st = item - unit * nst;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_src(unit, i, st),
                                       (__attribute__((address_space(3))) void*)(Wring + (item & 1) * 4096 + i * 1024), 16, 0, 2);
  };
  bf16x8 ax[2][4], ag[2][4];
  auto load_a = [&](int item, int set) {
    const int st = item % nst;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      ax[set][s4] = *reinterpret_cast<const bf16x8*>(arp + st * 128 + s4 * 32);
      if (PRO) ag[set][s4] = *reinterpret_cast<const bf16x8*>(gp + st * 128 + s4 * 32);
    }
  };
  // the weights of items 0 / 1 may already be in flight (issued before the barrier that opened this phase); the activations never are
  if (total > 0) { if (!prefetched) dma_item(0); }
  if (total > 1) { if (!prefetched) dma_item(1); }
  if (total > 0) load_a(0, 0);
  if (total > 1) load_a(1, 1);
  float ri = 1.f;
  if (PRO) {
    f32x4 sq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int ssq_tiles = p.K / 16;
    for (int t = threadIdx.x; t < ssq_tiles; t += blockDim.x) {
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
      for (int w = 0; w < CH_WAVES; ++w) t += ssq_w[w][threadIdx.x];
      rinv_s[threadIdx.x] = rsqrtf(t / (float)p.K + a.eps);
    }
    __syncthreads();
    ri = rinv_s[fr & 7];
  }
  int foff[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) foff[s4] = (((4 * s4 + fg) ^ fr) & 15) * 16;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int item = 0;
  auto body = [&](auto setc) {
    constexpr int SET = decltype(setc)::value;
    // W(item) and A(item) have landed when at most the next item's pieces are outstanding (loads retire in order; a wave's own
    // epilogue stores in between only make the count stricter)
    if (item + 1 < total) { if (PRO) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
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
    if (item + 2 < total) { load_a(item + 2, SET); dma_item(item + 2); }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if (s4 & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4], acc0, 0, 0, 0);
    }
  };
  for (int u = 0; u < my_units; ++u) {
    if (active) {
      for (int s = 0; s < nst; ++s) {
        if (item & 1) body(std::integral_constant<int, 1>{}); else body(std::integral_constant<int, 0>{});
        ++item;
      }
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc0[r] + acc1[r];
      *reinterpret_cast<f32x4*>(red + (((u & 1) * CH_WAVES + wave) * 64 + lane) * 4) = v;
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (wave == 0) {
      // the tile's first wave sums the slices in slice order (deterministic) and finishes the tile
      f32x4 v = {0.f, 0.f, 0.f, 0.f}, uu = {0.f, 0.f, 0.f, 0.f};
      const float* rb = red + (u & 1) * CH_WAVES * 64 * 4;
      for (int s8 = 0; s8 < slices; ++s8) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(rb + ((s8 * tpb) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += x[r];
        if (swiglu) {
          const f32x4 y = *reinterpret_cast<const f32x4*>(rb + ((s8 * 2 + 1) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) uu[r] += y[r];
        }
      }
      chain_finish(p, a.M, v, uu, ((int)blockIdx.x + u * G) * tpb * 16, lane);
    }
  }
  if (!has_next) return;
  // ---- hand-off: arrive, put the next phase's first weight stages in flight, then wait for the peers
  const unsigned long long target = chain_arrive(a);
  {
    const bool sw2 = q.kind == 1;
    const int tpb2 = sw2 ? 2 : 1, nkb2 = q.K / 128;
    int sl2n = CH_WAVES / tpb2;
    while (sl2n > 1 && nkb2 < 2 * sl2n) --sl2n;
    const int units2 = (q.N / 16) / tpb2;
    const bool act2 = wave < sl2n * tpb2 && (int)blockIdx.x < units2 && !(a.dbg & 4);
    if (act2) {
      const int s2 = sw2 ? (wave >> 1) : wave, ts2 = sw2 ? (wave & 1) : 0;
      const int a0 = (int)(((int64_t)s2 * nkb2) / sl2n), a1 = (int)(((int64_t)(s2 + 1) * nkb2) / sl2n);
      const int my2 = (units2 - 1 - (int)blockIdx.x) / G + 1, n2 = a1 - a0, tot2 = my2 * n2;
      for (int it2 = 0; it2 < 2 && it2 < tot2; ++it2) {
        const int unit = it2 / n2, st = it2 - unit * n2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 4 * i + dr;
          int wr = ((int)blockIdx.x + unit * G) * tpb2 * 16 + ts2 * 16 + row;
          wr = wr < q.N ? wr : q.N - 1;
          const char* src = reinterpret_cast<const char*>(q.W) + (int64_t)wr * q.ldw * 2 + (int64_t)(a0 + st) * 256 + ((dslot ^ row) & 15) * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(Wring + (it2 & 1) * 4096 + i * 1024), 16, 0, 2);
        }
      }
    }
  }
  chain_wait(a, target);
}

__global__ __launch_bounds__(CH_WAVES * 64) void decode_chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char ch_lds[];
  __shared__ float rinv_s[8];
  __shared__ float ssq_w[CH_WAVES][8];
  // phase 0: att . Wo^T (+ residual); 1: SwiGLU(norm(h) . W13^T); 2: act . W2^T (+ residual); 3: norm'(h) . Wqkv'^T of the next layer
  chain_phase<false>(a, a.ph[0], a.ph[1], true, ch_lds, rinv_s, ssq_w, false);
  chain_phase<true>(a, a.ph[1], a.ph[2], true, ch_lds, rinv_s, ssq_w, !(a.dbg & 4));
  chain_phase<false>(a, a.ph[2], a.ph[3], a.nph > 3, ch_lds, rinv_s, ssq_w, !(a.dbg & 4));
  if (a.nph > 3) chain_phase<true>(a, a.ph[3], a.ph[3], false, ch_lds, rinv_s, ssq_w, !(a.dbg & 4));
}

}  // namespace

// One layer's GEMV chain of the decode step (see the header of this file).  qkv_next == NULL: the last layer (three phases).
// Returns A3V_ERR_SHAPE when the geometry is not taken (the caller runs the five-launch form).
int a3v_decode_chain(const a3v_llama_layer* L, const a3v_llama_layer* Lnext, void* h, void* qkv, void* att, void* act, float* ssq, void* ws,
                     const float* cos_sin, int B, int dim, int H, int Hkv, int hd, int ffn, int Smax, int pos, float eps, void* stream) {
  if (B <= 0 || B > 8 || dim % 128 || (H * hd) % 128 || ffn % 128 || (2 * ffn) % 32 || dim % 16) return A3V_ERR_SHAPE;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return A3V_ERR_SHAPE;
  const int64_t ldq = (int64_t)(H + 2 * Hkv) * hd;
  ChainArgs a{};
  a.M = B; a.eps = eps; a.dbg = A3V_ENV_INT("A3V_CHAIN_DBG", 0);
  a.bar = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + A3V_WS_ATTN_COUNTERS - 16);
  a.err = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + A3V_WS_ATTN_COUNTERS - 8);
  ChainPhase& p0 = a.ph[0];
  p0 = ChainPhase{};
  p0.A = (const bf16_t*)att; p0.lda = (int64_t)H * hd; p0.W = (const bf16_t*)L->wo; p0.ldw = (int64_t)H * hd; p0.C = (bf16_t*)h; p0.ldc = dim;
  p0.res = (const bf16_t*)h; p0.ldr = dim; p0.ssq_out = ssq; p0.N = dim; p0.K = H * hd; p0.kind = 0;
  ChainPhase& p1 = a.ph[1];
  p1 = ChainPhase{};
  p1.A = (const bf16_t*)h; p1.lda = dim; p1.W = (const bf16_t*)L->w13; p1.ldw = dim; p1.C = (bf16_t*)act; p1.ldc = ffn; p1.norm_w = (const bf16_t*)L->ffn_norm_w;
  p1.ssq_in = ssq; p1.N = 2 * ffn; p1.K = dim; p1.kind = 1;
  ChainPhase& p2 = a.ph[2];
  p2 = ChainPhase{};
  p2.A = (const bf16_t*)act; p2.lda = ffn; p2.W = (const bf16_t*)L->w2; p2.ldw = ffn; p2.C = (bf16_t*)h; p2.ldc = dim; p2.res = (const bf16_t*)h; p2.ldr = dim;
  p2.ssq_out = ssq; p2.N = dim; p2.K = ffn; p2.kind = 0;
  a.nph = 3;
  if (Lnext) {
    ChainPhase& p3 = a.ph[3];
    p3 = ChainPhase{};
    p3.A = (const bf16_t*)h; p3.lda = dim; p3.W = (const bf16_t*)Lnext->wqkv; p3.ldw = dim; p3.C = (bf16_t*)qkv; p3.ldc = ldq; p3.norm_w = (const bf16_t*)Lnext->attn_norm_w;
    p3.ssq_in = ssq; p3.N = (int)ldq; p3.K = dim; p3.kind = 2;
    p3.cos_sin = cos_sin; p3.k_cache = (bf16_t*)Lnext->k_cache; p3.vt_cache = (bf16_t*)Lnext->vt_cache; p3.H = H; p3.Hkv = Hkv; p3.hd = hd; p3.Smax = Smax; p3.pos = pos;
    a.nph = 4;
  }
  const size_t lds = (size_t)CH_WAVES * CH_RING + 2 * CH_WAVES * CH_RED;
  static bool attr[A3V_MAX_DEV][1] = {};
  const int rc = a3v_dyn_lds_once(attr, 0, (const void*)decode_chain_kernel, 150 * 1024);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(decode_chain_kernel, dim3(cus), dim3(CH_WAVES * 64), lds, (hipStream_t)stream, a);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}
