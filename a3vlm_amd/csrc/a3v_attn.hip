// Attention for gfx950: softmax(q k^T / sqrt(hd) [+ right-aligned causal]) v.
//
// Layout contract (elements, explicit strides): q/out [b][s][h][hd]; k [b][hkv][s][hd];
// vt = V TRANSPOSED [b][hkv][hd][s].  V^T lets both MFMA products run "transposed":
//     S^T = K . Q^T   (A = K rows from LDS, B = Q rows from registers)
//     O^T = V^T . P^T (A = V^T rows from LDS, B = P straight from the S^T accumulators)
// With v_mfma_f32_32x32x16_bf16 the C/D map is col = lane&31, row = (r&3)+8(r>>2)+4(lane>>5),
// so a lane owns ONE query column of S^T and of O^T: the online softmax (max, exp, sum,
// rescale) is lane-local plus one lane<->lane+32 exchange, and P never leaves registers --
// the 16 accumulator values of a lane are, in register order, exactly the k-elements the
// P^T B-operand wants when V^T rows are read at kv = base + 4*(lane>>5) + {0..3, 8..11}.
//
//  * attn_prefill_bf16_kernel<HD,CAUSAL>: 4 waves x 32 query rows, KV tiles of 64 through
//    LDS (K rows XOR-swizzled for ds_read_b128, V^T rows for ds_read_b64), two waves per SIMD.
//  * attn_decode_bf16_kernel<HD>: Sq == 1, split-KV (flash-decoding), HBM-bound streaming of
//    K rows / V^T rows with 16-byte loads, + combine kernel.
//  * attn_f32_kernel: fp32 parity path (any Sq), one wave per (b, h, q).
#include "a3v_common.h"
#include <type_traits>
#include <cstdlib>
#include <atomic>
#include <mutex>

namespace {

struct AttnArgs {
  const void* q; const void* k; const void* vt; void* out;
  int64_t q_sb, q_ss, q_sh;     // q strides: batch, seq, head
  int64_t k_sb, k_sh, k_ss;     // k strides: batch, kv-head, seq   (hd contiguous)
  int64_t v_sb, v_sh, v_sd;     // vt strides: batch, kv-head, d    (seq contiguous)
  int64_t o_sb, o_ss, o_sh;
  int B, Sq, Sk, H, Hkv;
  float scale_log2;             // log2(e) / sqrt(hd)
  float scale;
  float* lse;                   // optional [B,H,Sq] log-sum-exp of the scaled scores (training backward)
  int head_group;               // causal prefill: heads per tile-rank-major group of the block order (1 = head-major)
  int staged_o;                 // prefill: O leaves through an LDS patch as whole rows (A3V_ATTN_STAGED_O=0: per-lane row stores)
  int lazy_rescale;             // prefill: the softmax reference only moves when a row's exponent would exceed 2^8 (A3V_ATTN_LAZY=0: every tile)
};

// one 16-B-per-lane LDS-DMA through a buffer descriptor: per-lane byte offset + wave-uniform byte offset (an SGPR)
__device__ __forceinline__ void buf_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

// own value and the value of lane ^ 32 in (a, b) (lanes < 32) / (b, a) (lanes >= 32).  Asm form: with the same value in both operands
// of __builtin_amdgcn_permlane32_swap hipcc uses ONE of the two results for both (v_max v, v, v in the ISA) and the exchange disappears.
__device__ __forceinline__ void xchg32(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// PSWAP: the P^T operand is regrouped so that a lane half holds 8 CONSECUTIVE keys (two v_permlane32_swap per 16 keys exchange the
// {4 hh .. 4 hh + 3} quarters between lanes l and l + 32), and the V^T fragment becomes ONE 16-B chunk per lane (ds_read_b128,
// lanes 0-31 chunk c, lanes 32-63 chunk c + 1: the 16 lanes of a read group cover 16 distinct bank slots) instead of two 8-B
// halves of two chunks (ds_read_b64 pairs where rows r and r + 16 of a lane half share their banks: 2-way on every read,
// 1.18e7 conflict cycles per launch against 3.2e6 active LDS cycles in profiles/r01u_pmc_decode_lds.txt).
template <int HD, bool CAUSAL, bool PSWAP = true>
__global__ __launch_bounds__(256, 2) void attn_prefill_bf16_kernel(AttnArgs p) {
  constexpr int KVB = 64;
  constexpr int KROW = HD * 2;            // bytes per K row in LDS
  constexpr int KCH = HD / 8;             // 16-B chunks per K row
  constexpr int K_LOADS = KVB * KCH / 256;   // 16-B chunks per thread (4 for HD=128, 2 for 64)
  constexpr int V_LOADS = HD * 8 / 256;      // V^T tile: HD rows x 8 chunks of 16 B
  constexpr int TILEB = KVB * KROW + HD * 128;   // K tile + V^T tile
  __shared__ __attribute__((aligned(1024))) char lds[2 * TILEB];   // double buffered: tile t+1 lands by LDS-DMA while tile t is consumed

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, XCD-aware: workgroups go to XCDs round-robin (id & 7), and every query tile of a (batch, head) reads the same
  // K / V^T, so XCD x takes a CONTIGUOUS range of (head, query tile) pairs -- a head's K/V then lives in ONE 4-MB L2 instead
  // of being fetched into all eight.  Within a head the heavy (late) causal tiles go first.
  const int nqt = (p.Sq + 127) / 128;
  int vb;
  {
    const int total = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, q = total >> 3, r = total & 7;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  int head_slot = vb / nqt;
  int qt = nqt - 1 - (vb - head_slot * nqt);
  if (CAUSAL && p.head_group > 1) {
    // causal tiles cost 2, 4, ..., 2 nqt KV tiles: with "heavy first" per head the long blocks of the XCD's last heads start late
    // and run alone (a list-scheduling simulation of the 7B prefill: 129 us against 103 for perfect packing).  Groups of
    // `head_group` heads are walked tile-rank-major instead (every head's heaviest tile, then every head's second, ...): the
    // group's K / V (head_group x 0.56 MB at S = 1091) still sits in the XCD's L2 while its tiles run.
    const int G = p.head_group, per = G * nqt;
    const int grp = vb / per, r = vb - grp * per;
    head_slot = grp * G + r % G;
    qt = nqt - 1 - r / G;
  }
  const int b = head_slot / p.H, h = head_slot - b * p.H;
  const int hk = h / (p.H / p.Hkv);
  const int q0 = qt * 128 + wave * 32;
  const int off = p.Sk - p.Sq;                 // right alignment (llama_ens5.py:181-185)

  const bf16_t* Q = (const bf16_t*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16_t* K = (const bf16_t*)p.k + b * p.k_sb + hk * p.k_sh;
  const bf16_t* VT = (const bf16_t*)p.vt + b * p.v_sb + hk * p.v_sh;

  const int ql = lane & 31, hh = lane >> 5;
  int qrow = q0 + ql;
  const int qrow_c = qrow < p.Sq ? qrow : p.Sq - 1;
  // Q fragments (B operand): Q[q][16*ks + 8*hh + e]
  bf16x8 qf[HD / 16];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qrow_c * p.q_ss + ks * 16 + hh * 8);

  f32x16 o[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // kv range for this block: all q rows of the block
  int kv_end = p.Sk;
  if (CAUSAL) {
    const int last_q = min(qt * 128 + 127, p.Sq - 1);
    kv_end = min(p.Sk, last_q + off + 1);
  }
  const int n_tiles = (kv_end + KVB - 1) / KVB;

  u32x4 kreg[K_LOADS], vreg[V_LOADS];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < K_LOADS; ++i) {
      const int id = tid + i * 256;
      const int row = id / KCH, ch = id % KCH;
      int kr = kv0 + row;
      kr = kr < p.Sk ? kr : p.Sk - 1;
      kreg[i] = *reinterpret_cast<const u32x4*>(K + (int64_t)kr * p.k_ss + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < V_LOADS; ++i) {
      const int id = tid + i * 256;
      const int d = id >> 3, ch = id & 7;
      const int kv = kv0 + ch * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (kv + 8 <= p.Sk) {
        v = *reinterpret_cast<const u32x4*>(VT + (int64_t)d * p.v_sd + kv);
      } else if (kv < p.Sk) {   // ragged tail: zero the columns >= Sk (0 * garbage must stay 0)
        const unsigned short* src = reinterpret_cast<const unsigned short*>(VT + (int64_t)d * p.v_sd + kv);
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (kv + j < p.Sk) ? src[j] : (unsigned short)0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (unsigned)e[2 * j] | ((unsigned)e[2 * j + 1] << 16);
      }
      vreg[i] = v;
    }
  };
  auto write_tile = [&](char* Ks, char* Vs) {
#pragma unroll
    for (int i = 0; i < K_LOADS; ++i) {
      const int id = tid + i * 256;
      const int row = id / KCH, ch = id % KCH;
      const int sw = (HD == 128) ? (row & 15) : ((row >> 1) & 7);
      *reinterpret_cast<u32x4*>(Ks + row * KROW + ((ch ^ sw) << 4)) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < V_LOADS; ++i) {
      const int id = tid + i * 256;
      const int d = id >> 3, ch = id & 7;
      *reinterpret_cast<u32x4*>(Vs + d * 128 + ((ch ^ ((d >> 1) & 7)) << 4)) = vreg[i];
    }
  };
  // LDS-DMA form of load_tile + write_tile for tiles whose 64 keys all exist (kv0 + 64 <= Sk): the chunk permutation is
  // applied on the per-lane SOURCE offset (DMA destinations are lane-linear), no VGPRs are held while the tile is in flight.
  // Buffer-descriptor form: the per-lane byte offsets are loop constants, the tile's position is one scalar offset per
  // instruction -- issuing a tile costs 8 VMEM instructions and no address VALU (the flat-address form spent ~500 cycles per tile).
  const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, 0x7fffffff, 0x00020000);
  const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)VT, 0, 0x7fffffff, 0x00020000);
  unsigned koff[K_LOADS], voff[V_LOADS];
#pragma unroll
  for (int i = 0; i < K_LOADS; ++i) {
    const int id = tid + i * 256;
    const int row = id / KCH, slot = id % KCH;
    const int sw = (HD == 128) ? (row & 15) : ((row >> 1) & 7);
    koff[i] = (unsigned)((row * p.k_ss + (slot ^ sw) * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < V_LOADS; ++i) {
    const int id = tid + i * 256;
    const int d = id >> 3, slot = id & 7;
    voff[i] = (unsigned)((d * p.v_sd + (slot ^ ((d >> 1) & 7)) * 8) * 2);
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_tile = [&](int kv0, char* Ks, char* Vs) {
    const unsigned ks_off = (unsigned)(kv0 * p.k_ss * 2), vs_off = (unsigned)(kv0 * 2);
#pragma unroll
    for (int i = 0; i < K_LOADS; ++i)
      buf_dma16(rsK, Ks + (wave_u * 64 + i * 256) * 16, koff[i], ks_off);
#pragma unroll
    for (int i = 0; i < V_LOADS; ++i)
      buf_dma16(rsV, Vs + (wave_u * 64 + i * 256) * 16, voff[i], vs_off);
  };
  auto full_tile = [&](int t) { return t * KVB + KVB <= p.Sk; };

  // Software pipeline with ONE barrier per tile: at the top of iteration t every wave waits for its own share of tile t's DMA
  // and meets the others -- which also proves that all of them finished iteration t-1, so the buffer tile t+1 goes to (the one
  // iteration t-1 read) is free and its DMA is issued right after the barrier, a full tile of compute ahead of its use.
  // A ragged last tile (keys past Sk inside it) takes the register path, which zero-fills the missing V^T columns.
#ifdef AP_STAMP
  unsigned long long* stamps = (vb == AP_STAMP && tid == 0) ? reinterpret_cast<unsigned long long*>(p.lse) : nullptr;
#define AP_ST(t, k) do { if (stamps && (t) < 64) stamps[(t) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AP_ST(t, k) do {} while (0)
#endif
  if (n_tiles > 0 && full_tile(0)) dma_tile(0, lds, lds + KVB * KROW);
  // Per-lane fragment bases: the K chunk swizzle looks at row bits 0-3 (HD 128) / 1-3 (HD 64) and the V^T one at bits 1-3, which the
  // block offsets (32 tb rows of K, 32 d rows of V^T) leave alone, and (2 ks | hh) ^ sw == (2 ks) ^ (hh ^ sw): every fragment
  // address is tile + block offset + (base ^ constant).  The loop is unrolled by two so that the tile buffer is a compile-time
  // constant too: the whole address becomes one v_xor + an instruction offset (the per-fragment row / swizzle / buffer arithmetic
  // was ~125 of the ~330 VALU instructions of a tile).
  const int kfb = ql * KROW + ((hh ^ ((HD == 128) ? (ql & 15) : ((ql >> 1) & 7))) << 4);
  const int vfb = ql * 128 + ((hh ^ ((ql >> 1) & 7)) << 4);
  auto body = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    AP_ST(t, 0);
    const int kv0 = t * KVB;
    char* Ks = lds + BUF * TILEB;
    char* Vs = Ks + KVB * KROW;
    if (full_tile(t)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      AP_ST(t, 6);
#ifndef AP_NO_BAR
      __syncthreads();
#endif
    } else {                          // this tile's buffer was last read in iteration t-2, which the barrier of t-1 closed
      load_tile(kv0);
      write_tile(Ks, Vs);
      __syncthreads();
    }
    AP_ST(t, 5);
#ifndef AP_NO_DMA
    if (t + 1 < n_tiles && full_tile(t + 1)) dma_tile(kv0 + KVB, lds + (1 - BUF) * TILEB, lds + (1 - BUF) * TILEB + KVB * KROW);
#endif
    AP_ST(t, 1);
    // causal: a tile whose first key lies past this wave's LAST query row contributes nothing to the wave (the block walks the
    // tiles its last wave needs); the wave only keeps the block's barrier / DMA cadence and leaves the SIMD to its partner
    if (CAUSAL && kv0 > q0 + 31 + off) return;
    // ---- S^T = K . Q^T : two 32-row kv blocks ----
    f32x16 s[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[tb][r] = 0.f;
    // the two key blocks' accumulation chains alternate (back-to-back MFMAs on ONE accumulator issue at ~72 cycles, not 32)
#ifdef AP_FRAG_NEAR      // fragments read right in front of their MFMA (the round-1..3 form; A/B builds)
    constexpr bool FRAG_AHEAD = false;
#else
    constexpr bool FRAG_AHEAD = HD == 128;      // hd 64 (the ViT) measured 15 % slower with the reads ahead: 34.5-36.0 against 29.8-30.6 us
#endif
    if constexpr (!FRAG_AHEAD) {
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + tb * 32 * KROW + (kfb ^ (ks << 5)));
          s[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[tb], 0, 0, 0);
        }
    } else {
      // fragment reads TWO k-steps (4 MFMAs, >= 128 cycles) ahead of their MFMAs: hipcc reuses one register pair and issues every read
      // right in front of its consumer (ds_read, s_waitcnt lgkmcnt, v_mfma: the LDS latency of every pair is exposed)
      bf16x8 kr[3][2];
#pragma unroll
      for (int pre = 0; pre < 2; ++pre)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) kr[pre][tb] = *reinterpret_cast<const bf16x8*>(Ks + tb * 32 * KROW + (kfb ^ (pre << 5)));
      asm volatile("" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        if (ks + 2 < HD / 16) {
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) kr[(ks + 2) % 3][tb] = *reinterpret_cast<const bf16x8*>(Ks + tb * 32 * KROW + (kfb ^ ((ks + 2) << 5)));
          asm volatile("" : "+v"(qf[ks]) :: "memory");      // the k-step's MFMAs (they read qf[ks]) stay BEHIND the reads issued above
        }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
#ifdef AP_NO_QK
          if (ks == 0)
#endif
          s[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[ks % 3][tb], qf[ks], s[tb], 0, 0, 0);
        }
      }
    }
    AP_ST(t, 2);
    // ---- mask + online softmax (lane owns query column ql; kv = 32tb + (r&3)+8(r>>2)+4hh) ----
#ifdef AP_NO_SM
    bf16x8 pf[2][2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) pf[tb][r >> 3][r & 7] = f2bf(s[tb][r]);
    const bool resc = false;
#else
    const int qlim = CAUSAL ? (qrow + off) : 0x7fffffff;
    // interior tiles (every key of the tile visible to every query row of this wave) skip the 32 compare/selects
    const bool need_mask = (kv0 + KVB > p.Sk) || (CAUSAL && kv0 + KVB - 1 > q0 + off);
    float mx = -INFINITY;
    if (need_mask) {
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool ok = (kv < p.Sk) && (kv <= qlim);
          s[tb][r] = ok ? s[tb][r] : -INFINITY;
        }
    }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[tb][r]);
    // lane <-> lane + 32 by v_permlane32_swap (VALU) instead of __shfl_xor (ds_bpermute_b32: an LDS-pipe round trip + lgkmcnt(0) in the
    // max -> reference -> exponentials chain of every tile); asm form, see xchg32
    if (p.lazy_rescale & 2) mx = fmaxf(mx, __shfl_xor(mx, 32, 64));      // (A3V_ATTN_LAZY=3: the ds_bpermute form, A/B runs)
    else { float x0, x1; xchg32(mx, x0, x1); mx = fmaxf(x0, x1); }
    const float m_new = fmaxf(m_run, mx);
    // Lazy rescale: m_run is the REFERENCE the exponentials are taken against, not necessarily the running maximum.  It only moves
    // when some row of the wave would otherwise see exp2 arguments above +LAZY (2^8: P <= 256 in bf16, the sums in fp32) -- after the
    // first tiles of a row that is rare, and the 64 multiplies of the O rescale + the l update (a quarter of the tile's VALU work,
    // which is what bounds this kernel) are skipped by a wave-uniform branch.  O / l and the LSE are unchanged in exact arithmetic.
    const bool grow = (p.lazy_rescale & 1) ? ((m_new - m_run) * p.scale_log2 > 8.f || m_run == -INFINITY) : true;
    const bool resc = __builtin_amdgcn_ballot_w64(grow && m_new != m_run) != 0;
    // rows past Sq (clamped duplicates) and fully-masked tiles keep m finite once any tile was seen
    const float m_tgt = resc ? m_new : m_run;
    const float m_use = (m_tgt == -INFINITY) ? 0.f : m_tgt;
    // raw v_exp_f32 (the arguments are <= 0 [<= LAZY] and results below the denormal range flush to 0, which is what a probability
    // that small should do; exp2f() wraps every call in a range fix-up: +4 VALU per element)
    float alpha = 1.f;
    if (resc) alpha = __builtin_amdgcn_exp2f((m_run - m_use) * p.scale_log2);
    m_run = m_tgt;
    float lsum = 0.f;
    const float mb = m_use * p.scale_log2;
    bf16x8 pf[2][2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef AP_NO_EXP
        const float pv = fmaf(s[tb][r], p.scale_log2, -mb);
#else
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[tb][r], p.scale_log2, -mb));
#endif
        lsum += pv;
        pf[tb][r >> 3][r & 7] = f2bf(pv);
      }
    l_run = resc ? l_run * alpha + lsum : l_run + lsum;
    if constexpr (PSWAP) {
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          u32x4 w;
          __builtin_memcpy(&w, &pf[tb][c], 16);           // dwords: keys 16c + 4hh + {0,1 | 2,3} and 16c + 8 + 4hh + {0,1 | 2,3}
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(w[e], w[2 + e], false, false);
            w[e] = sw[0];                                 // lanes < 32: own first quarter   | lanes >= 32: partner's third quarter
            w[2 + e] = sw[1];                             // lanes < 32: partner's 2nd quarter | lanes >= 32: own fourth quarter
          }
          __builtin_memcpy(&pf[tb][c], &w, 16);           // lanes < 32: keys 16c + 0..7, lanes >= 32: keys 16c + 8..15
        }
    }
    if (resc) {
#pragma unroll
      for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
#endif
    AP_ST(t, 3);
    // ---- O^T += V^T . P^T ----
    if constexpr (PSWAP && FRAG_AHEAD) {
      // the four V^T fragments of the next (key block, 16-key chunk) group are read while the current group's four MFMAs run
      constexpr int ND = HD / 32;
      bf16x8 vr[2][ND];
#pragma unroll
      for (int d = 0; d < ND; ++d) vr[0][d] = *reinterpret_cast<const bf16x8*>(Vs + d * 32 * 128 + (vfb ^ (0 << 4)));
      asm volatile("" ::: "memory");
#pragma unroll
      for (int gI = 0; gI < 4; ++gI) {
        const int tb = gI >> 1, c = gI & 1;
        if (gI + 1 < 4) {
          const int c16n = 4 * ((gI + 1) >> 1) + 2 * ((gI + 1) & 1);
#pragma unroll
          for (int d = 0; d < ND; ++d) vr[(gI + 1) & 1][d] = *reinterpret_cast<const bf16x8*>(Vs + d * 32 * 128 + (vfb ^ (c16n << 4)));
          asm volatile("" : "+v"(pf[tb][c]) :: "memory");   // the group's MFMAs (they read pf[tb][c]) stay BEHIND the reads issued above
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) {
#ifdef AP_NO_PV
          if (gI == 0)
#endif
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[gI & 1][d], pf[tb][c], o[d], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < HD / 32; ++d) {   // (the d blocks' chains interleaved, see the QK product)
          const int drow = d * 32 + ql;
          const int g = (drow >> 1) & 7;
          const char* vp = Vs + drow * 128 + hh * 8;
          const int c16 = 4 * tb + 2 * c;      // 16-B chunk of kv columns {0..7}; lane half hh takes its 8-B half (k = 4hh..4hh+3)
          bf16x8 vf;
          if constexpr (PSWAP) {               // lane half hh takes the WHOLE chunk c16 + hh (8 consecutive keys)
            vf = *reinterpret_cast<const bf16x8*>(Vs + d * 32 * 128 + (vfb ^ (c16 << 4)));
          } else {
            const u32x2 a0 = *reinterpret_cast<const u32x2*>(vp + ((c16 ^ g) << 4));
            const u32x2 a1 = *reinterpret_cast<const u32x2*>(vp + (((c16 + 1) ^ g) << 4));
            const u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
            __builtin_memcpy(&vf, &av, 16);
          }
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[tb][c], o[d], 0, 0, 0);
        }
    }
    AP_ST(t, 4);
  };
  {
    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
      body(t, std::integral_constant<int, 0>{});
      body(t + 1, std::integral_constant<int, 1>{});
    }
    if (t < n_tiles) body(t, std::integral_constant<int, 0>{});
  }

#ifdef AP_STAMP
  if (stamps) return;
#endif
  // ---- normalise and store O[q][d], d = 32*db + 8*g + 4*hh + {0..3} ----
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
#ifndef AP_STAMP
  if (p.lse && qrow < p.Sq && hh == 0)
#else
  if (false)
#endif
    p.lse[((int64_t)b * p.H + h) * p.Sq + qrow] = m_run * p.scale + __logf(l_tot);
  if (p.staged_o && !(p.o_ss & 7) && !(p.o_sh & 7) && !(p.o_sb & 7) && !(reinterpret_cast<uintptr_t>(p.out) & 15)) {
    // The lane owns one query row: stored directly, an instruction writes 16 bytes into each of 32 rows (partial lines, ~6 B/clk/CU:
    // tools/ubench/stores.hip).  The wave's 32 x HD tile goes through a private LDS patch (the K / V^T buffers are free after one
    // more barrier) and leaves as whole rows, 16 bytes per lane.  8-byte slot s of row r sits at slot s ^ ((r & (HD/8 - 1)) << 1).
    constexpr int ROWB = HD * 2, NPAIR = HD / 8, RPI = 64 / NPAIR;          // row bytes, 16-byte pairs per row, rows per store instruction
    __syncthreads();
    char* patch = lds + wave * (32 * ROWB);
    char* wrow = patch + ql * ROWB;
    const int wx = (ql & (NPAIR - 1)) << 1;
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][g4 * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(wrow + (((d * 8 + g4 * 2 + hh) ^ wx) << 3)) = ov;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int pr = lane % NPAIR, rr = lane / NPAIR;
    bf16_t* Ob = (bf16_t*)p.out + b * p.o_sb + h * p.o_sh + pr * 8;
    const int qw = qt * 128 + wave * 32;
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int r = it * RPI + rr;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + r * ROWB + ((pr ^ (r & (NPAIR - 1))) << 4));
      if (qw + r < p.Sq) *reinterpret_cast<bf16x8*>(Ob + (int64_t)(qw + r) * p.o_ss) = v;
    }
    return;
  }
  if (qrow < p.Sq) {
    bf16_t* O = (bf16_t*)p.out + b * p.o_sb + (int64_t)qrow * p.o_ss + h * p.o_sh;
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][g4 * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(O + d * 32 + g4 * 8 + hh * 4) = ov;
      }
  }
}

#ifdef A3V_EXPERIMENTS   // round 6: built, bit-identical, NOT faster (profiles/r06_dropped.md, profiles/r06a_*): `make EXPERIMENTS=1`, A3V_ATTN_PERSIST=1
// ------------------------------------------------------------------------------------
// attn_prefill_persist_kernel (round 6): the 4-wave / 32-rows-per-wave kernel above as PERSISTENT blocks.
//   profiles/r05_dropped.md: with neither DMA nor arithmetic the launch above still takes 41 of 130 us at B 8 x S 1091 -- 2 304 short
//   blocks, each paying dispatch, lane constants, a Q fetch from HBM with nothing to hide under, and an O store with a barrier in
//   front of it.  Here 2 blocks per CU stay resident and WALK the (batch, head, query tile) units:
//   * eight unit queues (one per XCD = blockIdx & 7, contiguous ranges of the same heavy-first / tile-rank-major order the launch
//     above uses, so a head group's K / V^T stays in that XCD's L2); a block's first unit is static, the following ones come from ONE
//     returning atomicAdd per unit, issued at the START of the unit and read at its last tile (the dequeue latency is never waited
//     for); a block leaves when its queue is empty.  (First build: an exhausted queue sent the block on to the next seven -- every
//     block then ends with seven BLOCKING atomics on words all 512 blocks hit at the same moment: +42 us on the empty-loop build,
//     gpurun_out/r06a_skeleton_gate.txt.  The XCDs' shares are as even as the one-block-per-unit launch makes them.)
//   * unit seam: behind the last P V product the next unit's Q fragments are fetched into the (dead) qf registers, ONE barrier frees
//     both tile buffers, the next unit's K / V^T tile 0 goes to buffer 0 by LDS-DMA, and only then does this unit's O leave through
//     the LDS patch (buffer 1) as whole rows: Q fetch, tile-0 DMA and the O stores all run under each other and under the partner
//     block; the stores are never waited for except by the vmcnt(0) in front of tile 0, whose DMA was issued before them;
//   * the counters (8 queues + 1 exit count) are left zero by the last block to exit.
// Same arithmetic, same order of every fp32 sum as the kernel above: outputs are bit-identical (tests/test_gpu_kernels.py).
// ------------------------------------------------------------------------------------
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_prefill_persist_kernel(AttnArgs p, int* ctr, int n_units) {
  constexpr int KVB = 64;
  constexpr int KROW = HD * 2;
  constexpr int KCH = HD / 8;
  constexpr int K_LOADS = KVB * KCH / 256;
  constexpr int V_LOADS = HD * 8 / 256;
  constexpr int TILEB = KVB * KROW + HD * 128;
  __shared__ __attribute__((aligned(1024))) char lds[2 * TILEB];
  __shared__ int next_slot;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nqt = (p.Sq + 127) / 128;
  const int off = p.Sk - p.Sq;                 // right alignment (llama_ens5.py:181-185)
  const int ql = lane & 31, hh = lane >> 5;

  // ---- unit queues
  const int uq = n_units >> 3, ur = n_units & 7;
  auto q_start = [&](int x) { return x < ur ? x * (uq + 1) : ur * (uq + 1) + (x - ur) * uq; };
  auto q_len = [&](int x) { return uq + (x < ur ? 1 : 0); };
  auto q_blocks = [&](int x) { return ((int)gridDim.x - x + 7) >> 3; };     // blocks whose first (static) unit came from queue x
  const int cur_q = blockIdx.x & 7;
  auto decode = [&](int vb, int& b, int& h, int& qt) {
    int head_slot = vb / nqt;
    qt = nqt - 1 - (vb - head_slot * nqt);
    if (CAUSAL && p.head_group > 1) {
      const int G = p.head_group, per = G * nqt;
      const int grp = vb / per, r = vb - grp * per;
      head_slot = grp * G + r % G;
      qt = nqt - 1 - r / G;
    }
    b = head_slot / p.H;
    h = head_slot - b * p.H;
  };

  // ---- lane constants of the LDS-DMA image and the fragment reads (head-independent)
  unsigned koff[K_LOADS], voff[V_LOADS];
#pragma unroll
  for (int i = 0; i < K_LOADS; ++i) {
    const int id = tid + i * 256;
    const int row = id / KCH, slot = id % KCH;
    const int sw = (HD == 128) ? (row & 15) : ((row >> 1) & 7);
    koff[i] = (unsigned)((row * p.k_ss + (slot ^ sw) * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < V_LOADS; ++i) {
    const int id = tid + i * 256;
    const int d = id >> 3, slot = id & 7;
    voff[i] = (unsigned)((d * p.v_sd + (slot ^ ((d >> 1) & 7)) * 8) * 2);
  }
  const int kfb = ql * KROW + ((hh ^ ((HD == 128) ? (ql & 15) : ((ql >> 1) & 7))) << 4);
  const int vfb = ql * 128 + ((hh ^ ((ql >> 1) & 7)) << 4);
  auto full_tile = [&](int t) { return t * KVB + KVB <= p.Sk; };

  // the last block to leave puts the counters back to zero for the next launch
  auto leave = [&]() {
    if (tid == 0) {
      const int done = atomicAdd(&ctr[8], 1);
      if (done == (int)gridDim.x - 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) __hip_atomic_store(&ctr[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  int b, h, qt;
  {
    const int loc = blockIdx.x >> 3;
    if (loc >= q_len(cur_q)) { leave(); return; }            // (host: grid <= n_units, so every block has a first unit)
    decode(q_start(cur_q) + loc, b, h, qt);
  }
  bf16x8 qf[HD / 16];
  auto fetch_q = [&](int b_, int h_, int qt_) {
    const bf16_t* Q = (const bf16_t*)p.q + b_ * p.q_sb + h_ * p.q_sh;
    const int qr = qt_ * 128 + wave * 32 + ql;
    const int qrc = qr < p.Sq ? qr : p.Sq - 1;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks)
      qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qrc * p.q_ss + ks * 16 + hh * 8);
  };
  fetch_q(b, h, qt);
  const bf16_t* K = (const bf16_t*)p.k + b * p.k_sb + (h / (p.H / p.Hkv)) * p.k_sh;
  const bf16_t* VT = (const bf16_t*)p.vt + b * p.v_sb + (h / (p.H / p.Hkv)) * p.v_sh;
  auto dma_tile = [&](const bf16_t* Kb, const bf16_t* Vb, int kv0, char* Ks, char* Vs) {
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, 0x7fffffff, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, 0x7fffffff, 0x00020000);
    const unsigned ks_off = (unsigned)(kv0 * p.k_ss * 2), vs_off = (unsigned)(kv0 * 2);
#pragma unroll
    for (int i = 0; i < K_LOADS; ++i) buf_dma16(rsK, Ks + (wave_u * 64 + i * 256) * 16, koff[i], ks_off);
#pragma unroll
    for (int i = 0; i < V_LOADS; ++i) buf_dma16(rsV, Vs + (wave_u * 64 + i * 256) * 16, voff[i], vs_off);
  };
#ifndef AP_NO_DMA
  if (full_tile(0)) dma_tile(K, VT, 0, lds, lds + KVB * KROW);
#endif

#ifdef APP_STAMP
  unsigned long long* stamps = ((int)blockIdx.x == APP_STAMP && tid == 0) ? reinterpret_cast<unsigned long long*>(p.lse) : nullptr;
  int unit_no = 0;
  // every block: [start, end, units, tiles, XCC id] behind the per-unit records
  unsigned long long* span = tid == 0 ? reinterpret_cast<unsigned long long*>(p.lse) + 32 * 8 + blockIdx.x * 8 : nullptr;
  unsigned long long tiles_done = 0;
  if (span) { span[0] = __builtin_amdgcn_s_memrealtime(); span[4] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)); }
#define APP_ST(k) do { if (stamps && unit_no < 32) stamps[unit_no * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define APP_ST(k) do {} while (0)
#endif
  for (;;) {
    APP_ST(0);
#ifdef APP_STAMP
    const unsigned long long unit_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- the NEXT unit's queue position: requested now, looked at behind this unit's last tile
    int pending = 0;
    if (tid == 0) pending = atomicAdd(&ctr[cur_q], 1);
    const int q0 = qt * 128 + wave * 32;
    const int qrow = q0 + ql;
    f32x16 o[HD / 32];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    int kv_end = p.Sk;
    if (CAUSAL) {
      const int last_q = min(qt * 128 + 127, p.Sq - 1);
      kv_end = min(p.Sk, last_q + off + 1);
    }
    const int n_tiles = (kv_end + KVB - 1) / KVB;

    u32x4 kreg[K_LOADS], vreg[V_LOADS];
    auto load_tile = [&](int kv0) {
#pragma unroll
      for (int i = 0; i < K_LOADS; ++i) {
        const int id = tid + i * 256;
        const int row = id / KCH, ch = id % KCH;
        int kr = kv0 + row;
        kr = kr < p.Sk ? kr : p.Sk - 1;
        kreg[i] = *reinterpret_cast<const u32x4*>(K + (int64_t)kr * p.k_ss + ch * 8);
      }
#pragma unroll
      for (int i = 0; i < V_LOADS; ++i) {
        const int id = tid + i * 256;
        const int d = id >> 3, ch = id & 7;
        const int kv = kv0 + ch * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (kv + 8 <= p.Sk) {
          v = *reinterpret_cast<const u32x4*>(VT + (int64_t)d * p.v_sd + kv);
        } else if (kv < p.Sk) {   // ragged tail: zero the columns >= Sk (0 * garbage must stay 0)
          const unsigned short* src = reinterpret_cast<const unsigned short*>(VT + (int64_t)d * p.v_sd + kv);
          unsigned short e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = (kv + j < p.Sk) ? src[j] : (unsigned short)0;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (unsigned)e[2 * j] | ((unsigned)e[2 * j + 1] << 16);
        }
        vreg[i] = v;
      }
    };
    auto write_tile = [&](char* Ks, char* Vs) {
#pragma unroll
      for (int i = 0; i < K_LOADS; ++i) {
        const int id = tid + i * 256;
        const int row = id / KCH, ch = id % KCH;
        const int sw = (HD == 128) ? (row & 15) : ((row >> 1) & 7);
        *reinterpret_cast<u32x4*>(Ks + row * KROW + ((ch ^ sw) << 4)) = kreg[i];
      }
#pragma unroll
      for (int i = 0; i < V_LOADS; ++i) {
        const int id = tid + i * 256;
        const int d = id >> 3, ch = id & 7;
        *reinterpret_cast<u32x4*>(Vs + d * 128 + ((ch ^ ((d >> 1) & 7)) << 4)) = vreg[i];
      }
    };

    auto body = [&](int t, auto bufc) {
      constexpr int BUF = decltype(bufc)::value;
      const int kv0 = t * KVB;
      char* Ks = lds + BUF * TILEB;
      char* Vs = Ks + KVB * KROW;
      if (t == n_tiles - 1 && tid == 0) {
        // resolve the next unit (thread 0; the atomic was issued a whole unit ago) and publish it in front of this tile's barrier
        const int loc = q_blocks(cur_q) + pending;
        next_slot = loc < q_len(cur_q) ? q_start(cur_q) + loc : -1;
      }
      if (full_tile(t)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0) APP_ST(1);
#ifndef AP_NO_BAR
        __syncthreads();
#endif
        if (t == 0) APP_ST(2);
      } else {                          // this tile's buffer was last read in iteration t-2 (or freed by the unit seam)
        load_tile(kv0);
        write_tile(Ks, Vs);
        __syncthreads();
      }
#ifndef AP_NO_DMA
      if (t + 1 < n_tiles && full_tile(t + 1)) dma_tile(K, VT, kv0 + KVB, lds + (1 - BUF) * TILEB, lds + (1 - BUF) * TILEB + KVB * KROW);
#endif
      if (CAUSAL && kv0 > q0 + 31 + off) return;
      // ---- S^T = K . Q^T : two 32-row kv blocks ----
      f32x16 s[2];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[tb][r] = 0.f;
      constexpr bool FRAG_AHEAD = HD == 128;
      if constexpr (!FRAG_AHEAD) {
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + tb * 32 * KROW + (kfb ^ (ks << 5)));
            s[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[tb], 0, 0, 0);
          }
      } else {
        bf16x8 kr[3][2];
#pragma unroll
        for (int pre = 0; pre < 2; ++pre)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) kr[pre][tb] = *reinterpret_cast<const bf16x8*>(Ks + tb * 32 * KROW + (kfb ^ (pre << 5)));
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
          if (ks + 2 < HD / 16) {
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) kr[(ks + 2) % 3][tb] = *reinterpret_cast<const bf16x8*>(Ks + tb * 32 * KROW + (kfb ^ ((ks + 2) << 5)));
            asm volatile("" : "+v"(qf[ks]) :: "memory");
          }
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) {
#ifdef AP_NO_QK
            if (ks == 0)
#endif
            s[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[ks % 3][tb], qf[ks], s[tb], 0, 0, 0);
          }
        }
      }
      // ---- mask + online softmax (lane owns query column ql; kv = 32tb + (r&3)+8(r>>2)+4hh) ----
#ifdef AP_NO_SM
      bf16x8 pf[2][2];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[tb][r >> 3][r & 7] = f2bf(s[tb][r]);
#else
      const int qlim = CAUSAL ? (qrow + off) : 0x7fffffff;
      const bool need_mask = (kv0 + KVB > p.Sk) || (CAUSAL && kv0 + KVB - 1 > q0 + off);
      float mx = -INFINITY;
      if (need_mask) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const bool ok = (kv < p.Sk) && (kv <= qlim);
            s[tb][r] = ok ? s[tb][r] : -INFINITY;
          }
      }
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[tb][r]);
      if (p.lazy_rescale & 2) mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      else { float x0, x1; xchg32(mx, x0, x1); mx = fmaxf(x0, x1); }
      const float m_new = fmaxf(m_run, mx);
      const bool grow = (p.lazy_rescale & 1) ? ((m_new - m_run) * p.scale_log2 > 8.f || m_run == -INFINITY) : true;
      const bool resc = __builtin_amdgcn_ballot_w64(grow && m_new != m_run) != 0;
      const float m_tgt = resc ? m_new : m_run;
      const float m_use = (m_tgt == -INFINITY) ? 0.f : m_tgt;
      float alpha = 1.f;
      if (resc) alpha = __builtin_amdgcn_exp2f((m_run - m_use) * p.scale_log2);
      m_run = m_tgt;
      float lsum = 0.f;
      const float mb = m_use * p.scale_log2;
      bf16x8 pf[2][2];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[tb][r], p.scale_log2, -mb));
          lsum += pv;
          pf[tb][r >> 3][r & 7] = f2bf(pv);
        }
      l_run = resc ? l_run * alpha + lsum : l_run + lsum;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          u32x4 w;
          __builtin_memcpy(&w, &pf[tb][c], 16);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(w[e], w[2 + e], false, false);
            w[e] = sw[0];
            w[2 + e] = sw[1];
          }
          __builtin_memcpy(&pf[tb][c], &w, 16);
        }
      if (resc) {
#pragma unroll
        for (int d = 0; d < HD / 32; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      }
#endif
      // ---- O^T += V^T . P^T ----
      if constexpr (FRAG_AHEAD) {
        constexpr int ND = HD / 32;
        bf16x8 vr[2][ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) vr[0][d] = *reinterpret_cast<const bf16x8*>(Vs + d * 32 * 128 + (vfb ^ (0 << 4)));
        asm volatile("" ::: "memory");
#pragma unroll
        for (int gI = 0; gI < 4; ++gI) {
          const int tb = gI >> 1, c = gI & 1;
          if (gI + 1 < 4) {
            const int c16n = 4 * ((gI + 1) >> 1) + 2 * ((gI + 1) & 1);
#pragma unroll
            for (int d = 0; d < ND; ++d) vr[(gI + 1) & 1][d] = *reinterpret_cast<const bf16x8*>(Vs + d * 32 * 128 + (vfb ^ (c16n << 4)));
            asm volatile("" : "+v"(pf[tb][c]) :: "memory");
          }
#pragma unroll
          for (int d = 0; d < ND; ++d) {
#ifdef AP_NO_PV
            if (gI == 0)
#endif
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[gI & 1][d], pf[tb][c], o[d], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < HD / 32; ++d) {
              const int c16 = 4 * tb + 2 * c;
              const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + d * 32 * 128 + (vfb ^ (c16 << 4)));
              o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[tb][c], o[d], 0, 0, 0);
            }
      }
    };
    {
      int t = 0;
      for (; t + 1 < n_tiles; t += 2) {
        body(t, std::integral_constant<int, 0>{});
        body(t + 1, std::integral_constant<int, 1>{});
      }
      if (t < n_tiles) body(t, std::integral_constant<int, 0>{});
    }

    // ---- unit seam
    APP_ST(3);
    const int vb_next = __builtin_amdgcn_readfirstlane(next_slot);   // published in front of the last tile's barrier
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
#ifndef APP_STAMP
    if (p.lse && qrow < p.Sq && hh == 0) p.lse[((int64_t)b * p.H + h) * p.Sq + qrow] = m_run * p.scale + __logf(l_tot);
#endif
    int nb = 0, nh = 0, nqt_ = 0;
    if (vb_next >= 0) {
      decode(vb_next, nb, nh, nqt_);
      fetch_q(nb, nh, nqt_);                    // qf is dead behind the last tile's K Q^T
    }
    const bool staged = p.staged_o && !(p.o_ss & 7) && !(p.o_sh & 7) && !(p.o_sb & 7) && !(reinterpret_cast<uintptr_t>(p.out) & 15);
    __syncthreads();                            // every wave is past its last tile: both tile buffers are free
    APP_ST(4);
    const bf16_t* Kn = K;
    const bf16_t* VTn = VT;
    if (vb_next >= 0) {
      Kn = (const bf16_t*)p.k + nb * p.k_sb + (nh / (p.H / p.Hkv)) * p.k_sh;
      VTn = (const bf16_t*)p.vt + nb * p.v_sb + (nh / (p.H / p.Hkv)) * p.v_sh;
#ifndef AP_NO_DMA
      if (full_tile(0)) dma_tile(Kn, VTn, 0, lds, lds + KVB * KROW);
#endif
    }
    APP_ST(5);
    if (staged) {
      // the wave's 32 x HD tile leaves through a private LDS patch in tile buffer 1 as whole rows (see the kernel above); buffer 1 is
      // next written by the DMA of the next unit's tile 1, issued behind that unit's first barrier, i.e. after every wave's reads below
      constexpr int ROWB = HD * 2, NPAIR = HD / 8, RPI = 64 / NPAIR;
      char* patch = lds + TILEB + wave * (32 * ROWB);
      char* wrow = patch + ql * ROWB;
      const int wx = (ql & (NPAIR - 1)) << 1;
#pragma unroll
      for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][g4 * 4 + e] * inv);
          *reinterpret_cast<bf16x4*>(wrow + (((d * 8 + g4 * 2 + hh) ^ wx) << 3)) = ov;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int pr = lane % NPAIR, rr = lane / NPAIR;
      bf16_t* Ob = (bf16_t*)p.out + b * p.o_sb + h * p.o_sh + pr * 8;
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int r = it * RPI + rr;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + r * ROWB + ((pr ^ (r & (NPAIR - 1))) << 4));
        if (q0 + r < p.Sq) *reinterpret_cast<bf16x8*>(Ob + (int64_t)(q0 + r) * p.o_ss) = v;
      }
    } else if (qrow < p.Sq) {
      bf16_t* O = (bf16_t*)p.out + b * p.o_sb + (int64_t)qrow * p.o_ss + h * p.o_sh;
#pragma unroll
      for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][g4 * 4 + e] * inv);
          *reinterpret_cast<bf16x4*>(O + d * 32 + g4 * 8 + hh * 4) = ov;
        }
    }
    APP_ST(6);
#ifdef APP_STAMP
    if (stamps && unit_no < 32) stamps[unit_no * 8 + 7] = (unsigned long long)n_tiles;
    ++unit_no;
    tiles_done += n_tiles;
    if (span) { span[1] = __builtin_amdgcn_s_memrealtime(); span[2] = unit_no; span[3] = tiles_done; span[5] = n_tiles; span[6] = unit_t0; }
#endif
    if (vb_next < 0) break;
    b = nb; h = nh; qt = nqt_; K = Kn; VT = VTn;
  }
  leave();
}

#endif  // A3V_EXPERIMENTS (attn_prefill_persist_kernel)

#ifdef A3V_EXPERIMENTS   // measured and not dispatched (profiles/r04g_attn_w64_experiment.txt): built only with `make EXPERIMENTS=1`, A3V_ATTN_W64=1
// ------------------------------------------------------------------------------------
// attn_prefill_w64_kernel (round 4): hd = 128, ONE wave per SIMD, 64 query rows per wave.
//   * the 32-row wave of the kernel above reads 1 KiB of LDS per MFMA (K / V^T fragments): at the MFMA rate that is the LDS
//     bandwidth of the CU.  Here a wave owns TWO 32-row query blocks (A, B) and every K / V^T fragment feeds two MFMAs.
//   * block = 128 query rows = 4 waves = (query half qw) x (key half kw): a step stages 128 keys, wave (qw, kw) takes the 64-key
//     sub-tile kw of every step for its 64 rows -- the two waves of a query half do the same amount of work under the causal
//     mask (the diagonal costs one sub-tile slot per block), and merge (O, m, l) through LDS at the end (flash-decoding inside
//     the block).  Each wave finalises ONE of the two query blocks of its half.
//   * software pipeline over 32-key UNITS u: phase(u) = { O += V^T(u-1) P(u-1) ; S(u+1) = K(u+1) Q^T } (32 MFMAs) beside the
//     softmax algebra of S(u): one element (fma, exp, sum, max, cvt, lane swap) per MFMA, placed by hand between the MFMA
//     statements.  The exponentials are taken SPECULATIVELY against the current reference of the row while the unit's maximum
//     is reduced beside them; only when some row of the wave outgrows the lazy-rescale bound (2^8) is the unit redone after a
//     rescale (wave-uniform, rare after a row's first unit): no decision sits between the maxima and the exponentials.
//   * register file owned by hand (the idiom of attn_bwd_dkv2_kernel): a[0:63] O^T of block A, a[64:127] of block B,
//     a[128:159] / a[160:191] the Q fragments of A / B; every MFMA is an asm statement on those literals; the scores land in
//     VGPRs (the VALU reads every element), P^T operands are VGPRs.  hipcc's own allocation of the builtin form moved the
//     accumulators between the two files around every branch (2000+ v_accvgpr moves, 1.6 KB of scratch).
//   * K sub-tiles run one step ahead of V^T: at the single barrier of step s, K(s+2) and V^T(s+1) are issued by LDS-DMA (asm
//     form: hipcc would put vmcnt(0) in front of every LDS read behind a builtin DMA) into the stages phase 1 of step s released.
// Same values as attn_prefill_bf16_kernel up to the order of the fp32 sums (tests compare both with the oracle).
// ------------------------------------------------------------------------------------
#define W64_CL8(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
#define W64_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", W64_CL8(1), W64_CL8(2), W64_CL8(3), W64_CL8(4), W64_CL8(5), W64_CL8(6), \
    W64_CL8(7), W64_CL8(8), W64_CL8(9), W64_CL8(10), W64_CL8(11), W64_CL8(12), W64_CL8(13), W64_CL8(14), W64_CL8(15), W64_CL8(16), W64_CL8(17), \
    W64_CL8(18), "a190", "a191"
typedef __attribute__((ext_vector_type(4))) int w64_i32x4;
template <int I, int N, typename F>
__device__ __forceinline__ void w64_sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); w64_sfor<I + 1, N>(f); }
}
// a[ACC:ACC+15] += A (VGPRs) x B (VGPRs)
template <int ACC>
__device__ __forceinline__ void w64_mfma_o(const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(ACC), "i"(ACC + 15) : W64_AGPRS);
}
// VGPR accumulator (ZERO ? = : +=) A (VGPRs) x a[BR:BR+3]
template <int BR, bool ZERO>
__device__ __forceinline__ void w64_mfma_s(f32x16& acc, const bf16x8& a) {
  if constexpr (ZERO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(a), "i"(BR), "i"(BR + 3) : W64_AGPRS);
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "i"(BR), "i"(BR + 3) : W64_AGPRS);
}
template <int R>
__device__ __forceinline__ float w64_agpr_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(R) : W64_AGPRS);
  return v;
}
template <int R>
__device__ __forceinline__ void w64_agpr_write(unsigned v) {
  asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "i"(R) : W64_AGPRS);
}
template <int R>
__device__ __forceinline__ void w64_agpr_scale(float f) {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a[%c2], %0" : "=&v"(t) : "v"(f), "i"(R) : W64_AGPRS);
}
__device__ __forceinline__ void w64_dma16(w64_i32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ float w64_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float w64_max(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ unsigned w64_cvt_pk(float lo, float hi) {
  unsigned d;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
  return d;
}

// own value and the value of lane ^ 32 in (a, b) or (b, a).  Asm form: with both operands of the builtin holding the same value hipcc
// uses ONE of the two results for both (v_max v, v, v in the ISA), i.e. the exchange silently disappears.
__device__ __forceinline__ void w64_xchg32(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_prefill_w64_kernel(AttnArgs p) {
  constexpr int HD = 128, KROW = 256;
  constexpr int SUB = 16384;          // one 64-key sub-tile: K 64 rows x 256 B, V^T 128 d-rows x 128 B
  constexpr int STAGE = 2 * SUB;      // a step's two sub-tiles (key halves)
  constexpr int RA = 0, RB = 64, QA = 128, QB = 160;     // owned AGPR ranges
  extern __shared__ __attribute__((aligned(1024))) char w64_lds[];
  char* const Kst = w64_lds;                  // [2 stages][2 subs][16 KB]
  char* const Vst = w64_lds + 2 * STAGE;      // [2 stages][2 subs][16 KB]
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)w64_lds);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qw = wave >> 1, kw = wave & 1;
  const int nqt = (p.Sq + 127) / 128;
  int vb;
  {   // XCD-aware block order, as in attn_prefill_bf16_kernel
    const int total = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, q = total >> 3, r = total & 7;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  int head_slot = vb / nqt;
  int qt = nqt - 1 - (vb - head_slot * nqt);
  if (CAUSAL && p.head_group > 1) {
    const int G = p.head_group, per = G * nqt;
    const int grp = vb / per, r = vb - grp * per;
    head_slot = grp * G + r % G;
    qt = nqt - 1 - r / G;
  }
  const int b = head_slot / p.H, h = head_slot - b * p.H;
  const int hk = h / (p.H / p.Hkv);
  const int q0 = qt * 128 + qw * 64;
  const int off = p.Sk - p.Sq;

  const bf16_t* Q = (const bf16_t*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16_t* K = (const bf16_t*)p.k + b * p.k_sb + hk * p.k_sh;
  const bf16_t* VT = (const bf16_t*)p.vt + b * p.v_sb + hk * p.v_sh;

  const int ql = lane & 31, hh = lane >> 5;
  const int qrA = q0 + ql, qrB = q0 + 32 + ql;

  // key range of the block / of this wave's 64 rows
  int kv_end_blk = p.Sk, kv_end_w = p.Sk;
  if (CAUSAL) {
    kv_end_blk = min(p.Sk, min(qt * 128 + 127, p.Sq - 1) + off + 1);
    kv_end_w = min(p.Sk, min(q0 + 63, p.Sq - 1) + off + 1);
  }
  if (q0 >= p.Sq) kv_end_w = 0;
#ifdef W64_NO_LOOP      // timing experiment (wrong results): the per-block cost alone
  kv_end_blk = 0; kv_end_w = 0;
#endif
  const int n_steps = __builtin_amdgcn_readfirstlane((kv_end_blk + 127) / 128);
  const int n_my = __builtin_amdgcn_readfirstlane(kv_end_w > 64 * kw ? (kv_end_w - 64 * kw + 127) / 128 : 0);

#ifdef W64_STAMP        // cycle stamps of one wave (block W64_STAMP, wave 0) into the lse buffer: tools/w64_stamps.py
  unsigned long long* stamps = (vb == W64_STAMP && tid == 0) ? reinterpret_cast<unsigned long long*>(p.lse + (int64_t)p.B * p.H * p.Sq) : nullptr;   // behind the lse rows (the tool allocates the room)
  int stamp_i = 0;
#define W64_ST() do { if (stamps && stamp_i < 512) { stamps[stamp_i++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define W64_ST() do {} while (0)
#endif
  // ---- staging: LDS-DMA for sub-tiles whose 64 keys exist, a register path (zero-filled V^T columns) for a ragged one ----
  w64_i32x4 rsK, rsV;
  {
    const uint64_t ka = (uint64_t)K, va = (uint64_t)VT;
    rsK[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ka); rsK[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ka >> 32) & 0xffff);
    rsK[2] = 0x7fffffff; rsK[3] = 0x00020000;
    rsV[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)va); rsV[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(va >> 32) & 0xffff);
    rsV[2] = 0x7fffffff; rsV[3] = 0x00020000;
  }
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + i * 256;
    const int row = id >> 4, slot = id & 15;
    koff[i] = (unsigned)((row * p.k_ss + (slot ^ (row & 15)) * 8) * 2);
    const int d = id >> 3, vs = id & 7;
    voff[i] = (unsigned)((d * p.v_sd + (vs ^ ((d >> 1) & 7)) * 8) * 2);
  }
  auto fetchK = [&](int s, int stg) __attribute__((always_inline)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kv0 = s * 128 + sub * 64;
      if (kv0 >= kv_end_blk) continue;
      const int ko = stg * STAGE + sub * SUB;
      if (kv0 + 64 <= p.Sk) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(kv0 * p.k_ss * 2));
#pragma unroll
        for (int i = 0; i < 4; ++i) w64_dma16(rsK, lds0 + ko + (wave * 64 + i * 256) * 16, koff[i], so);
      } else {
        char* Ks = Kst + ko;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int id = tid + i * 256;
          const int row = id >> 4, ch = id & 15;
          int kr = kv0 + row;
          kr = kr < p.Sk ? kr : p.Sk - 1;
          const u32x4 v = *reinterpret_cast<const u32x4*>(K + (int64_t)kr * p.k_ss + ch * 8);
          *reinterpret_cast<u32x4*>(Ks + row * KROW + ((ch ^ (row & 15)) << 4)) = v;
        }
      }
    }
  };
  auto fetchV = [&](int s, int stg) __attribute__((always_inline)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kv0 = s * 128 + sub * 64;
      if (kv0 >= kv_end_blk) continue;
      const int vo = 2 * STAGE + stg * STAGE + sub * SUB;
      if (kv0 + 64 <= p.Sk) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(kv0 * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) w64_dma16(rsV, lds0 + vo + (wave * 64 + i * 256) * 16, voff[i], so);
      } else {
        char* Vs = w64_lds + vo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int id = tid + i * 256;
          const int d = id >> 3, ch = id & 7;
          const int kv = kv0 + ch * 8;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (kv + 8 <= p.Sk) {
            v = *reinterpret_cast<const u32x4*>(VT + (int64_t)d * p.v_sd + kv);
          } else if (kv < p.Sk) {
            const unsigned short* src = reinterpret_cast<const unsigned short*>(VT + (int64_t)d * p.v_sd + kv);
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (kv + j < p.Sk) ? src[j] : (unsigned short)0;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (unsigned)e[2 * j] | ((unsigned)e[2 * j + 1] << 16);
          }
          *reinterpret_cast<u32x4*>(Vs + d * 128 + ((ch ^ ((d >> 1) & 7)) << 4)) = v;
        }
      }
    }
  };

  W64_ST();
  asm volatile("s_nop 4" ::: W64_AGPRS);      // descriptor words fresh from v_readfirstlane
  if (n_steps > 0) { fetchK(0, 0); fetchV(0, 0); }
  if (n_steps > 1) fetchK(1, 1);

  // Q fragments (B operand of S^T = K Q^T): Q[row][16 ks + 8 hh + e] -> a[QA + 4 ks ..], a[QB + 4 ks ..]; O^T = 0
  {
    const int ra = qrA < p.Sq ? qrA : p.Sq - 1, rb = qrB < p.Sq ? qrB : p.Sq - 1;
    u32x4 ta[HD / 16], tb[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      ta[ks] = *reinterpret_cast<const u32x4*>(Q + (int64_t)ra * p.q_ss + ks * 16 + hh * 8);
      tb[ks] = *reinterpret_cast<const u32x4*>(Q + (int64_t)rb * p.q_ss + ks * 16 + hh * 8);
    }
    w64_sfor<0, 128>([&](auto i) { w64_agpr_write<decltype(i)::value>(0u); });
    w64_sfor<0, HD / 16>([&](auto ks) {
      w64_sfor<0, 4>([&](auto j) {
        w64_agpr_write<QA + 4 * decltype(ks)::value + decltype(j)::value>(ta[decltype(ks)::value][decltype(j)::value]);
        w64_agpr_write<QB + 4 * decltype(ks)::value + decltype(j)::value>(tb[decltype(ks)::value][decltype(j)::value]);
      });
    });
  }

  float mA = -INFINITY, mB = -INFINITY, mbA = 0.f, mbB = 0.f, lA = 0.f, lB = 0.f;   // reference, reference * scale_log2 (0 while -inf), per-lane partial sums
  f32x16 s0A, s0B, s1A, s1B;               // scores of the even / odd unit in flight
  bf16x8 p0A[2], p0B[2], p1A[2], p1B[2];   // P^T operands of the even / odd unit
  const float c = p.scale_log2;
  const bool lazy = (p.lazy_rescale & 1) != 0;

  const int kfb = ql * KROW + ((hh ^ (ql & 15)) << 4);
  const int vfb = ql * 128 + ((hh ^ ((ql >> 1) & 7)) << 4);

  // causal / ragged mask of one unit (keys kb .. kb + 31); interior units skip it (wave-uniform)
  auto mask_unit = [&](f32x16& sA, f32x16& sB, int kb) __attribute__((always_inline)) {
    const bool need = (kb + 31 >= p.Sk) || (CAUSAL && kb + 31 > q0 + off);
    if (!need) return;
    const int limA = CAUSAL ? qrA + off : 0x7fffffff, limB = CAUSAL ? qrB + off : 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const bool in = kv < p.Sk;
      sA[r] = (in && kv <= limA) ? sA[r] : -INFINITY;
      sB[r] = (in && kv <= limB) ? sB[r] : -INFINITY;
    }
  };
  // the unit again after a rescale (rare): new reference, O^T and l scaled, exponentials / sums / P^T operand from scratch
  auto redo = [&](auto rc, const f32x16& s, float& m, float& mb, float& l, float mx, float (&t)[4], unsigned (&w)[8]) __attribute__((always_inline)) {
    constexpr int R0 = decltype(rc)::value;
    const float mn = fmaxf(m, mx);
    const float mu = (mn == -INFINITY) ? 0.f : mn;
    const float alpha = __builtin_amdgcn_exp2f((m - mu) * c);
    m = mn; mb = mu * c;
    l *= alpha;
    asm volatile("s_nop 1" ::: W64_AGPRS);
    w64_sfor<0, 64>([&](auto i) { w64_agpr_scale<R0 + decltype(i)::value>(alpha); });
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mb));
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = (pv[r] + pv[r + 4]) + (pv[r + 8] + pv[r + 12]);
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r] = w64_cvt_pk(pv[2 * r], pv[2 * r + 1]);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(w[4 * cc + e], w[4 * cc + 2 + e], false, false);
        w[4 * cc + e] = sw[0];
        w[4 * cc + 2 + e] = sw[1];
      }
  };

  // phase(u): PV: O += V^T(u-1) P(u-1) (Vs = that unit's sub-tile, TB = its 32-key half); QK: S(u+1) = K(u+1) Q^T (Ks likewise);
  // softmax of the current unit's scores (cA, cB) -> P operands (pcA, pcB).  MFMA order per fragment pair g: O_A, O_B (V^T fragment g),
  // S_A, S_B (K fragment g); one softmax element after every MFMA (two when the phase has only one of the products).
  auto phase = [&](auto pvc, auto qkc, auto tbc, const char* Vs, const char* Ks, f32x16& cA, f32x16& cB, f32x16& nA, f32x16& nB,
                   bf16x8 (&pcA)[2], bf16x8 (&pcB)[2], const bf16x8 (&ppA)[2], const bf16x8 (&ppB)[2]) __attribute__((always_inline)) {
    constexpr bool PV = decltype(pvc)::value, QK = decltype(qkc)::value;
    constexpr int TB = decltype(tbc)::value;
    constexpr int SPM = (PV && QK) ? 1 : 2;       // softmax elements per MFMA
    float tA[4], tB[4], mxA = 0.f, mxB = 0.f, sprev = 0.f, cs = c;
    float ev[32];
    unsigned wA[8], wB[8];
    // slice(j): exponential of element j (and its place in the maximum chain), then the bookkeeping of element j - 1 (sum, bf16 pair,
    // lane swap) -- one element behind, so that no instruction waits on the transcendental in front of it
    auto tail = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value, r = j & 15;
      constexpr bool isB = j >= 16;
      float (&t)[4] = isB ? tB : tA;
      unsigned (&w)[8] = isB ? wB : wA;
      if constexpr (r < 4) t[r] = ev[j]; else t[r & 3] += ev[j];
      if constexpr (r & 1) w[r >> 1] = w64_cvt_pk(ev[j - 1], ev[j]);
      if constexpr ((r & 7) == 5 || (r & 7) == 7) {
        constexpr int cc = r >> 3, e2 = ((r & 7) == 7) ? 1 : 0;
        const auto sw = __builtin_amdgcn_permlane32_swap(w[4 * cc + e2], w[4 * cc + 2 + e2], false, false);
        w[4 * cc + e2] = sw[0];
        w[4 * cc + 2 + e2] = sw[1];
        asm volatile("" : "+v"(w[4 * cc + e2]), "+v"(w[4 * cc + 2 + e2]));
      } else if constexpr (r & 1) {
        asm volatile("" : "+v"(w[r >> 1]));
      }
      asm volatile("" : "+v"(t[r & 3]));
    };
    auto slice = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value, r = j & 15;
      constexpr bool isB = j >= 16;
      f32x16& s = isB ? cB : cA;
      float& mx = isB ? mxB : mxA;
      asm volatile("" : "+s"(cs));           // a new name for the scale per slice: the slice's arithmetic cannot move above this point
      const float sv = s[r];
      ev[j] = __builtin_amdgcn_exp2f(fmaf(sv, cs, -(isB ? mbB : mbA)));
      if constexpr (r == 0) mx = sv;
      else if constexpr (r == 15) mx = w64_max(mx, sv);
      else if constexpr ((r & 1) == 0) mx = w64_max3(mx, sprev, sv);
      else sprev = sv;
      asm volatile("" : "+v"(ev[j]), "+v"(mx));
      if constexpr (j > 0) tail(std::integral_constant<int, j - 1>{});
    };
#ifdef W64_NO_SM        // timing experiment (wrong results): MFMAs and fragment reads only
    auto slice_x = [&](auto jc) __attribute__((always_inline)) {};
#define slice slice_x
#endif
    auto vfrag = [&](int g) __attribute__((always_inline)) {
      return *reinterpret_cast<const bf16x8*>(Vs + (g & 3) * 4096 + (vfb ^ ((4 * TB + 2 * (g >> 2)) << 4)));
    };
    auto kfrag = [&](int g) __attribute__((always_inline)) { return *reinterpret_cast<const bf16x8*>(Ks + TB * 8192 + (kfb ^ (g << 5))); };
    bf16x8 vf[2], kf[2];
    if constexpr (PV) vf[0] = vfrag(0);
    if constexpr (QK) kf[0] = kfrag(0);
    asm volatile("" ::: "memory");
    w64_sfor<0, 8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int NM = (PV ? 2 : 0) + (QK ? 2 : 0);          // MFMAs of this fragment pair
#ifndef W64_NO_FRAG     // timing experiment (wrong results): one fragment pair per phase
      if constexpr (g < 7) {
        if constexpr (PV) vf[(g + 1) & 1] = vfrag(g + 1);
        if constexpr (QK) kf[(g + 1) & 1] = kfrag(g + 1);
        asm volatile("" ::: "memory");
      }
#else
      if constexpr (g == 0) { if constexpr (PV) vf[1] = vf[0]; if constexpr (QK) kf[1] = kf[0]; }
#endif
      int k = 0;
      // S first, O last: the scores' last writers are two MFMAs away from the end of the phase (an XDL result needs ~19 wait states
      // before a VALU may read it; hipcc knows nothing about the asm statements)
      if constexpr (QK) {
        w64_mfma_s<QA + 4 * g, g == 0>(nA, kf[g & 1]);
        w64_sfor<0, SPM>([&](auto q) { slice(std::integral_constant<int, (g * NM + 0) * SPM + decltype(q)::value>{}); });
        w64_mfma_s<QB + 4 * g, g == 0>(nB, kf[g & 1]);
        w64_sfor<0, SPM>([&](auto q) { slice(std::integral_constant<int, (g * NM + 1) * SPM + decltype(q)::value>{}); });
      }
      if constexpr (PV) {
        constexpr int o = QK ? 2 : 0;
        w64_mfma_o<RA + 16 * (g & 3)>(vf[g & 1], ppA[g >> 2]);
        w64_sfor<0, SPM>([&](auto q) { slice(std::integral_constant<int, (g * NM + o) * SPM + decltype(q)::value>{}); });
        w64_mfma_o<RB + 16 * (g & 3)>(vf[g & 1], ppB[g >> 2]);
        w64_sfor<0, SPM>([&](auto q) { slice(std::integral_constant<int, (g * NM + o + 1) * SPM + decltype(q)::value>{}); });
      }
      (void)k;
#ifdef W64_STAMP_FINE
      if constexpr ((g & 1) == 1) W64_ST();
#endif
    });
#ifdef W64_NO_SM
#undef slice
    w64_sfor<0, 4>([&](auto i) { tA[decltype(i)::value] = 0.f; tB[decltype(i)::value] = 0.f; });
    w64_sfor<0, 8>([&](auto i) { wA[decltype(i)::value] = 0x3c003c00u; wB[decltype(i)::value] = 0x3c003c00u; });
#else
    tail(std::integral_constant<int, 31>{});
#endif
    if constexpr (!PV) asm volatile("s_nop 15\n\ts_nop 3" ::: W64_AGPRS);     // scores of a phase without O MFMAs -> their first VALU reader
    {
      float x0, x1;
      w64_xchg32(mxA, x0, x1); mxA = w64_max(x0, x1);
      w64_xchg32(mxB, x0, x1); mxB = w64_max(x0, x1);
    }
    const bool gA = lazy ? (mxA > -INFINITY && (mA == -INFINITY || (mxA - mA) * c > 8.f)) : (mxA > mA);
    const bool gB = lazy ? (mxB > -INFINITY && (mB == -INFINITY || (mxB - mB) * c > 8.f)) : (mxB > mB);
    if (__builtin_amdgcn_ballot_w64(gA || gB) != 0) {
      asm volatile("s_nop 15\n\ts_nop 15" ::: W64_AGPRS);       // the phase's last O MFMAs -> the AGPR reads of the rescale
      redo(std::integral_constant<int, RA>{}, cA, mA, mbA, lA, mxA, tA, wA);
      redo(std::integral_constant<int, RB>{}, cB, mB, mbB, lB, mxB, tB, wB);
      asm volatile("s_nop 4" ::: W64_AGPRS);
    }
#ifdef W64_STAMP_FINE
    W64_ST();
#endif
    lA += (tA[0] + tA[1]) + (tA[2] + tA[3]);
    lB += (tB[0] + tB[1]) + (tB[2] + tB[3]);
    __builtin_memcpy(&pcA[0], &wA[0], 16); __builtin_memcpy(&pcA[1], &wA[4], 16);
    __builtin_memcpy(&pcB[0], &wB[0], 16); __builtin_memcpy(&pcB[1], &wB[4], 16);
  };
  // O += V^T(u) P(u) of a wave's LAST unit (second half of sub-tile Vs)
  auto pv_tail = [&](const char* Vs, const bf16x8 (&ppA)[2], const bf16x8 (&ppB)[2]) __attribute__((always_inline)) {
    w64_sfor<0, 8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (g & 3) * 4096 + (vfb ^ ((4 + 2 * (g >> 2)) << 4)));
      w64_mfma_o<RA + 16 * (g & 3)>(vf, ppA[g >> 2]);
      w64_mfma_o<RB + 16 * (g & 3)>(vf, ppB[g >> 2]);
    });
  };
  using T_ = std::true_type; using F_ = std::false_type;
  using TB0 = std::integral_constant<int, 0>; using TB1 = std::integral_constant<int, 1>;

  W64_ST();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  W64_ST();
  if (n_my > 0) {   // S(0): unit 0 = step 0, first half of the wave's sub-tile
    const char* Ks = Kst + kw * SUB;
    w64_sfor<0, 8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kfb ^ (g << 5)));
      w64_mfma_s<QA + 4 * g, g == 0>(s0A, kf);
      w64_mfma_s<QB + 4 * g, g == 0>(s0B, kf);
    });
    asm volatile("s_nop 15\n\ts_nop 3" ::: W64_AGPRS);         // S(0) -> its first VALU reader
  }
  auto step = [&](int s, auto stc) __attribute__((always_inline)) {
    constexpr int SG = decltype(stc)::value;            // stage of step s (K(s), V^T(s)); the other one holds K(s+1) / V^T(s-1)
    const char* Kcur = Kst + SG * STAGE + kw * SUB;
    const char* Knxt = Kst + (1 - SG) * STAGE + kw * SUB;
    const char* Vcur = Vst + SG * STAGE + kw * SUB;
    const char* Vprv = Vst + (1 - SG) * STAGE + kw * SUB;
    const int kb = s * 128 + kw * 64;
    W64_ST();
    // phase 1: unit 2s (scores s0) beside PV of unit 2s-1 (V^T(s-1), second half) and QK of unit 2s+1 (K(s), second half -> s1)
    if (s < n_my) {
      mask_unit(s0A, s0B, kb);
      if (s > 0) phase(T_{}, T_{}, TB1{}, Vprv, Kcur, s0A, s0B, s1A, s1B, p0A, p0B, p1A, p1B);
      else phase(F_{}, T_{}, TB1{}, Vprv, Kcur, s0A, s0B, s1A, s1B, p0A, p0B, p1A, p1B);
    } else if (s == n_my && n_my > 0) {
      pv_tail(Vprv, p1A, p1B);
    }
    W64_ST();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W64_ST();
    __syncthreads();
    W64_ST();
#ifndef W64_NO_DMA      // timing experiment (wrong results): no staging inside the loop
    if (s + 2 < n_steps) fetchK(s + 2, SG);
    if (s + 1 < n_steps) fetchV(s + 1, 1 - SG);
#endif
    W64_ST();
    // phase 2: unit 2s+1 (scores s1) beside PV of unit 2s (V^T(s), first half) and QK of unit 2s+2 (K(s+1), first half -> s0)
    if (s < n_my) {
      mask_unit(s1A, s1B, kb + 32);
      if (s + 1 < n_my) phase(T_{}, T_{}, TB0{}, Vcur, Knxt, s1A, s1B, s0A, s0B, p1A, p1B, p0A, p0B);
      else phase(T_{}, F_{}, TB0{}, Vcur, Knxt, s1A, s1B, s0A, s0B, p1A, p1B, p0A, p0B);
    }
  };
  {
    int s = 0;
    for (; s + 1 < n_steps; s += 2) {
      step(s, std::integral_constant<int, 0>{});
      step(s + 1, std::integral_constant<int, 1>{});
    }
    if (s < n_steps) step(s, std::integral_constant<int, 0>{});
  }
  if (n_my == n_steps && n_my > 0) pv_tail(Vst + ((n_steps - 1) & 1) * STAGE + kw * SUB, p1A, p1B);
  asm volatile("s_nop 15\n\ts_nop 15" ::: W64_AGPRS);           // the last MFMAs -> the AGPR reads below

  W64_ST();
#ifdef W64_STAMP
  if (stamps) return;
#endif
  // ---- merge the two key halves: wave (qw, kw) finalises query block kw of its half and hands the other block's partial over ----
  __syncthreads();
  constexpr int XSZ = 66 * 64 * 4;                     // 64 accumulator values + reference + partial sum per lane
  float* xo = reinterpret_cast<float*>(w64_lds + wave * XSZ) + lane;
  const float* xi = reinterpret_cast<const float*>(w64_lds + (wave ^ 1) * XSZ) + lane;
  auto send = [&](auto rc, float m, float l) __attribute__((always_inline)) {
    w64_sfor<0, 64>([&](auto i) { xo[decltype(i)::value * 64] = w64_agpr_read<decltype(rc)::value + decltype(i)::value>(); });
    xo[64 * 64] = m;
    xo[65 * 64] = l;
  };
  auto finish = [&](auto rc, float m, float l, int qrow, int qbase) __attribute__((always_inline)) {
    constexpr int R0 = decltype(rc)::value;
    const float m2 = xi[64 * 64], l2 = xi[65 * 64];
    const float mn = fmaxf(m, m2);
    const float mu = (mn == -INFINITY) ? 0.f : mn;
    const float a1 = __builtin_amdgcn_exp2f((m - mu) * c), a2 = __builtin_amdgcn_exp2f((m2 - mu) * c);
    const float lp = l * a1 + l2 * a2;
    float x0, x1;
    w64_xchg32(lp, x0, x1);
    const float l_tot = x0 + x1;
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (p.lse && qrow < p.Sq && hh == 0) p.lse[((int64_t)b * p.H + h) * p.Sq + qrow] = mn * p.scale + __logf(l_tot);
    const float f1 = a1 * inv, f2 = a2 * inv;
    // whole-row stores through a private LDS patch (see attn_prefill_bf16_kernel)
    constexpr int ROWB = HD * 2, NPAIR = HD / 8, RPI = 64 / NPAIR;
    char* patch = w64_lds + 4 * XSZ + 1024 + wave * (32 * ROWB);
    char* wrow = patch + ql * ROWB;
    const int wx = (ql & (NPAIR - 1)) << 1;
    w64_sfor<0, 16>([&](auto qc) {                     // 4 accumulator values -> one 8-byte slot
      constexpr int d = decltype(qc)::value >> 2, g4 = decltype(qc)::value & 3;
      bf16x4 ov;
      w64_sfor<0, 4>([&](auto ec) {
        constexpr int e = decltype(ec)::value, i = d * 16 + g4 * 4 + e;
        ov[e] = f2bf(w64_agpr_read<R0 + i>() * f1 + xi[i * 64] * f2);
      });
      *reinterpret_cast<bf16x4*>(wrow + (((d * 8 + g4 * 2 + hh) ^ wx) << 3)) = ov;
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int pr = lane % NPAIR, rr = lane / NPAIR;
    bf16_t* Ob = (bf16_t*)p.out + b * p.o_sb + h * p.o_sh + pr * 8;
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int r = it * RPI + rr;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + r * ROWB + ((pr ^ (r & (NPAIR - 1))) << 4));
      if (qbase + r < p.Sq) *reinterpret_cast<bf16x8*>(Ob + (int64_t)(qbase + r) * p.o_ss) = v;
    }
  };
  if (kw == 0) send(std::integral_constant<int, RB>{}, mB, lB); else send(std::integral_constant<int, RA>{}, mA, lA);
  __syncthreads();
  if (kw == 0) finish(std::integral_constant<int, RA>{}, mA, lA, qrA, q0); else finish(std::integral_constant<int, RB>{}, mB, lB, qrB, q0 + 32);
}

#endif  // A3V_EXPERIMENTS (attn_prefill_w64_kernel)

#ifdef A3V_EXPERIMENTS
// ------------------------------------------------------------------------------------
// Ping-pong prefill kernel (hd = 128; round 3) -- EXPERIMENT, built only with `make EXPERIMENTS=1` and selected by A3V_ATTN_PP=1: correct
// (the attention tests pass on it) but 160-168 us at the 7B shape against 134-138 for the 128-row kernel; what the stamps of its
// segments say is recorded in profiles/r03f_attn_pp_experiment.txt and DESIGN.md section 9.
//  The kernel above is paced by the dependency chain of ONE wave per tile (fragment
// latency -> 16 MFMAs -> ~230 VALU of softmax -> 16 MFMAs) with one partner wave from an unrelated block as its only cover: cycle
// stamps put a tile at ~4400 cycles per SIMD for 2 x 1024 cycles of MFMA.  Here a block is 8 waves = two groups of four on the
// schedule of the ring GEMM: a group's iteration is ONE MFMA segment [S(t+1) = K(t+1) Q^T ; O += V(t) P(t)] (32 MFMAs, its fragment
// reads inside) and ONE VALU segment [softmax of tile t+1, DMA issue, counted wait], the two groups are half a period apart and a
// block barrier separates the segments -- while one wave of a SIMD feeds the matrix pipe its partner exponentiates.
//   * 256 query rows per block (group g: rows 128 g .., wave: 32 rows), the blocks aligned to the END of the sequence so that the
//     ragged block is the cheapest one of a causal head (S = 1091: 18 + 14 + 10 + 6 + 2 tiles instead of 4 + 8 + 12 + 16 + 18);
//     both groups read the SAME K / V^T tiles (half the DMA and L2 traffic per query row of the 128-row kernel);
//   * K / V^T stages of 32 KiB in a ring of FIVE (all 160 KiB of the CU): stage u is issued three (group 0) / four (group 1)
//     iterations before its K tile is read, 4 LDS-DMA pieces per wave and stage, and every wave only ever waits for its OWN pieces
//     (counted vmcnt at the end of its VALU segment: everything but the last two stages it issued), the barrier publishes them;
//   * a wave whose rows lie wholly before a tile's first key (causal) keeps the barrier / DMA cadence and skips the arithmetic.
// Group g runs  M(-1) V(0) M(0) V(1) ... V(n-1) M(n-1)  on the block steps g, g+1, ...;  M(t) = [QK(t+1), PV(t)], V(t) = softmax(t).
// ------------------------------------------------------------------------------------
#ifndef PP_PRIO_ON
#define PP_PRIO_ON 0
#endif
#define PP_PRIO(x) do { if (PP_PRIO_ON) __builtin_amdgcn_s_setprio(x); } while (0)
template <bool CAUSAL>
__global__ __launch_bounds__(512) void attn_prefill_pp_kernel(AttnArgs p) {
  constexpr int HD = 128, KVB = 64, KROW = HD * 2, NST = 5;
  constexpr int TILEB = KVB * KROW + HD * 128;          // 32 KiB: K tile [64][128] + V^T tile [128][64]
  __shared__ __attribute__((aligned(1024))) char lds[NST * TILEB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
  const int nqb = (p.Sq + 255) / 256;
  int vb;
  {
    const int total = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, q = total >> 3, r = total & 7;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  int head_slot = vb / nqb, j = vb - head_slot * nqb;   // j = 0: the LAST 256 rows (heaviest under the causal mask) first
  if (CAUSAL && p.head_group > 1) {                     // tile-rank-major walk over groups of heads (see the kernel above)
    const int G = p.head_group, per = G * nqb;
    const int g2 = vb / per, r = vb - g2 * per;
    head_slot = g2 * G + r % G;
    j = r / G;
  }
  const int b = head_slot / p.H, h = head_slot - b * p.H;
  const int hk = h / (p.H / p.Hkv);
  const int hi = p.Sq - 256 * j, lo = max(0, hi - 256);  // the block's query rows [lo, hi)
  const int q0 = lo + wave * 32;                          // this wave's rows [q0, q0 + 32) ∩ [lo, hi)
  const int off = p.Sk - p.Sq;
  const bf16_t* Q = (const bf16_t*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16_t* K = (const bf16_t*)p.k + b * p.k_sb + hk * p.k_sh;
  const bf16_t* VT = (const bf16_t*)p.vt + b * p.v_sb + hk * p.v_sh;
  const int ql = lane & 31, hh = lane >> 5;
  const int qrow = q0 + ql;
  const int qrow_c = qrow < hi ? qrow : hi - 1;
  const bool wave_live = q0 < hi;
  bf16x8 qf[HD / 16];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qrow_c * p.q_ss + ks * 16 + hh * 8);
  f32x16 o[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // tiles of the block (its last row decides) and of this wave (its own last row)
  const int kv_end = CAUSAL ? min(p.Sk, hi - 1 + off + 1) : p.Sk;
  const int n = (kv_end + KVB - 1) / KVB;
  const int q_last = min(q0 + 31, hi - 1);
  const int nw = !wave_live ? 0 : (CAUSAL ? min(n, (min(p.Sk, q_last + off + 1) + KVB - 1) / KVB) : n);
  // DMA: 4 pieces of 1 KiB per wave and stage -- K rows 8 (wave) .. + 7 twice 4 rows?  a piece = 64 lanes x 16 B: K: 4 rows x 256 B,
  // V^T: 8 rows x 128 B.  The chunk swizzles of the 128-row kernel, applied on the source side.
  const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)min((int64_t)0x7fffffff, ((int64_t)(p.Sk - 1) * p.k_ss + HD) * 2), 0x00020000);
  const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)VT, 0, (int)min((int64_t)0x7fffffff, ((int64_t)(HD - 1) * p.v_sd + p.Sk) * 2), 0x00020000);
  unsigned koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + i * 512;                       // 1024 16-byte chunks per K tile: row = id / 16, physical chunk id % 16
    const int row = id >> 4, slot = id & 15;
    koff[i] = (unsigned)((row * p.k_ss + (slot ^ (row & 15)) * 8) * 2);
    const int d = id >> 3, vs = id & 7;                 // 1024 chunks per V^T tile: row d = id / 8, physical chunk id % 8
    voff[i] = (unsigned)((d * p.v_sd + (vs ^ ((d >> 1) & 7)) * 8) * 2);
  }
  auto dma_stage = [&](int u) {
    char* Ks = lds + (u % NST) * TILEB;
    char* Vs = Ks + KVB * KROW;
    const unsigned ks_off = (unsigned)((int64_t)u * KVB * p.k_ss * 2), vs_off = (unsigned)(u * KVB * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) buf_dma16(rsK, Ks + (wave * 64 + i * 512) * 16, koff[i], ks_off);
#pragma unroll
    for (int i = 0; i < 2; ++i) buf_dma16(rsV, Vs + (wave * 64 + i * 512) * 16, voff[i], vs_off);
  };
  const int kfb = ql * KROW + ((hh ^ (ql & 15)) << 4);
  const int vfb = ql * 128 + ((hh ^ ((ql >> 1) & 7)) << 4);
  f32x16 s[2];
  bf16x8 pf[2][2];

  // (slot = ring slot of the tile; the slot base only has bits above the per-lane fragment bases, so base | slot base is formed once
  //  per segment and every fragment address is one v_xor + an instruction offset)
  // Fragment order i = 0..15.  QK: (ks = i >> 1, tb = i & 1) -- the two key blocks' chains ALTERNATE, and PV: (tb, c = i >> 2, d = i & 3)
  // -- the four d blocks' chains interleave: back-to-back MFMAs on one accumulator issue at ~72 cycles instead of 32, and in this
  // kernel no partner wave fills the gaps (the other group is in its VALU segment).  The fragments are read AHEAD of the MFMAs
  // that use them, pinned by sched_group_barrier: left to itself hipcc reads every fragment into the same four registers right
  // before its MFMA (read, lgkmcnt(0), MFMA, read, ... = one LDS latency per MFMA: 2300 cycles per segment by the stamps).
  auto k_frag = [&](int kb, int i) {
    return *reinterpret_cast<const bf16x8*>(lds + (i & 1) * 32 * KROW + (kb ^ ((i >> 1) << 5)));
  };
  auto v_frag = [&](int vbs, int i) {
    const int d = i & 3, c16 = 4 * (i >> 3) + 2 * ((i >> 2) & 1);
    return *reinterpret_cast<const bf16x8*>(lds + KVB * KROW + d * 32 * 128 + (vbs ^ (c16 << 4)));
  };
  auto zero_s = [&]() {
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[tb][r] = 0.f;
  };
  auto qk = [&](int slot) {                             // S^T = K . Q^T of the tile in `slot` (first and only product of the segment)
    const int kb = kfb | (slot * TILEB);
    bf16x8 kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = k_frag(kb, i);
    zero_s();
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[i >> 1], s[i & 1], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
  };
  auto pv = [&](int slot) {                             // O^T += V^T . P^T of the tile in `slot` (alone in its segment: the last one)
    const int vbs = vfb | (slot * TILEB);
    bf16x8 vf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) vf[i] = v_frag(vbs, i);
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i], pf[i >> 3][(i >> 2) & 1], o[i & 3], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
  };
  auto qk_pv = [&](int slot1, int slot) {               // the steady segment: S^T of tile t + 1, then O^T += of tile t
    const int kb = kfb | (slot1 * TILEB), vbs = vfb | (slot * TILEB);
    bf16x8 kf[16], vf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = k_frag(kb, i);
    zero_s();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[i >> 1], s[i & 1], 0, 0, 0);
      vf[i] = v_frag(vbs, i);                            // the V^T fragments arrive under the QK MFMAs
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i], pf[i >> 3][(i >> 2) & 1], o[i & 3], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);   // 16 K reads
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one QK MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one V^T read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // 16 PV MFMAs
  };
  auto softmax = [&](int t) {                           // s (tile t) -> pf, running max / sum, lazy O rescale (see the kernel above)
    const int kv0 = t * KVB;
    const int qlim = CAUSAL ? (qrow + off) : 0x7fffffff;
    const bool need_mask = (kv0 + KVB > p.Sk) || (CAUSAL && kv0 + KVB - 1 > q0 + off);
    float mx = -INFINITY;
    if (need_mask) {
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool ok = (kv < p.Sk) && (kv <= qlim);
          s[tb][r] = ok ? s[tb][r] : -INFINITY;
        }
    }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[tb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const bool grow = (p.lazy_rescale & 1) ? ((m_new - m_run) * p.scale_log2 > 8.f || m_run == -INFINITY) : true;
    const bool resc = __builtin_amdgcn_ballot_w64(grow && m_new != m_run) != 0;
    const float m_tgt = resc ? m_new : m_run;
    const float m_use = (m_tgt == -INFINITY) ? 0.f : m_tgt;
    float alpha = 1.f;
    if (resc) alpha = __builtin_amdgcn_exp2f((m_run - m_use) * p.scale_log2);
    m_run = m_tgt;
    float lsum = 0.f;
    const float mb = m_use * p.scale_log2;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(fmaf(s[tb][r], p.scale_log2, -mb));
        lsum += pr;
        pf[tb][r >> 3][r & 7] = f2bf(pr);
      }
    l_run = resc ? l_run * alpha + lsum : l_run + lsum;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 w;
        __builtin_memcpy(&w, &pf[tb][c], 16);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(w[e], w[2 + e], false, false);
          w[e] = sw[0];
          w[2 + e] = sw[1];
        }
        __builtin_memcpy(&pf[tb][c], &w, 16);
      }
    if (resc) {
#pragma unroll
      for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
  };
  // counted wait at the end of V(t): everything but the stages this group issued beyond t + 1 + grp may stay in flight
  auto wait_own = [&](int t) {
    const int last = min(n - 1, t + 3 + grp), allowed = last - (t + 1 + grp);
    if (allowed >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (allowed == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // V^T columns past Sk of a ragged last tile are zeroed in LDS (0 x garbage must stay 0): by group 0 in its V(n-1), when every
  // piece of the stage has landed (its own: confirmed in V(n-2), group 1's in V(n-3)) and before anyone's PV(n-1)
  auto zero_ragged = [&]() {
    const int c0 = p.Sk - (n - 1) * KVB;                 // first invalid column of the tile
    if (c0 >= KVB || grp != 0) return;
    char* Vs = lds + ((n - 1) % NST) * TILEB + KVB * KROW;
    const int d = tid >> 1, half = tid & 1;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = half * 4 + cc;
      char* chunk = Vs + d * 128 + ((c ^ ((d >> 1) & 7)) << 4);
      if (c * 8 >= c0) {
        *reinterpret_cast<u32x4*>(chunk) = u32x4{0u, 0u, 0u, 0u};
      } else if (c * 8 + 8 > c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c * 8 + e >= c0) reinterpret_cast<unsigned short*>(chunk)[e] = 0;
      }
    }
  };

  // ---- prologue: group 0 issues stages 0..2, group 1 stages 0..3; stage 0 (group 1: and 1) confirmed before the first barrier
  {
    const int pre = min(n, 3 + grp);
    for (int u = 0; u < pre; ++u) dma_stage(u);
    const int keep = max(0, pre - 1 - grp);              // stages that may stay in flight
    if (keep >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (keep == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- the 2 n + 2 block steps.  Every wave executes every barrier; a group's segments are M(-1) V(0) M(0) ... V(n-1) M(n-1), group 1
  // one step behind group 0.  Straight-line code per segment kind (a single loop that picked M or V by the step's parity made
  // hipcc shuffle all 96 accumulator registers through v_mov at every merge: 216 moves per iteration, 2.8x slower than the
  // 128-row kernel).  A wave's live part covers its own nw tiles; the rest of the block's n tiles it only paces (barriers, DMA).
  // raw s_barrier: __syncthreads() would also wait for vmcnt(0), i.e. for every LDS-DMA piece in flight (the whole point of the ring);
  // LDS hazards are covered by the counted vmcnt (DMA -> reads) and by the lgkmcnt(0) behind the one segment that writes LDS.
#ifdef PP_STAMP
  unsigned long long* stamps = (vb == PP_STAMP && (tid == 0 || tid == 256)) ? reinterpret_cast<unsigned long long*>(p.lse) + (tid ? 512 : 0) : nullptr;
  int sti = 0;
#define PP_ST(code) do { if (stamps && sti < 250) { stamps[2 * sti] = (code); stamps[2 * sti + 1] = __builtin_amdgcn_s_memtime(); ++sti; } } while (0)
#else
#define PP_ST(code) do {} while (0)
#endif
  auto bar = [&]() { PP_ST(1); __builtin_amdgcn_s_barrier(); PP_ST(2); };
  auto next = [&](int sl) { return sl + 1 == NST ? 0 : sl + 1; };
  auto vseg_io = [&](int t) {                            // the part of V(t) every wave does: issue stage t + 3 + grp, zero the ragged tail, counted wait
    if (t + 3 + grp < n) dma_stage(t + 3 + grp);
    if (t == n - 1) {
      zero_ragged();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    wait_own(t);
  };
  if (grp == 1) bar();
  int slot = 0;                                          // slot of tile t in the loop below
  if (nw > 0) {
    bar();                                               // M(-1)
    PP_PRIO(1);
    qk(0);
    PP_PRIO(0);
    bar();                                               // V(0)
    softmax(0);
    vseg_io(0);
    int t = 0;
    for (; t + 1 < nw; ++t) {
      const int slot1 = next(slot);
      bar();                                             // M(t)
      PP_PRIO(1);
      qk_pv(slot1, slot);
      PP_ST(3);
      PP_PRIO(0);
      PP_ST(4);
      bar();                                             // V(t + 1)
      softmax(t + 1);
      PP_ST(5);
      vseg_io(t + 1);
      PP_ST(6);
      slot = slot1;
    }
    bar();                                               // M(nw - 1)
    PP_PRIO(1);
    pv(slot);
    PP_PRIO(0);
  }
  // pacing part: segments 2 nw + 1 .. 2 n of this wave's group (or all 2 n + 1 of them for a wave with no rows)
  for (int seg = nw > 0 ? 2 * nw + 1 : 0; seg <= 2 * n; ++seg) {
    bar();
    if (seg & 1) vseg_io(seg >> 1);
  }
  if (grp == 0) bar();
  // ---- normalise and store O[q][d], d = 32*db + 8*g + 4*hh + {0..3} (through the wave's LDS patch as whole rows)
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
#ifdef PP_STAMP
  if (stamps) return;
  if (false)
#else
  if (p.lse && qrow < hi && hh == 0)
#endif
    p.lse[((int64_t)b * p.H + h) * p.Sq + qrow] = m_run * p.scale + __logf(l_tot);
  __syncthreads();
  {
    constexpr int ROWB = HD * 2, NPAIR = HD / 8, RPI = 64 / NPAIR;
    char* patch = lds + wave * (32 * ROWB);
    char* wrow = patch + ql * ROWB;
    const int wx = (ql & (NPAIR - 1)) << 1;
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[d][g4 * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(wrow + (((d * 8 + g4 * 2 + hh) ^ wx) << 3)) = ov;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int pr = lane % NPAIR, rr = lane / NPAIR;
    bf16_t* Ob = (bf16_t*)p.out + b * p.o_sb + h * p.o_sh + pr * 8;
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int r = it * RPI + rr;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + r * ROWB + ((pr ^ (r & (NPAIR - 1))) << 4));
      if (q0 + r < hi) *reinterpret_cast<bf16x8*>(Ob + (int64_t)(q0 + r) * p.o_ss) = v;
    }
  }
}

#endif  // A3V_EXPERIMENTS

// Fused combine of the decode kernels (all threads of the block call it; `po` = this block's partial, already written with plain stores)
template <int HD>
__device__ __forceinline__ void decode_combine_tail(const AttnArgs& p, float* part, float* po, int nsplit, int* counters, int b, int h, int tid) {
  // Fused combine (decode step): the block that arrives last at the (batch, head) counter merges the nsplit partials.
  // The splits of a head may run on different XCDs (private L2s), so the hand-off uses agent-coherent accesses: this
  // block re-writes its HD+2 partial values with sc0 sc1 stores (write-through), waits for the acknowledgement, bumps
  // the counter; the last block reads all partials with sc0 sc1 loads.  (An agent-scope fence instead = L2 write-back
  // + invalidate per wave: measured 5x slower on the GEMV that uses the same scheme.)
  __shared__ int last_s;
  __syncthreads();                                    // po[] of this block is complete (plain stores, same CU)
  if (tid < HD + 2) st_agent(po + tid, po[tid]);      // L1 is write-through: the value read back is this block's own
  agent_wait();
  __syncthreads();
  if (tid == 0) {
    int* ctr = counters + b * p.H + h;
    const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = old == nsplit - 1;
    if (old == nsplit - 1) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last_s || tid >= HD) return;
  const float* pp = part + (int64_t)(b * p.H + h) * nsplit * (HD + 2);
  float acc = 0.f, l = 0.f;
  if (nsplit <= 8) {                                  // all loads in flight at once: one memory round trip
    float mv[8], av[8], lv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float* q = pp + (s < nsplit ? s : nsplit - 1) * (HD + 2);
      mv[s] = ld_agent_issue(q + HD);
      av[s] = ld_agent_issue(q + tid);
      lv[s] = ld_agent_issue(q + HD + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(mv[0]), "+v"(mv[1]), "+v"(mv[2]), "+v"(mv[3]), "+v"(mv[4]), "+v"(mv[5]), "+v"(mv[6]), "+v"(mv[7])::"memory");
    asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]));
    asm volatile("" : "+v"(lv[0]), "+v"(lv[1]), "+v"(lv[2]), "+v"(lv[3]), "+v"(lv[4]), "+v"(lv[5]), "+v"(lv[6]), "+v"(lv[7]));
    float m = -INFINITY;
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s < nsplit) m = fmaxf(m, mv[s]);
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < nsplit) {
        const float w = (mv[s] == -INFINITY) ? 0.f : __expf(mv[s] - m);
        acc += w * av[s];
        l += w * lv[s];
      }
  } else {
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, ld_agent(pp + s * (HD + 2) + HD));
    for (int s = 0; s < nsplit; ++s) {
      const float ms = ld_agent(pp + s * (HD + 2) + HD);
      const float a1 = ld_agent(pp + s * (HD + 2) + tid);
      const float l1 = ld_agent(pp + s * (HD + 2) + HD + 1);
      const float w = (ms == -INFINITY) ? 0.f : __expf(ms - m);
      acc += w * a1;
      l += w * l1;
    }
  }
  ((bf16_t*)p.out)[b * p.o_sb + h * p.o_sh + tid] = f2bf(acc / l);
}

// ------------------------------------------------------------------------------------
// Decode (Sq == 1): grid (nsplit, H, B); block 256.  part[b][h][split] = {o[HD], m, l}
// ------------------------------------------------------------------------------------
template <int HD, bool NT = false>   // NT: non-temporal policy on the K / V^T stream (read once per step by one block)
__global__ __launch_bounds__(256) void attn_decode_bf16_kernel(AttnArgs p, float* part, int nsplit, int chunk, int* counters) {
  constexpr int LPR = HD / 8;            // lanes per K row (16-B each)
  constexpr int RPW = 64 / LPR;          // K rows per wave-load
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  float* sc = reinterpret_cast<float*>(dsm);       // chunk scores / probabilities
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (p.H / p.Hkv);
  const int kv_lo = sp * chunk;
  const int kv_hi = min(p.Sk, kv_lo + chunk);
  const int n = kv_hi - kv_lo;
  float* po = part + ((int64_t)(b * p.H + h) * nsplit + sp) * (HD + 2);
  if (n <= 0) {          // cannot happen with decode_plan's splits (every split owns >= 1 key); kept for safety
    if (tid < HD) po[tid] = 0.f;
    if (tid == 0) { po[HD] = -INFINITY; po[HD + 1] = 0.f; }
    if (!counters) return;
  }
  const bf16_t* Q = (const bf16_t*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16_t* K = (const bf16_t*)p.k + b * p.k_sb + hk * p.k_sh;
  const bf16_t* VT = (const bf16_t*)p.vt + b * p.v_sb + hk * p.v_sh;
  if (n > 0) {
  // phase 1: scores.  lane -> (row-in-group, 8 d's); U independent row groups in flight per wave
  float qv[8];
  load8(Q + (lane % LPR) * 8, qv);
  constexpr int U = 8;
  const int lr = lane / LPR, lc = (lane % LPR) * 8;
  for (int r0 = wave * RPW * U; r0 < n; r0 += 4 * RPW * U) {
    bf16x8 kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int r = r0 + u * RPW + lr;
      r = r < n ? r : n - 1;
      const bf16x8* kp = reinterpret_cast<const bf16x8*>(K + (int64_t)(kv_lo + r) * p.k_ss + lc);
      kk[u] = NT ? __builtin_nontemporal_load(kp) : *kp;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf(qv[e], (float)kk[u][e], acc);
#pragma unroll
      for (int o2 = LPR / 2; o2 > 0; o2 >>= 1) acc += __shfl_xor(acc, o2, 64);
      const int r = r0 + u * RPW + lr;
      if (r < n && (lane % LPR) == 0) sc[r] = acc * p.scale;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int i = tid; i < n; i += 256) mx = fmaxf(mx, sc[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float ls = 0.f;
  for (int i = tid; i < n; i += 256) {
    const float e = __expf(sc[i] - mx);
    // P is rounded to bf16 before P.V in the prefill kernel and in bf16 SDPA; keep fp32 sum
    sc[i] = e;
    ls += e;
  }
  ls = wave_sum(ls);
  __syncthreads();
  if (lane == 0) red[4 + wave] = ls;
  __syncthreads();
  ls = red[4] + red[5] + red[6] + red[7];
  // zero-pad the probabilities to a multiple of 8 for the vector loop
  const int n8 = (n + 7) & ~7;
  for (int i = n + tid; i < n8; i += 256) sc[i] = 0.f;
  __syncthreads();

  // phase 2: o[d] = sum_kv p[kv] * VT[d][kv].  wave -> HD/4 consecutive d rows, 8 rows (8 x 16 B per lane) in
  // flight per batch, lanes -> 8-kv chunks of the row; one shuffle reduction per row at the end of a batch.
  // Loads are unconditional 16-B vectors (kv_lo + n8 <= Skmax: the cache row is allocated to a multiple of 64);
  // columns >= Sk are masked on the VALUES after the load (a select on the load itself would serialise them).
  constexpr int DB = 8;
  for (int d0 = wave * (HD / 4); d0 < (wave + 1) * (HD / 4); d0 += DB) {
    float acc[DB];
#pragma unroll
    for (int j = 0; j < DB; ++j) acc[j] = 0.f;
    for (int c = lane * 8; c < n8; c += 64 * 8) {
      bf16x8 vv[DB];
#pragma unroll
      for (int j = 0; j < DB; ++j) {
        const bf16x8* vp = reinterpret_cast<const bf16x8*>(VT + (int64_t)(d0 + j) * p.v_sd + kv_lo + c);
        vv[j] = NT ? __builtin_nontemporal_load(vp) : *vp;
      }
      float pv[8];
      const int nvalid = p.Sk - (kv_lo + c);          // >= 8 except in the last chunk
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = e < nvalid ? sc[c + e] : 0.f;
#pragma unroll
      for (int j = 0; j < DB; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j] = fmaf(pv[e], e < nvalid ? (float)vv[j][e] : 0.f, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < DB; ++j) {
      const float t = wave_sum(acc[j]);
      if (lane == 0) po[d0 + j] = t;
    }
  }
  if (tid == 0) { po[HD] = mx; po[HD + 1] = ls; }
  }  // n > 0
  if (!counters) return;
  decode_combine_tail<HD>(p, part, po, nsplit, counters, b, h, tid);
}

// Sums over the 16 (8) lanes of a DPP row (half row) with four (three) v_add_f32_dpp: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror.  (__shfl_xor compiles to ds_bpermute_b32: an LDS-pipe round trip per step.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)));
}
__device__ __forceinline__ float row8_sum(float v) { return dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(v))); }
__device__ __forceinline__ float row16_sum(float v) { return dpp_add<0x140>(row8_sum(v)); }
__device__ __forceinline__ float row8_max(float v) { return dpp_max<0x141>(dpp_max<0x4E>(dpp_max<0xB1>(v))); }

// ------------------------------------------------------------------------------------
// Decode, wave-streaming form (round 2): grid (nsplit, H, B), block 512 = 8 waves; nsplit = 1 when B*H alone fills the CUs.
// The first form streams K, synchronises the block for the softmax, then streams V^T: every block of the chip changes phase
// together and ~17 of its 37 us at context 1100 did not scale with the bytes (tools/attn_decode_bench.py).  Here each WAVE owns
// every eighth 64-key tile: the tile's K rows (16 KB) AND its V^T columns (128 B of each of the HD rows, 16 KB) are in flight
// together (32 x 16 B per lane), the next tile's K rows are requested before this tile's softmax, scores -> wave-private online softmax -> P.V
// without leaving the wave; the eight waves merge (m, l, o) once through LDS.  No block-wide phase change, and with one block
// per (batch, head) no cross-block hand-off either.
// ------------------------------------------------------------------------------------
template <int HD, bool NT>
__global__ __launch_bounds__(512) void attn_decode_wave_kernel(AttnArgs p, float* part, int nsplit, int chunk, int* counters) {
  constexpr int LPR = HD / 8;        // lanes per K row (16 B each)
  constexpr int RPW = 64 / LPR;      // K rows per wave-wide load
  constexpr int NKL = 64 / RPW;      // K loads per 64-key tile
  constexpr int NVL = HD / 8;        // V^T loads per tile (8 d rows x 128 B per load)
  __shared__ __attribute__((aligned(16))) float sc_s[8][64];
  __shared__ float comb[8][HD + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (p.H / p.Hkv);
  const int kv_lo = sp * chunk;
  const int n = min(p.Sk, kv_lo + chunk) - kv_lo;
  float* po = part + ((int64_t)(b * p.H + h) * nsplit + sp) * (HD + 2);
  const bf16_t* Q = (const bf16_t*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16_t* K = (const bf16_t*)p.k + b * p.k_sb + hk * p.k_sh + (int64_t)kv_lo * p.k_ss;
  const bf16_t* VT = (const bf16_t*)p.vt + b * p.v_sb + hk * p.v_sh + kv_lo;
  // 64-key tiles of the block's n keys go round-robin to the eight waves: every V^T request is a whole, aligned 128-byte line
  // (contiguous eighths of the keys start at multiples of 8 keys: Sk = 1110 put every wave 32 B into a line, 43 us against 30)
  const int lo = wave * 64, hi = max(n, 0);
  float m = -INFINITY, l = 0.f, acc[NVL];
#pragma unroll
  for (int j = 0; j < NVL; ++j) acc[j] = 0.f;
  const int lr = lane / LPR, lc = (lane % LPR) * 8;
  const int dr = lane >> 3, c8 = (lane & 7) * 8;
  if (hi > lo) {
    float qv[8];
    load8(Q + lc, qv);
    const int last_vec = (hi - 1) & ~7;
    bf16x8 kk[NKL];
    auto load_k = [&](int t0) {
#pragma unroll
      for (int u = 0; u < NKL; ++u) {
        const int r = min(t0 + u * RPW + lr, hi - 1);
        const bf16x8* kp = reinterpret_cast<const bf16x8*>(K + (int64_t)r * p.k_ss + lc);
        kk[u] = NT ? __builtin_nontemporal_load(kp) : *kp;
      }
    };
    load_k(lo);
    // software pipeline: V^T(t) is requested before the scores of tile t are computed, K(t+1) before its softmax / P.V -- the
    // memory system always has 16-32 KB per wave outstanding (without it every wave of the chip asks for its 32 KB at the same
    // moment, computes while HBM idles, and asks again: 40 us instead of the 33 of the two-phase form)
    for (int t0 = lo; t0 < hi; t0 += 512) {
      bf16x8 vv[NVL];
      const int kvc = min(t0 + c8, last_vec);              // vectors past the wave's range re-read its last one (masked below)
#pragma unroll
      for (int j = 0; j < NVL; ++j) {
        const bf16x8* vp = reinterpret_cast<const bf16x8*>(VT + (int64_t)(j * 8 + dr) * p.v_sd + kvc);
        vv[j] = NT ? __builtin_nontemporal_load(vp) : *vp;
      }
#pragma unroll
      for (int u = 0; u < NKL; ++u) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(qv[e], (float)kk[u][e], a);
        a = LPR == 16 ? row16_sum(a) : row8_sum(a);
        if ((lane % LPR) == 0) sc_s[wave][u * RPW + lr] = (t0 + u * RPW + lr < hi) ? a * p.scale : -INFINITY;
      }
      if (t0 + 512 < hi) load_k(t0 + 512);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // wave-private round trip through LDS: no barrier, in-order LDS
      const f32x4 sa = *reinterpret_cast<const f32x4*>(&sc_s[wave][c8]);
      const f32x4 sb = *reinterpret_cast<const f32x4*>(&sc_s[wave][c8 + 4]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float s8[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
      float mt = s8[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) mt = fmaxf(mt, s8[e]);
      mt = row8_max(mt);
      const float mn = fmaxf(m, mt);                         // finite: the tile holds at least one key of the range
      const float alpha = (m == -INFINITY) ? 0.f : __expf(m - mn);
      float pe[8], ps = 0.f;
      const int nvalid = hi - (t0 + c8);                     // <= 0 for the vectors past the range
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pe[e] = e < nvalid ? __expf(s8[e] - mn) : 0.f;
        ps += pe[e];
      }
      l = l * alpha + ps;
#pragma unroll
      for (int j = 0; j < NVL; ++j) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(pe[e], e < nvalid ? (float)vv[j][e] : 0.f, a);   // the cache tail may hold anything
        acc[j] = acc[j] * alpha + a;
      }
      m = mn;
    }
  }
#pragma unroll
  for (int j = 0; j < NVL; ++j) {
    acc[j] = row8_sum(acc[j]);
  }
  l = row8_sum(l);
  if ((lane & 7) == 0) {
#pragma unroll
    for (int j = 0; j < NVL; ++j) comb[wave][j * 8 + dr] = acc[j];
  }
  if (lane == 0) { comb[wave][HD] = m; comb[wave][HD + 1] = l; }
  __syncthreads();
  if (tid < HD) {
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 8; ++w) M = fmaxf(M, comb[w][HD]);
    float o = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float mw = comb[w][HD];
      const float sw = (mw == -INFINITY) ? 0.f : __expf(mw - M);
      o += sw * comb[w][tid];
      L += sw * comb[w][HD + 1];
    }
    if (nsplit == 1 && counters) {
      ((bf16_t*)p.out)[b * p.o_sb + h * p.o_sh + tid] = f2bf(o / L);
    } else {
      po[tid] = o;
      if (tid == 0) { po[HD] = M; po[HD + 1] = L; }
    }
  }
  if (!counters || nsplit == 1) return;
  decode_combine_tail<HD>(p, part, po, nsplit, counters, b, h, tid);
}

template <int HD>
__global__ void attn_decode_combine_kernel(const float* part, void* out, int64_t o_sb, int64_t o_sh,
                                           int H, int nsplit, int dtype) {
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const float* pp = part + (int64_t)(b * H + h) * nsplit * (HD + 2);
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, pp[s * (HD + 2) + HD]);
  float acc = 0.f, l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float ms = pp[s * (HD + 2) + HD];
    const float w = (ms == -INFINITY) ? 0.f : __expf(ms - m);
    acc += w * pp[s * (HD + 2) + d];
    l += w * pp[s * (HD + 2) + HD + 1];
  }
  const float v = acc / l;
  if (dtype == A3V_BF16) ((bf16_t*)out)[b * o_sb + h * o_sh + d] = f2bf(v);
  else ((float*)out)[b * o_sb + h * o_sh + d] = v;
}

// ------------------------------------------------------------------------------------
// Generic attention (fp32 parity path; also bf16 with head dims the MFMA kernels do not
// cover): one wave per (q row, head, batch); any hd <= 256, any Sq.  fp32 math throughout.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void attn_generic_kernel(AttnArgs p, int hd, int causal) {
  __shared__ float pbuf[64];
  __shared__ float qs[256];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (p.H / p.Hkv);
  const T* Q = (const T*)p.q + b * p.q_sb + (int64_t)qi * p.q_ss + h * p.q_sh;
  const T* K = (const T*)p.k + b * p.k_sb + hk * p.k_sh;
  const T* VT = (const T*)p.vt + b * p.v_sb + hk * p.v_sh;
  for (int d = lane; d < hd; d += 64) qs[d] = Cvt<T>::ld(Q + d);
  __syncthreads();
  const int kv_end = causal ? min(p.Sk, qi + (p.Sk - p.Sq) + 1) : p.Sk;
  float m = -INFINITY, l = 0.f;
  float o[4] = {0.f, 0.f, 0.f, 0.f};   // d = lane + 64*i
  for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
    const int kv = kv0 + lane;
    float s = -INFINITY;
    if (kv < kv_end) {
      const T* kr = K + (int64_t)kv * p.k_ss;
      float a = 0.f;
      for (int d = 0; d < hd; ++d) a = fmaf(qs[d], Cvt<T>::ld(kr + d), a);
      s = a * p.scale;
    }
    const float mt = wave_max(s);
    const float mn = fmaxf(m, mt);
    const float alpha = __expf(m - mn);   // m = -inf on the first tile -> 0
    const float pv = (kv < kv_end) ? __expf(s - mn) : 0.f;
    l = l * alpha + wave_sum(pv);
    m = mn;
    __syncthreads();
    pbuf[lane] = pv;
    __syncthreads();
    const int cnt = min(64, kv_end - kv0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = lane + 64 * i;
      if (d < hd) {
        const T* vr = VT + (int64_t)d * p.v_sd + kv0;
        float a = o[i] * alpha;
        for (int j = 0; j < cnt; ++j) a = fmaf(pbuf[j], Cvt<T>::ld(vr + j), a);
        o[i] = a;
      }
    }
  }
  T* O = (T*)p.out + b * p.o_sb + (int64_t)qi * p.o_ss + h * p.o_sh;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = lane + 64 * i;
    if (d < hd) Cvt<T>::st(O + d, o[i] / l);
  }
  if (p.lse && lane == 0) p.lse[((int64_t)b * p.H + h) * p.Sq + qi] = m + __logf(l);
}

inline void decode_plan(int B, int H, int Sk, int* nsplit, int* chunk) {
  // ~768 blocks: at B*H = 256 three splits of the context beat four by 5-8 % for 600..1500 keys (larger chunks fill the 64 lanes of
  // the V^T scan and the K scan's 128-row passes; sweep with tools/attn_decode_bench.py under -DA3V_ABLATION, A3V_DECODE_WANT)
#ifndef A3V_DECODE_BLOCKS
#define A3V_DECODE_BLOCKS 768
#endif
  int want = (A3V_DECODE_BLOCKS + B * H - 1) / (B * H);
#ifdef A3V_ABLATION
  { const int e = A3V_ENV_INT("A3V_DECODE_WANT", 0); if (e) want = e; }
#endif
  int maxs = (Sk + 127) / 128;
  int ns = want < maxs ? want : maxs;
  if (ns < 1) ns = 1;
  int ch = (Sk + ns - 1) / ns;
  ch = (ch + 63) & ~63;
  ns = (Sk + ch - 1) / ch;
  *nsplit = ns;
  *chunk = ch;
}

}  // namespace

extern "C" int64_t a3v_attention_scratch_floats(int B, int H, int hd, int Sk) {
  int ns, ch;
  decode_plan(B, H, Sk, &ns, &ch);
  return (int64_t)B * H * ns * (hd + 2);
}

#ifdef A3V_EXPERIMENTS
static int attn_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v & ~7;
    if (n <= 0) n = 256;
  }
  return n;
}

// Unit-queue counters of the persistent prefill kernel: 16 ints per launch (8 queues, 1 exit count), handed out round-robin from a
// per-device pool of 256 slots that is allocated and zeroed ONCE (every launch leaves its slot zero), so launches in flight on
// different streams never share a slot.  nullptr (e.g. first use under stream capture, where hipMalloc is illegal) = the caller
// takes the one-block-per-unit launch.
static int* attn_counter_slot() {
  constexpr int NSLOT = 256, MAXDEV = 16;
  static int* pool[MAXDEV] = {};
  static std::atomic<unsigned> seq{0};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  if (!pool[dev]) {
    std::lock_guard<std::mutex> g(mu);
    if (!pool[dev]) {
      int* ptr = nullptr;
      if (hipMalloc(&ptr, NSLOT * 16 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      if (hipMemset(ptr, 0, NSLOT * 16 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ptr); return nullptr; }
      pool[dev] = ptr;
    }
  }
  return pool[dev] + (seq.fetch_add(1) % NSLOT) * 16;
}
#endif

static int attention_impl(const void* q, const void* k, const void* vt, void* out, int B, int Sq, int Sk,
                          int H, int Hkv, int hd, const int64_t* strides, int causal, float* scratch,
                          float* lse, int dtype, void* stream) {
  if (!q || !k || !vt || !out || !strides || B <= 0 || Sq <= 0 || Sk <= 0 || H <= 0 || Hkv <= 0) return A3V_ERR_ARG;
  if (H % Hkv) return A3V_ERR_SHAPE;
  if (causal && Sk < Sq) return A3V_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  AttnArgs p;
  p.q = q; p.k = k; p.vt = vt; p.out = out;
  p.q_sb = strides[0]; p.q_ss = strides[1]; p.q_sh = strides[2];
  p.k_sb = strides[3]; p.k_sh = strides[4]; p.k_ss = strides[5];
  p.v_sb = strides[6]; p.v_sh = strides[7]; p.v_sd = strides[8];
  p.o_sb = strides[9]; p.o_ss = strides[10]; p.o_sh = strides[11];
  p.B = B; p.Sq = Sq; p.Sk = Sk; p.H = H; p.Hkv = Hkv;
  p.scale = 1.0f / sqrtf((float)hd);
  p.scale_log2 = p.scale * 1.4426950408889634f;
  p.lse = lse;
  p.head_group = 1;
  p.staged_o = A3V_ENV_INT("A3V_ATTN_STAGED_O", 1) != 0;
  p.lazy_rescale = A3V_ENV_INT("A3V_ATTN_LAZY", 1);     // bit 0: lazy rescale; bit 1: ds_bpermute form of the row-maximum exchange (A/B)
  if (lse && Sq == 1 && dtype == A3V_BF16 && (hd == 64 || hd == 128)) return A3V_ERR_ARG;  // decode kernel has no LSE output
  if (dtype == A3V_F32) {
    if (hd > 256) return A3V_ERR_SHAPE;
    hipLaunchKernelGGL(attn_generic_kernel<float>, dim3(Sq, H, B), dim3(64), 0, st, p, hd, causal);
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  if (dtype != A3V_BF16) return A3V_ERR_DTYPE;
  if (hd != 128 && hd != 64) {   // bf16 with an uncommon head dim: generic (slow, correct) kernel
    if (hd > 256) return A3V_ERR_SHAPE;
    hipLaunchKernelGGL(attn_generic_kernel<bf16_t>, dim3(Sq, H, B), dim3(64), 0, st, p, hd, causal);
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  for (int i = 0; i < 12; ++i)
    if (i != 2 && i != 4 && i != 11 && (strides[i] % 8)) return A3V_ERR_SHAPE;  // 16-B vector access
  if ((strides[2] % 8) || (strides[4] % 8) || (strides[11] % 4)) return A3V_ERR_SHAPE;
  if (Sq == 1) {
    if (!scratch) return A3V_ERR_ARG;
    int ns, ch;
    decode_plan(B, H, Sk, &ns, &ch);
    const size_t shm = (size_t)(ch + 8) * sizeof(float);
    const bool we = A3V_ENV_INT("A3V_ATTN_DECODE_WAVE_STANDALONE", 0) == 1;     // tuning runs (tools/attn_decode_bench.py): the wave-streaming form here too
    if (we && dtype == A3V_BF16) {
      int ns2 = (256 + B * H - 1) / (B * H);
      if (ns2 > ns) ns2 = ns;
      int ch2 = (((Sk + ns2 - 1) / ns2) + 63) & ~63;
      ns2 = (Sk + ch2 - 1) / ch2;
      if (hd == 128) {
        hipLaunchKernelGGL((attn_decode_wave_kernel<128, true>), dim3(ns2, H, B), dim3(512), 0, st, p, scratch, ns2, ch2, (int*)nullptr);
        hipLaunchKernelGGL(attn_decode_combine_kernel<128>, dim3(H, B), dim3(128), 0, st, scratch, out, p.o_sb, p.o_sh, H, ns2, dtype);
      } else {
        hipLaunchKernelGGL((attn_decode_wave_kernel<64, true>), dim3(ns2, H, B), dim3(512), 0, st, p, scratch, ns2, ch2, (int*)nullptr);
        hipLaunchKernelGGL(attn_decode_combine_kernel<64>, dim3(H, B), dim3(64), 0, st, scratch, out, p.o_sb, p.o_sh, H, ns2, dtype);
      }
      A3V_LAUNCH_CHECK();
      return A3V_OK;
    }
    if (hd == 128) {
      hipLaunchKernelGGL(attn_decode_bf16_kernel<128>, dim3(ns, H, B), dim3(256), shm, st, p, scratch, ns, ch, (int*)nullptr);
      A3V_LAUNCH_CHECK();
      hipLaunchKernelGGL(attn_decode_combine_kernel<128>, dim3(H, B), dim3(128), 0, st, scratch, out, p.o_sb, p.o_sh, H, ns, dtype);
    } else {
      hipLaunchKernelGGL(attn_decode_bf16_kernel<64>, dim3(ns, H, B), dim3(256), shm, st, p, scratch, ns, ch, (int*)nullptr);
      A3V_LAUNCH_CHECK();
      hipLaunchKernelGGL(attn_decode_combine_kernel<64>, dim3(H, B), dim3(64), 0, st, scratch, out, p.o_sb, p.o_sh, H, ns, dtype);
    }
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
#ifdef A3V_EXPERIMENTS
  // hd 128, long enough sequences, vector-aligned output: the 8-wave ping-pong kernel (experiment, A3V_ATTN_PP=1)
  if (hd == 128 && Sq >= 256 && A3V_ENV_INT("A3V_ATTN_PP", 0) != 0 && !(strides[9] % 8) && !(strides[10] % 8) && !(strides[11] % 8) &&
      !(reinterpret_cast<uintptr_t>(out) & 15) && (int64_t)Sk * strides[5] * 2 < (1LL << 31) && (int64_t)hd * strides[8] * 2 < (1LL << 31)) {
    dim3 gp(((Sq + 255) / 256) * H * B);
    if (causal && (gp.x & 7) == 0 && ((B * H) & 7) == 0) {
      const int ge = A3V_ENV_INT("A3V_ATTN_HEAD_GROUP", 0);
      int want = 16;
      while (want > 1 && (int64_t)want * Sk * hd * 4 > (9 << 20)) want >>= 1;
      if (ge > 0) want = ge;
      int G = want < 1 ? 1 : want;
      while (G > 1 && ((B * H) / 8) % G) G >>= 1;
      p.head_group = G;
    }
    if (causal) hipLaunchKernelGGL(attn_prefill_pp_kernel<true>, gp, dim3(512), 0, st, p);
    else hipLaunchKernelGGL(attn_prefill_pp_kernel<false>, gp, dim3(512), 0, st, p);
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
#endif
  dim3 grid(((Sq + 127) / 128) * H * B);
#ifdef A3V_EXPERIMENTS
  // hd 128: the one-wave-per-SIMD kernel with 64 query rows per wave (experiment, A3V_ATTN_W64=1)
  const bool w64 = hd == 128 && A3V_ENV_INT("A3V_ATTN_W64", 0) != 0 && !(strides[9] % 8) && !(strides[10] % 8) && !(strides[11] % 8) &&
                   !(reinterpret_cast<uintptr_t>(out) & 15) && (int64_t)Sk * strides[5] * 2 < (1LL << 31) && (int64_t)hd * strides[8] * 2 < (1LL << 31);
#else
  const bool w64 = false;
#endif
  if (causal && (grid.x & 7) == 0 && ((B * H) & 7) == 0) {
    // default: the largest power of two (<= 16) whose K + V^T fit ~9 MB (measured best: 16 heads at S = 1091, 8 at S ~ 2000 --
    // twice the 4-MB L2, the Infinity Cache absorbs the rest; tools/ab_attn_order.py): 160.8 -> 142.3 us at S = 1091
    const int ge = A3V_ENV_INT("A3V_ATTN_HEAD_GROUP", 0);      // > 0: forces the group size (A/B runs)
    int want = 16;
    while (want > 1 && (int64_t)want * Sk * hd * 4 > (9 << 20)) want >>= 1;
    if (ge > 0) want = ge;
    int G = want < 1 ? 1 : want;
    while (G > 1 && ((B * H) / 8) % G) G >>= 1;            // groups must not straddle an XCD's range of heads
    p.head_group = G;
  }
#ifdef A3V_EXPERIMENTS
  // persistent walk (round 6 experiment, A3V_ATTN_PERSIST=1): two blocks per CU pull (batch, head, query tile) units from per-XCD
  // queues; taken whenever there are more units than resident blocks and a counter slot is available
  const int n_units = (int)grid.x;
  int* ctr = nullptr;
  const int bpc = hd == 128 ? 2 : 3;          // resident blocks per CU (registers: 256 / 164 per lane)
  if (!w64 && A3V_ENV_INT("A3V_ATTN_PERSIST", 0) != 0 && A3V_ENV_INT("A3V_ATTN_PSWAP", 1) != 0 && n_units > bpc * attn_cu_count() &&
      (causal || hd == 64) &&                 // (hd 128 without the mask is not a shape of this model; its build of the walk spills)
      (int64_t)Sk * strides[5] * 2 < (1LL << 31) && (int64_t)hd * strides[8] * 2 < (1LL << 31))
    ctr = attn_counter_slot();
  if (ctr) {
    const dim3 pg(bpc * attn_cu_count());
    if (hd == 128) {
      hipLaunchKernelGGL((attn_prefill_persist_kernel<128, true>), pg, dim3(256), 0, st, p, ctr, n_units);
    } else {
      if (causal) hipLaunchKernelGGL((attn_prefill_persist_kernel<64, true>), pg, dim3(256), 0, st, p, ctr, n_units);
      else hipLaunchKernelGGL((attn_prefill_persist_kernel<64, false>), pg, dim3(256), 0, st, p, ctr, n_units);
    }
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
#endif
  if (w64) {
#ifdef A3V_EXPERIMENTS
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute((const void*)attn_prefill_w64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      (void)hipFuncSetAttribute((const void*)attn_prefill_w64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr_done = true;
    }
    if (causal) hipLaunchKernelGGL(attn_prefill_w64_kernel<true>, grid, dim3(256), 128 * 1024, st, p);
    else hipLaunchKernelGGL(attn_prefill_w64_kernel<false>, grid, dim3(256), 128 * 1024, st, p);
#endif
  } else if (hd == 128) {
    const bool ps = A3V_ENV_INT("A3V_ATTN_PSWAP", 1) != 0;          // A3V_ATTN_PSWAP=0: the 8-B-half V^T reads (A/B runs)
    if (causal) {
      if (ps) hipLaunchKernelGGL((attn_prefill_bf16_kernel<128, true, true>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_prefill_bf16_kernel<128, true, false>), grid, dim3(256), 0, st, p);
    } else {
      if (ps) hipLaunchKernelGGL((attn_prefill_bf16_kernel<128, false, true>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_prefill_bf16_kernel<128, false, false>), grid, dim3(256), 0, st, p);
    }
  } else {
    if (causal) hipLaunchKernelGGL((attn_prefill_bf16_kernel<64, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_prefill_bf16_kernel<64, false>), grid, dim3(256), 0, st, p);
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

extern "C" int a3v_attention(const void* q, const void* k, const void* vt, void* out, int B, int Sq, int Sk,
                             int H, int Hkv, int hd, const int64_t* strides, int causal, float* scratch,
                             int dtype, void* stream) {
  return attention_impl(q, k, vt, out, B, Sq, Sk, H, Hkv, hd, strides, causal, scratch, nullptr, dtype, stream);
}

extern "C" int a3v_attention_lse(const void* q, const void* k, const void* vt, void* out, float* lse, int B, int Sq,
                                 int Sk, int H, int Hkv, int hd, const int64_t* strides, int causal, int dtype,
                                 void* stream) {
  if (!lse) return A3V_ERR_ARG;
  return attention_impl(q, k, vt, out, B, Sq, Sk, H, Hkv, hd, strides, causal, nullptr, lse, dtype, stream);
}

// Decode attention with the split combine folded into the same launch (see attn_decode_bf16_kernel); `counters` are
// B*H zero-initialised ints that the kernel leaves zero.  bf16, hd in {64, 128}.
int a3v_attention_decode_fused(const void* q, const void* k, const void* vt, void* out, int B, int Sk, int H, int Hkv, int hd,
                               const int64_t* strides, float* scratch, int* counters, void* stream) {
  if (!q || !k || !vt || !out || !strides || !scratch || !counters || B <= 0 || Sk <= 0 || H <= 0 || Hkv <= 0) return A3V_ERR_ARG;
  if (H % Hkv || (hd != 64 && hd != 128)) return A3V_ERR_SHAPE;
  AttnArgs p;
  p.q = q; p.k = k; p.vt = vt; p.out = out;
  p.q_sb = strides[0]; p.q_ss = strides[1]; p.q_sh = strides[2];
  p.k_sb = strides[3]; p.k_sh = strides[4]; p.k_ss = strides[5];
  p.v_sb = strides[6]; p.v_sh = strides[7]; p.v_sd = strides[8];
  p.o_sb = strides[9]; p.o_ss = strides[10]; p.o_sh = strides[11];
  p.B = B; p.Sq = 1; p.Sk = Sk; p.H = H; p.Hkv = Hkv;
  p.scale = 1.0f / sqrtf((float)hd);
  p.scale_log2 = p.scale * 1.4426950408889634f;
  p.lse = nullptr;
  p.head_group = 1;
  p.staged_o = 0;
  int ns, ch;
  decode_plan(B, H, Sk, &ns, &ch);
  const size_t shm = (size_t)(ch + 8) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const bool nt = A3V_ENV_INT("A3V_ATTN_DECODE_NT", 1) != 0;       // default: non-temporal KV stream (=0 for A/B runs)
  if (A3V_ENV_INT("A3V_ATTN_DECODE_WAVE", 1) != 0) {     // default: the wave-streaming form (=0: the two-phase form)
    // one block per (batch, head) when that alone covers the CUs, else the fewest splits that do (never more than the
    // two-phase plan: the scratch buffer is sized for that one)
    int ns2 = (256 + B * H - 1) / (B * H);
#ifdef A3V_ABLATION
    { const int e = A3V_ENV_INT("A3V_DECODE_WAVE_SPLITS", 0); if (e > 0) ns2 = e; }      // sweeps: splits of the wave-streaming form
#endif
    if (ns2 > ns) ns2 = ns;
    int ch2 = (Sk + ns2 - 1) / ns2;
    ch2 = (ch2 + 63) & ~63;
    ns2 = (Sk + ch2 - 1) / ch2;
    const dim3 grid(ns2, H, B);
    if (hd == 128) {
      if (nt) hipLaunchKernelGGL((attn_decode_wave_kernel<128, true>), grid, dim3(512), 0, st, p, scratch, ns2, ch2, counters);
      else hipLaunchKernelGGL((attn_decode_wave_kernel<128, false>), grid, dim3(512), 0, st, p, scratch, ns2, ch2, counters);
    } else {
      if (nt) hipLaunchKernelGGL((attn_decode_wave_kernel<64, true>), grid, dim3(512), 0, st, p, scratch, ns2, ch2, counters);
      else hipLaunchKernelGGL((attn_decode_wave_kernel<64, false>), grid, dim3(512), 0, st, p, scratch, ns2, ch2, counters);
    }
    A3V_LAUNCH_CHECK();
    return A3V_OK;
  }
  if (hd == 128) {
    if (nt) hipLaunchKernelGGL((attn_decode_bf16_kernel<128, true>), dim3(ns, H, B), dim3(256), shm, st, p, scratch, ns, ch, counters);
    else hipLaunchKernelGGL((attn_decode_bf16_kernel<128, false>), dim3(ns, H, B), dim3(256), shm, st, p, scratch, ns, ch, counters);
  } else {
    if (nt) hipLaunchKernelGGL((attn_decode_bf16_kernel<64, true>), dim3(ns, H, B), dim3(256), shm, st, p, scratch, ns, ch, counters);
    else hipLaunchKernelGGL((attn_decode_bf16_kernel<64, false>), dim3(ns, H, B), dim3(256), shm, st, p, scratch, ns, ch, counters);
  }
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

