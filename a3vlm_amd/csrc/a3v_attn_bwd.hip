// MFMA attention backward for gfx950 (bf16, hd in {64,128}, self-attention Sq == Sk == S).
//
// Same "transposed product" trick as the forward kernel (a3v_attn.hip): with v_mfma_f32_32x32x16 the
// accumulator of  X^T = A . B^T  puts one B-row index on the LANE (col = lane&31) and the A-row index
// on the 16 registers, and those 16 registers are -- in order -- exactly the k-elements a B operand
// wants when the matching A operand rows are read at  base + 4*(lane>>5) + {0..3, 8..11}.
//
//  dQ kernel (grid over q tiles; mirrors the forward):      lane <-> query
//      S^T  = K . Q^T ,  dP^T = V . dO^T            (A = K / V rows from LDS, B = Q / dO in registers)
//      P^T  = exp(S^T*scale - lse[q]) ;  dS^T = P^T o (dP^T - D[q]) * scale     (lane-local)
//      dQ^T += K^T . dS^T                            (A = K^T rows from LDS, B = dS^T from the accumulators)
//  dK/dV kernels (grid over kv tiles; loop over queries and the n_rep query heads; one launch per output):   lane <-> key
//      S    = Q . K^T ,  dP = dO . V^T               (A = Q / dO rows from LDS, B = K / V in registers)
//      P, dS with lse[q], D[q] per REGISTER row (read from LDS)
//      dV^T += dO^T . P ,  dK^T += Q^T . dS          (A = dO^T / Q^T fragments by transpose reads of the dO / Q row tiles, B = P / dS)
// No transposed copies of K, Q or dO exist (ds_read_b64_tr_b16 delivers the transposed MFMA operands); no atomics anywhere.
#include "a3v_common.h"
#include <type_traits>

namespace {

struct BwdArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dout;
  const float* lse; const float* D;                            // [B,H,S], [B,S,H]
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  // packed mode (dqkv != nullptr): the three kernels write straight into the gradient of the fused qkv activation
  // [B*S, (H + 2 Hkv) * hd], q and k parts rotated back by -theta (the backward of apply_rotary_emb), instead of dq / dk / dv
  bf16_t* dqkv; int64_t ld_qkv; const float* cos_sin; int rope_pos0;
  int64_t k_sb, k_sh, v_sb, v_ss, v_sh;
  int B, S, Sp, H, Hkv, causal;
  float scale;
  int group_q, group_kv;     // causal: heads per tile-rank-major group of the block order (dQ kernel / dK, dV kernels); 1 = head-major
#ifdef AB_STAMP
  unsigned long long* stamps;   // timing experiment (make EXTRA=-DAB_STAMP=<block>): s_memtime stamps of that block's wave 0, [kernel][64 iterations][8]
#endif
};
#ifdef AB_STAMP
#define AB_ST(kern, it, k) do { if (stamps && (it) < 64) stamps[((kern) * 64 + (it)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AB_ST(kern, it, k) do {} while (0)
#endif

// 16-B chunk swizzle of a row tile: logical chunk c of row r sits at physical chunk c ^ tile_swz(r).  Two access patterns must
// both be conflict-free: (a) row fragments, ds_read_b128 by 16 lanes on 16 consecutive rows at one logical chunk -> the swizzle
// must be a bijection on 16 consecutive rows; (b) transpose reads (ds_read_b64_tr_b16), where one half-wave reads 64 B (four
// chunks) of each of FOUR consecutive rows -> the four rows must differ in the upper two chunk bits.  HD 128 (16 chunks = one
// 256-B bank row per tile row): the row's two low bits go to the upper chunk bits and bits 2-3 to the lower ones.  With the
// plain `r & 15` the four rows of (b) shared their 64 banks: 4-way conflicts on every transpose read (17.7 M conflict cycles per
// launch in profiles/r01t_pmc_attention.txt, ~6 extra cycles per read).
template <int HD>
__device__ __forceinline__ int tile_swz(int row) {
  return (HD == 128) ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row >> 1) & 7);
}

// Tile staging: a 64-row x HD tile of row-major bf16 rows goes global -> LDS by LDS-DMA (16-B chunk XOR swizzle of the forward K
// tile: logical chunk ch of row r sits at physical chunk ch ^ swizzle(r)) without passing through VGPRs; it has landed after
// `s_waitcnt vmcnt(0)` + a workgroup barrier.  One wave instruction writes 1 KiB (lane-linear), so the chunk swizzle is applied on
// the SOURCE side: thread id -> (row id / KCH, physical chunk id % KCH) fetches logical chunk (physical ^ swizzle(row)).  Buffer-
// descriptor form: the per-lane byte offsets (dma_lane_offsets) are loop constants shared by every tile with the same row stride;
// the tile's first row goes into the descriptor base and the bytes left up to the end of the last valid row into num_records, so
// rows past the end read as zeros (their scores are masked).
__device__ __forceinline__ void buf_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ void buf_dma4(__amdgpu_buffer_rsrc_t rs, void* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 4, voff, 0, 0, 0);
}
// Per-lane byte offset of DMA instruction 0 of a tile (thread id -> row id / KCH, physical chunk id % KCH, source chunk = physical ^
// swizzle(row)).  Instruction i covers the rows 256 / KCH further down (16 at HD 128, 32 at HD 64): tile_swz only looks at row bits
// below that, so its offset is this one + i * dma_step() -- ONE live VGPR per row stride instead of 64 KCH / 256 of them (the dK
// kernel, which has no VGPR to spare, used to recompute the whole set with ~60 VALU instructions per loop iteration).
template <int HD>
__device__ __forceinline__ unsigned dma_lane_offset0(int64_t row_stride, int tid) {
  constexpr int KCH = HD / 8;
  const int row = tid / KCH, pc = tid % KCH;
  return (unsigned)((row * row_stride + ((pc ^ tile_swz<HD>(row)) << 3)) * 2);
}
template <int HD>
__device__ __forceinline__ unsigned dma_step(int64_t row_stride) { return (unsigned)((256 / (HD / 8)) * row_stride * 2); }
// rows [row0, row0 + 64) of a [nrows, HD] matrix with `row_stride` elements between rows
template <int HD>
__device__ __forceinline__ void stage_rows_dma(const bf16_t* base, int64_t row_stride, int row0, int nrows, char* lds, unsigned voff0, int wave) {
  const int64_t left = ((int64_t)(nrows - 1 - row0) * row_stride + HD) * 2;          // bytes from the tile's first row to the end of the last valid row
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)row0 * row_stride), 0, (int)(left < 0x7fffffff ? left : 0x7fffffff), 0x00020000);
  const unsigned step = dma_step<HD>(row_stride);
#pragma unroll
  for (int i = 0; i < 64 * (HD / 8) / 256; ++i) buf_dma16(rs, lds + (wave * 64 + i * 256) * 16, voff0 + i * step);
}

// A operand fragment of a row tile: row (32 tb + lane&31), 16-B chunk 2 ks + hh.  The chunk swizzle only looks at the row's low four
// bits and (2 ks | hh) ^ sw == (2 ks) ^ (hh ^ sw), so the address is  tile + 32 tb ROWB + (frag_rows_base ^ (ks << 5)):  one per-lane
// constant, one v_xor per fragment (the row / swizzle arithmetic per fragment was a third of the loops' VALU instructions).
template <int HD>
__device__ __forceinline__ int frag_rows_base(int ql, int hh) { return ql * (HD * 2) + ((hh ^ tile_swz<HD>(ql)) << 4); }
template <int HD>
__device__ __forceinline__ bf16x8 frag_rows(const char* lds, int fbase, int tb, int ks) {
  return *reinterpret_cast<const bf16x8*>(lds + tb * 32 * (HD * 2) + (fbase ^ (ks << 5)));
}

// A operand of the accumulator-order contractions (dQ^T += K^T dS^T, dV^T += dO^T P^T, dK^T += Q^T dS^T), taken from the ROW tile
// instead of a transposed copy: element e of lane (ql, hh) is
// T[kv = 32 tb + 16 c + 4 hh + (e & 3) + 8 (e >> 2)][d = 32 db + ql], i.e. four consecutive rows of one column -- what gfx950's
// ds_read_b64_tr_b16 delivers: the 16 lanes of a group pass the addresses of a 4 x 16 block (lane i: row i >> 2, columns
// 4 (i & 3)..+3) and lane i receives column i (tools/ubench/trread.hip).  Rows r..r+3 carry different chunk swizzles, so the four
// 32-B row segments of one read fall on different banks.  Addressing as for frag_rows: the two row groups' (rows r, r + 8) per-lane
// bases for db = tb = c = 0 are computed once (tr_bases); block (db, tb, c) is  tile + (32 tb + 16 c) ROWB + (base ^ (db << 6)).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
struct TrBase { int lo, hi; };
template <int HD>
__device__ __forceinline__ TrBase tr_bases(int lane) {
  const int hh = lane >> 5, ql = lane & 31, i = lane & 15;
  const int chunk16 = 2 * (ql >> 4) + ((i & 3) >> 1);
  const int row0 = 4 * hh + (i >> 2), row1 = row0 + 8;
  TrBase b;
  b.lo = row0 * (HD * 2) + ((chunk16 ^ tile_swz<HD>(row0)) << 4) + (i & 1) * 8;
  b.hi = row1 * (HD * 2) + ((chunk16 ^ tile_swz<HD>(row1)) << 4) + (i & 1) * 8;
  return b;
}
template <int HD>
__device__ __forceinline__ bf16x8 frag_rows_tr(const char* lds, TrBase tb0, int db, int tb, int c) {
  const char* blk = lds + (32 * tb + 16 * c) * (HD * 2);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(blk + (tb0.lo ^ (db << 6))));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(blk + (tb0.hi ^ (db << 6))));
  bf16x8 f;
  const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  __builtin_memcpy(&f, v, 16);
  return f;
}

// Read back a wave's 32 x HD bf16 tile from its LDS patch (8-byte slot s of row r at slot s ^ ((r & (HD/8 - 1)) << 1), written by the
// lane that owns row r) and store it as whole rows, 16 bytes per lane: rows row0 .. row0 + 31 of a matrix with `ld` elements per
// row, rows >= nrows skipped.  (Per-lane row stores put 16 bytes into each of 32 rows per instruction: tools/ubench/stores.hip.)
template <int HD>
__device__ __forceinline__ void store_patch_rows(const char* patch, bf16_t* base, int64_t ld, int row0, int nrows, int lane) {
  constexpr int ROWB = HD * 2, NPAIR = HD / 8, RPI = 64 / NPAIR;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int pr = lane % NPAIR, rr = lane / NPAIR;
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int r = it * RPI + rr;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + r * ROWB + ((pr ^ (r & (NPAIR - 1))) << 4));
    if (row0 + r < nrows) *reinterpret_cast<bf16x8*>(base + (int64_t)(row0 + r) * ld + pr * 8) = v;
  }
}

// Store one [32 x HD] accumulator set (lane: one row, 4 consecutive d at 32 db + 8 g4 + 4 hh): as a plain [.., HD] row at `plain`,
// or (packed mode) into the fused-qkv gradient row `prow` at column slot * HD, with the inverse rotary rotation of position pos
// when `rotate`.  The mode is a template parameter of the kernels: with a run-time test (or per-store pointer selects) the dK
// kernel went from 240 VGPRs to 256 + 18 spills inside its query loop (two waves per SIMD leave no slack).
template <int HD, bool PACKED>
__device__ __forceinline__ void store_grad_rows(const BwdArgs& p, f32x16 (&acc)[HD / 32], bf16_t* plain, int64_t prow, int slot, int pos,
                                                bool rotate, int hh) {
  if constexpr (!PACKED) {
    bf16_t* O = plain;
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f2bf(acc[d][g4 * 4 + e]);
        *reinterpret_cast<bf16x4*>(O + d * 32 + g4 * 8 + hh * 4) = ov;
      }
    return;
  }
  bf16_t* O = p.dqkv + prow * p.ld_qkv + (int64_t)slot * HD;
  const float* CS = p.cos_sin + (int64_t)pos * HD;             // [pos][HD / 2][cos, sin]
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dc = d * 32 + g4 * 8 + hh * 4;
      float v0 = acc[d][g4 * 4 + 0], v1 = acc[d][g4 * 4 + 1], v2 = acc[d][g4 * 4 + 2], v3 = acc[d][g4 * 4 + 3];
      if (rotate) {        // dx = R(-theta) dy on the (even, odd) pairs (LLM/llama_ens5.py:123-135 backward)
        const f32x4 cs = *reinterpret_cast<const f32x4*>(CS + dc);
        const float a0 = v0 * cs[0] + v1 * cs[1], a1 = -v0 * cs[1] + v1 * cs[0];
        const float a2 = v2 * cs[2] + v3 * cs[3], a3 = -v2 * cs[3] + v3 * cs[2];
        v0 = a0; v1 = a1; v2 = a2; v3 = a3;
      }
      bf16x4 ov;
      ov[0] = f2bf(v0); ov[1] = f2bf(v1); ov[2] = f2bf(v2); ov[3] = f2bf(v3);
      *reinterpret_cast<bf16x4*>(O + dc) = ov;
    }
}

// ------------------------------------------------------------------ dQ
// 1-D grid -> (tile, head slot) with the XCD-aware order of attn_prefill_bf16_kernel: workgroup id & 7 is the XCD, and each XCD
// takes a contiguous range of (head, tile) pairs so the K / V / Q / dO tiles a head's blocks share stay in one L2.
// G > 1 (causal): groups of G heads are walked tile-rank-major -- every head's heaviest tile, then every head's second, ... -- so
// that the long blocks of an XCD's last heads do not start late and run alone (see attn_prefill_bf16_kernel).
__device__ __forceinline__ void xcd_head_tile(int nt, int& head_slot, int& t, int G = 1) {
  const int total = gridDim.x, id = blockIdx.x;
  const int xcd = id & 7, q = total >> 3, r = total & 7;
  const int vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  if (G > 1) {
    const int per = G * nt, grp = vb / per, rr = vb - grp * per;
    head_slot = grp * G + rr % G;
    t = rr / G;
    return;
  }
  head_slot = vb / nt;
  t = vb - head_slot * nt;
}

template <int HD, bool PACKED>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(BwdArgs p) {
  constexpr int TILE = 64 * HD * 2;                 // bytes of one 64-row tile (== HD x 128 B)
  // two (K, V) tile pairs: tile t+1 streams in by LDS-DMA while tile t is consumed (one barrier per tile; the synchronous
  // global -> VGPR -> LDS staging was ~16 % of the kernel, profiles/r01l_attn_bwd_staging_variants.txt)
  __shared__ __attribute__((aligned(1024))) char lds[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqt = (p.S + 127) / 128;
  int head_slot, ti;
  xcd_head_tile(nqt, head_slot, ti, p.group_q);
  const int qt = nqt - 1 - ti, b = head_slot / p.H, h = head_slot - b * p.H;     // heavy causal tiles first within a head
  const int hk = h / (p.H / p.Hkv);
  const int ql = lane & 31, hh = lane >> 5;
  const int qrow = qt * 128 + wave * 32 + ql;
  const int qc = qrow < p.S ? qrow : p.S - 1;
  const bf16_t* Q = p.q + (((int64_t)b * p.S + qc) * p.H + h) * HD;
  const bf16_t* DO = p.dout + (((int64_t)b * p.S + qc) * p.H + h) * HD;
  const bf16_t* K = p.k + b * p.k_sb + hk * p.k_sh;
  const bf16_t* V = p.v + b * p.v_sb + hk * p.v_sh;
  bf16x8 qf[HD / 16], dof[HD / 16];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(Q + ks * 16 + hh * 8);
    dof[ks] = *reinterpret_cast<const bf16x8*>(DO + ks * 16 + hh * 8);
  }
  const float lse2 = p.lse[((int64_t)b * p.H + h) * p.S + qc] * 1.4426950408889634f;
  const float sl2 = p.scale * 1.4426950408889634f;
  const float Dq = p.D[((int64_t)b * p.S + qc) * p.H + h];
  f32x16 acc[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  int kv_end = p.S;
  if (p.causal) kv_end = min(p.S, min(qt * 128 + 127, p.S - 1) + 1);
  const int n_tiles = (kv_end + 63) / 64;
  const unsigned koff = dma_lane_offset0<HD>(HD, tid), voff = dma_lane_offset0<HD>(p.v_ss, tid);
  const int fbase = frag_rows_base<HD>(ql, hh);
  const TrBase trb = tr_bases<HD>(lane);
  stage_rows_dma<HD>(K, HD, 0, p.S, lds, koff, wave);
  stage_rows_dma<HD>(V, p.v_ss, 0, p.S, lds + TILE, voff, wave);
#ifdef AB_STAMP
  unsigned long long* stamps = (p.stamps && blockIdx.x == AB_STAMP && tid == 0) ? p.stamps : nullptr;
#endif
  auto body = [&](int t, auto bufc) {             // unrolled by two, compile-time buffer index (see the dK / dV kernel)
    constexpr int BUF = decltype(bufc)::value;
    AB_ST(0, t, 0);
    const int kv0 = t * 64;
    const char* Ks = lds + BUF * 2 * TILE;        // dQ^T += K^T . dS^T reads K^T fragments out of the K row tile (transpose reads)
    const char* Vs = Ks + TILE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AB_ST(0, t, 1);
    __syncthreads();                              // tile t has landed; every wave is done with the other pair (tile t - 1)
    AB_ST(0, t, 2);
    if (t + 1 < n_tiles) {
      char* nx = lds + (1 - BUF) * 2 * TILE;
      stage_rows_dma<HD>(K, HD, kv0 + 64, p.S, nx, koff, wave);
      stage_rows_dma<HD>(V, p.v_ss, kv0 + 64, p.S, nx + TILE, voff, wave);
    }
    AB_ST(0, t, 3);
    if (p.causal && kv0 > qt * 128 + wave * 32 + 31) return;     // every key of the tile is past this wave's last query row: nothing to add
    // one 32-key block at a time: S^T and dP^T accumulators (32 VGPRs) are dead before the next block starts, which keeps
    // the kernel under 256 VGPRs = two waves per SIMD
    bf16x8 dsf[2][2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Ks, fbase, tb, ks), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Vs, fbase, tb, ks), dof[ks], dp, 0, 0, 0);
      }
      // P = exp2(s * scale*log2e - lse*log2e) with the raw v_exp_f32; interior tiles (all 64 keys visible to all 32 query
      // rows of the wave) skip the compare/select per element
      const bool need_mask = (kv0 + 64 > p.S) || (p.causal && kv0 + 63 > qt * 128 + wave * 32);
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool ok = (kv < p.S) && (!p.causal || kv <= qrow);
          const float ev = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lse2));      // unconditional + select: no exec branch per element
          dsf[tb][r >> 3][r & 7] = f2bf(ok ? ev * (dp[r] - Dq) : 0.f);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lse2));
          dsf[tb][r >> 3][r & 7] = f2bf(pr * (dp[r] - Dq));
        }
      }
    }
    AB_ST(0, t, 4);
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int d = 0; d < HD / 32; ++d)      // the d blocks' chains interleaved: back-to-back MFMAs on one accumulator issue at ~72 cycles, not 32
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_tr<HD>(Ks, trb, d, tb, c), dsf[tb][c], acc[d], 0, 0, 0);
    AB_ST(0, t, 5);
  };
  {
    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
      body(t, std::integral_constant<int, 0>{});
      body(t + 1, std::integral_constant<int, 1>{});
    }
    if (t < n_tiles) body(t, std::integral_constant<int, 0>{});
  }
  // dS was formed WITHOUT the 1 / sqrt(hd) factor (dS = P o (dP - D), one multiply less per score): it is applied to the sums here
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] *= p.scale;
  int q_e = qrow, hh_e = hh;                 // opaque copies: keeps the store's address arithmetic out of the key loop (see dK below)
  asm volatile("" : "+v"(q_e), "+v"(hh_e));
  if constexpr (PACKED) {
    // the wave's 32 x HD tile of dq, rotated back, through a private LDS patch (the K / V tiles are free after one more barrier)
    // into whole rows of the fused-qkv gradient (see store_patch_rows)
    __syncthreads();
    int ql_e = q_e - (qt * 128 + wave * 32);
    char* patch = lds + wave * (32 * HD * 2);
    char* wrow = patch + ql_e * (HD * 2);
    const int wx = (ql_e & (HD / 8 - 1)) << 1;
    const float* CS = p.cos_sin + (int64_t)(p.rope_pos0 + (q_e < p.S ? q_e : p.S - 1)) * HD;
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int dc = d * 32 + g4 * 8 + hh_e * 4;
        const f32x4 cs = *reinterpret_cast<const f32x4*>(CS + dc);
        const float v0 = acc[d][g4 * 4 + 0], v1 = acc[d][g4 * 4 + 1], v2 = acc[d][g4 * 4 + 2], v3 = acc[d][g4 * 4 + 3];
        bf16x4 ov;
        ov[0] = f2bf(v0 * cs[0] + v1 * cs[1]); ov[1] = f2bf(-v0 * cs[1] + v1 * cs[0]);
        ov[2] = f2bf(v2 * cs[2] + v3 * cs[3]); ov[3] = f2bf(-v2 * cs[3] + v3 * cs[2]);
        *reinterpret_cast<bf16x4*>(wrow + (((d * 8 + g4 * 2 + hh_e) ^ wx) << 3)) = ov;
      }
    store_patch_rows<HD>(patch, p.dqkv + (int64_t)b * p.S * p.ld_qkv + (int64_t)h * HD, p.ld_qkv, qt * 128 + wave * 32, p.S, lane);
  } else {
    if (q_e < p.S)
      store_grad_rows<HD, PACKED>(p, acc, p.dq + (((int64_t)b * p.S + q_e) * p.H + h) * HD, (int64_t)b * p.S + q_e, h, p.rope_pos0 + q_e, true, hh_e);
  }
}

// ------------------------------------------------------------------ dK, dV
// Two specialisations of one kernel, launched back to back: WHICH == 0 accumulates dV (needs S only), WHICH == 1
// accumulates dK (needs S and dP).  Each keeps ONE [HD x 32-key] accumulator set (64 VGPRs) instead of two, which brings
// the kernel under 256 VGPRs = two waves per SIMD and cuts the LDS tiles staged per query tile from four to two / three;
// the price is one extra S^T recompute (5 instead of 4 tile products), paid back ~1.5x by the doubled occupancy.
template <int HD, int WHICH, bool PACKED>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(BwdArgs p) {
  constexpr int TILE = 64 * HD * 2;
  // two (Q, dO) tile pairs + their lse / D rows: pair i+1 streams in by LDS-DMA while pair i is consumed (see the dQ kernel)
  __shared__ __attribute__((aligned(1024))) char lds[4 * TILE + 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int head_slot, kt_;
  xcd_head_tile((p.S + 127) / 128, head_slot, kt_, p.group_kv);
  const int b = head_slot / p.Hkv, hk = head_slot - b * p.Hkv;
  const int nrep = p.H / p.Hkv;
  const int kl = lane & 31, hh = lane >> 5;
  const int kvrow = kt_ * 128 + wave * 32 + kl;
  const int kc = kvrow < p.S ? kvrow : p.S - 1;
  const bf16_t* Kr = p.k + b * p.k_sb + hk * p.k_sh + (int64_t)kc * HD;
  const bf16_t* Vr = p.v + b * p.v_sb + hk * p.v_sh + (int64_t)kc * p.v_ss;
  bf16x8 kf[HD / 16], vf[WHICH == 1 ? HD / 16 : 1];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(Kr + ks * 16 + hh * 8);
    if (WHICH == 1) vf[ks] = *reinterpret_cast<const bf16x8*>(Vr + ks * 16 + hh * 8);
  }
  f32x16 acc[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  const float sl2 = p.scale * 1.4426950408889634f;
  const int q_begin = p.causal ? (kt_ * 128) / 64 : 0;      // first q tile that can see this block's keys
  const int n_qt = (p.S + 63) / 64;
  const int n_it = n_qt - q_begin, total = nrep * n_it;
  // pair `it` = (head repetition it / n_it, query tile q_begin + it % n_it) -> buffer it & 1:
  //   [Q rows | dO rows] (S^T recompute from row fragments; Q^T for dK / dO^T for dV by transpose reads) + lse[64] | D[64]
  // (the pair indices are carried as counters: two integer divisions per iteration were ~50 instructions)
  const unsigned qoff = dma_lane_offset0<HD>((int64_t)p.H * HD, tid);          // Q and dO rows share the row stride H*HD
  const int fbase = frag_rows_base<HD>(kl, hh);
  const TrBase trb = tr_bases<HD>(lane);
  int rep_n = 0, qi_n = 0;                       // (head repetition, query tile index) of the next pair to issue
  auto issue = [&](int it) {
    const int h = hk * nrep + rep_n, q0 = (q_begin + qi_n) * 64;
    if (++qi_n == n_it) { qi_n = 0; ++rep_n; }
    char* buf = lds + (it & 1) * 2 * TILE;
    stage_rows_dma<HD>(p.q + ((int64_t)b * p.S * p.H + h) * HD, (int64_t)p.H * HD, q0, p.S, buf, qoff, wave);
    stage_rows_dma<HD>(p.dout + ((int64_t)b * p.S * p.H + h) * HD, (int64_t)p.H * HD, q0, p.S, buf + TILE, qoff, wave);
    float* sm = reinterpret_cast<float*>(lds + 4 * TILE) + (it & 1) * 128;
    if (wave == 0) {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lse + ((int64_t)b * p.H + h) * p.S + q0), 0, (p.S - q0) * 4, 0x00020000);
      buf_dma4(rs, sm, lane * 4);
    }
    if (WHICH == 1 && wave == 1) {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.D + ((int64_t)b * p.S + q0) * p.H + h), 0, ((p.S - 1 - q0) * p.H + 1) * 4, 0x00020000);
      buf_dma4(rs, sm + 64, lane * p.H * 4);
    }
  };
  if (total > 0) issue(0);
#ifdef AB_STAMP
  unsigned long long* stamps = (p.stamps && blockIdx.x == AB_STAMP && tid == 0) ? p.stamps : nullptr;
#endif
  int qi_c = 0;                                  // query tile index of the pair being consumed
  // The loop is unrolled by two with the buffer index a compile-time constant: every LDS read address is then (per-lane base ^ constant)
  // + an instruction offset.  With a run-time buffer base each of the 40 - 64 fragment reads of an iteration paid its own v_or / v_add.
  auto body = [&](int it, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    AB_ST(1 + WHICH, it, 0);
    {
      const int q0 = (q_begin + qi_c) * 64;
      if (++qi_c == n_it) qi_c = 0;
      const char* Qs = lds + BUF * 2 * TILE;
      const char* T1 = Qs + TILE;
      const float* lse_s = reinterpret_cast<const float*>(lds + 4 * TILE) + BUF * 128;
      const float* D_s = lse_s + 64;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      AB_ST(1 + WHICH, it, 1);
      __syncthreads();                       // pair `it` has landed; every wave is done with the other buffer (pair it - 1)
      AB_ST(1 + WHICH, it, 2);
      if (it + 1 < total) issue(it + 1);
      AB_ST(1 + WHICH, it, 3);
      if (p.causal && q0 + 63 < kt_ * 128 + wave * 32) return;            // every query row of the tile is before this wave's first key
      bf16x8 bf[2][2];                       // P (dV) or dS (dK) as the B operand of the accumulation products
      const int kv_hi = kt_ * 128 + wave * 32 + 31;                        // last key of this wave (wave-uniform)
      const bool interior = (q0 + 64 <= p.S) && (kv_hi < p.S) && (!p.causal || q0 >= kv_hi);
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Qs, fbase, tb, ks), kf[ks], s, 0, 0, 0);
          if (WHICH == 1) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(T1, fbase, tb, ks), vf[ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int qb = tb * 32 + 8 * g4 + 4 * hh;                       // 4 consecutive query rows of the tile
          f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qb);
#pragma unroll
          for (int e = 0; e < 4; ++e) l4[e] *= 1.4426950408889634f;       // lse * log2(e)
          f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
          if (WHICH == 1) d4 = *reinterpret_cast<const f32x4*>(D_s + qb);
          // interior pairs (all 64 query rows exist and see all 32 keys of this wave): no compare / select per element; on
          // the others the exponential is evaluated unconditionally and selected (an `ok ? exp : 0` form compiles to one
          // exec-mask branch per element: 32 per tile)
          if (interior) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = g4 * 4 + e;
              const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -l4[e]));
              bf[tb][r >> 3][r & 7] = WHICH == 0 ? f2bf(pr) : f2bf(pr * (dp[r] - d4[e]));
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = g4 * 4 + e;
              const int qg = q0 + qb + e;
              const bool ok = (qg < p.S) && (kvrow < p.S) && (!p.causal || kvrow <= qg);
              const float ev = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -l4[e]));
              const float val = WHICH == 0 ? ev : ev * (dp[r] - d4[e]);
              bf[tb][r >> 3][r & 7] = f2bf(ok ? val : 0.f);
            }
          }
        }
      }
      AB_ST(1 + WHICH, it, 4);
      const char* At = WHICH == 0 ? T1 : Qs;      // dV^T += dO^T . P^T ; dK^T += Q^T . dS^T : transposed operands read out of the row tiles
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int d = 0; d < HD / 32; ++d)    // (chains interleaved, see the dQ kernel)
            acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_tr<HD>(At, trb, d, tb, c), bf[tb][c], acc[d], 0, 0, 0);
      AB_ST(1 + WHICH, it, 5);
    }
  };
  {
    int it = 0;
    for (; it + 1 < total; it += 2) {
      body(it, std::integral_constant<int, 0>{});
      body(it + 1, std::integral_constant<int, 1>{});
    }
    if (it < total) body(it, std::integral_constant<int, 0>{});
  }
  if (WHICH == 1) {        // dS was formed without the 1 / sqrt(hd) factor (see the dQ kernel)
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][r] *= p.scale;
  }
  // (the stores are written out here rather than through store_grad_rows: routed through the helper, this kernel went from
  //  240 VGPRs to 256 + 18 spills inside the query loop -- two waves per SIMD leave the allocator no slack)
  if (kvrow < p.S) {
    if constexpr (!PACKED) {
      bf16_t* O = (WHICH == 0 ? p.dv : p.dk) + (((int64_t)b * p.Hkv + hk) * p.S + kvrow) * HD;
#pragma unroll
      for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          bf16x4 a;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = f2bf(acc[d][g4 * 4 + e]);
          *reinterpret_cast<bf16x4*>(O + d * 32 + g4 * 8 + hh * 4) = a;
        }
    }
  }
  if constexpr (PACKED) {
    // packed mode: the wave's 32 x HD tile goes through a private LDS patch (the row tiles are free after one more barrier) and
    // leaves as whole rows of the fused-qkv gradient
    __syncthreads();
    char* patch = lds + wave * (32 * HD * 2);
    char* wrow = patch + kl * (HD * 2);
    const int wx = (kl & (HD / 8 - 1)) << 1;
    const float* CS = p.cos_sin + (int64_t)(p.rope_pos0 + kc) * HD;          // [pos][HD / 2][cos, sin] (kc: the clamped row)
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int dc = d * 32 + g4 * 8 + hh * 4;
        float v0 = acc[d][g4 * 4 + 0], v1 = acc[d][g4 * 4 + 1], v2 = acc[d][g4 * 4 + 2], v3 = acc[d][g4 * 4 + 3];
        if (WHICH == 1) {      // dK: rotate back, dx = R(-theta) dy on the (even, odd) pairs (LLM/llama_ens5.py:123-135 backward)
          const f32x4 cs = *reinterpret_cast<const f32x4*>(CS + dc);
          const float a0 = v0 * cs[0] + v1 * cs[1], a1 = -v0 * cs[1] + v1 * cs[0];
          const float a2 = v2 * cs[2] + v3 * cs[3], a3 = -v2 * cs[3] + v3 * cs[2];
          v0 = a0; v1 = a1; v2 = a2; v3 = a3;
        }
        bf16x4 ov;
        ov[0] = f2bf(v0); ov[1] = f2bf(v1); ov[2] = f2bf(v2); ov[3] = f2bf(v3);
        *reinterpret_cast<bf16x4*>(wrow + (((d * 8 + g4 * 2 + hh) ^ wx) << 3)) = ov;
      }
    store_patch_rows<HD>(patch, p.dqkv + (int64_t)b * p.S * p.ld_qkv + (int64_t)(WHICH == 0 ? p.H + p.Hkv + hk : p.H + hk) * HD, p.ld_qkv,
                         kt_ * 128 + wave * 32, p.S, lane);
  }
}

}  // namespace

// declared in a3v_train.hip
extern "C" int a3v_transpose(const void* src, int64_t ld_src, int64_t bs_src, void* dst, int64_t ld_dst, int64_t bs_dst,
                             int R, int C, int Rpad, int batch, int dtype, void* stream);

// The MFMA path used to keep K^T / Q^T / dO^T images here; since the kernels read their transposed operands out of the row
// tiles it needs no scratch.  A non-NULL `workspace` still selects the MFMA path (NULL = generic kernels): 256 bytes suffice.
extern "C" int64_t a3v_attention_bwd_workspace_bytes(int B, int S, int H, int Hkv, int hd) {
  (void)B; (void)S; (void)H; (void)Hkv; (void)hd;
  return 256;
}

#ifdef AB_STAMP
static unsigned long long* g_ab_stamps = nullptr;
extern "C" void a3v_debug_set_bwd_stamps(void* ptr) { g_ab_stamps = (unsigned long long*)ptr; }   // >= 3 * 64 * 8 * 8 bytes, zeroed
#endif
// the three MFMA kernels; D must already hold rowsum(dO o O)
static int attention_bwd_mfma_impl(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb,
                                   int64_t v_ss, int64_t v_sh, const void* dout, const float* lse, const float* D, void* dq,
                                   void* dk, void* dv, void* dqkv, int64_t ld_qkv, const float* cos_sin, int rope_pos0, int B, int S, int H,
                                   int Hkv, int hd, int causal, void* stream) {
  if (hd != 64 && hd != 128) return A3V_ERR_ARG;
  if ((k_sb % 8) || (k_sh % 8) || (v_sb % 8) || (v_ss % 8) || (v_sh % 8)) return A3V_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int Sp = (S + 63) / 64 * 64;
  // (no transposed images: all three kernels take their transposed MFMA operands from the row tiles with ds_read_b64_tr_b16;
  //  the workspace argument is kept for ABI stability and not touched)
  BwdArgs p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.dout = (const bf16_t*)dout;
  p.lse = lse; p.D = D;
  p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
  p.dqkv = (bf16_t*)dqkv; p.ld_qkv = ld_qkv; p.cos_sin = cos_sin; p.rope_pos0 = rope_pos0;
  p.k_sb = k_sb; p.k_sh = k_sh; p.v_sb = v_sb; p.v_ss = v_ss; p.v_sh = v_sh;
  p.B = B; p.S = S; p.Sp = Sp; p.H = H; p.Hkv = Hkv; p.causal = causal;
  p.scale = 1.0f / sqrtf((float)hd);
#ifdef AB_STAMP
  p.stamps = g_ab_stamps;
#endif
  dim3 gq(((S + 127) / 128) * H * B), gk(((S + 127) / 128) * Hkv * B);
  auto group_of = [&](int heads, unsigned blocks) {         // as in the forward (a3v_attn.hip), with Q + dO + K + V per head: 8 heads at S = 1091 (592.8 -> 550.2 us for the three kernels)
    if (!causal || (blocks & 7) || (heads & 7)) return 1;
    const int ge = A3V_ENV_INT("A3V_ATTN_HEAD_GROUP", 0);
    int want = 16;
    while (want > 1 && (int64_t)want * S * hd * 4 > (9 << 19)) want >>= 1;
    if (ge > 0) want = ge;
    int G = want < 1 ? 1 : want;
    while (G > 1 && (heads / 8) % G) G >>= 1;
    return G;
  };
  p.group_q = group_of(B * H, gq.x);
  p.group_kv = group_of(B * Hkv, gk.x);
#define A3V_BWD_LAUNCH(HDV, PK)                                                                  \
  do {                                                                                           \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HDV, PK>), gq, dim3(256), 0, st, p);                  \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 0, PK>), gk, dim3(256), 0, st, p);              \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 1, PK>), gk, dim3(256), 0, st, p);              \
  } while (0)
  if (hd == 128) {
    if (dqkv) A3V_BWD_LAUNCH(128, true); else A3V_BWD_LAUNCH(128, false);
  } else {
    if (dqkv) A3V_BWD_LAUNCH(64, true); else A3V_BWD_LAUNCH(64, false);
  }
#undef A3V_BWD_LAUNCH
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// bf16 MFMA path of a3v_attention_bwd (called from a3v_train.hip); D must already hold rowsum(dO o O).
extern "C" int a3v_attention_bwd_mfma(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb,
                                      int64_t v_ss, int64_t v_sh, const void* dout, const float* lse, const float* D, void* dq,
                                      void* dk, void* dv, void* workspace, int B, int S, int H, int Hkv, int hd, int causal,
                                      void* stream) {
  if (!workspace || !dq || !dk || !dv) return A3V_ERR_ARG;
  return attention_bwd_mfma_impl(q, k, k_sb, k_sh, v, v_sb, v_ss, v_sh, dout, lse, D, dq, dk, dv, nullptr, 0, nullptr, 0, B, S, H, Hkv, hd,
                                 causal, stream);
}

// Packed form: the gradient of the fused qkv activation in one go (dq | dk | dv columns of [B*S, (H + 2 Hkv) * hd], q and k parts
// rotated back), i.e. a3v_attention_bwd + a3v_rope_bwd_pack without the dq / dk / dv round trip.  D as above.
extern "C" int a3v_attention_bwd_mfma_packed(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb,
                                             int64_t v_ss, int64_t v_sh, const void* dout, const float* lse, const float* D, void* dqkv,
                                             int64_t ld_qkv, const float* cos_sin, int rope_pos0, int B, int S, int H, int Hkv, int hd,
                                             int causal, void* stream) {
  if (!dqkv || !cos_sin || ld_qkv % 4 || rope_pos0 < 0) return A3V_ERR_ARG;
  return attention_bwd_mfma_impl(q, k, k_sb, k_sh, v, v_sb, v_ss, v_sh, dout, lse, D, nullptr, nullptr, nullptr, dqkv, ld_qkv, cos_sin, rope_pos0,
                                 B, S, H, Hkv, hd, causal, stream);
}

