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

// ------------------------------------------------------------------ dK + dV in ONE pass (round 4)
// The two kernels above run two waves per SIMD at 254 VGPRs: no room to hold a fragment ahead of its MFMA, so hipcc emits
// `ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma` 48 times per query tile and the matrix pipe waits out an LDS round trip per MFMA
// (the partner wave of an unrelated block is the only cover): 0.14 of the MFMA peak, and S^T is recomputed in each of them.
// Here a block is still 4 waves x 32 keys, but ONE wave per SIMD with the whole 512-register file, both accumulator sets
// (dV^T, dK^T: 128 registers), K and V of the wave's keys (64) and a software pipeline over 32-query UNITS u = (pair, half):
//     step u:   A(u+1)  S = Q K^T, dP = dO V^T of the NEXT unit      16 MFMAs, fragments fetched one step ahead
//             ∥ B(u)    P = exp2(S sl2 - lse), dS = P o (dP - D)      ~100 VALU in the shadow of A's MFMAs
//               C(u)    dV^T += dO^T P, dK^T += Q^T dS                16 MFMAs, transpose-read fragments fetched at the top of the step
//             ∥ fetch the row fragments of A(u+2)
// so every MFMA finds its operands in registers and the only synchronisation is one block barrier per pair (64 queries).
// (Q, dO) pairs stream through a ring of FOUR LDS buffers by LDS-DMA, issued two iterations ahead of their first read.  4 tile products per
// (query, key) tile pair instead of the 5 of the two-launch form; same fragment helpers, same arithmetic per element, same
// per-accumulator MFMA order: the results are bit-identical to attn_bwd_dkv_kernel<.., 0 / 1, ..>.
//
// Register plan.  hipcc places the A / B operands of an MFMA builtin in arch VGPRs only and has no pressure-aware scheduling: with the
// loop-invariant K / V fragments there the loop wants ~300 arch VGPRs, the invariants are spilled (a scratch reload + vmcnt(0) in front of
// every MFMA), and two inlined copies of the step disagree on where the accumulators live (128 v_accvgpr_mov per step).  So the
// accumulator file is OWNED BY HAND, as in the one-wave-per-SIMD attention forward of the programming guide: every MFMA of the loop is an
// asm statement on literal AGPRs
//     a[0:63] dV^T   a[64:127] dK^T   a[128:159] K fragments   a[160:191] V fragments   a[192:223] S, dP of even units   a[224:255] of odd units
// with its A operand (and P / dS) in compiler-allocated VGPRs; S / dP are read back by asm v_accvgpr_read at the top of a step (the
// MFMAs that wrote them are a whole C phase = 16 MFMAs back, far beyond the 18 wait states an XDL result needs) and the compiler never
// sees an AGPR: tests/test_kernel_budget_cpu.py checks that the kernel has no scratch and no compiler-generated v_accvgpr_* (a compiler
// spill into the owned range would be silent corruption).
template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
#define A3V_A16(b) "a" #b
// every owned AGPR, as clobbers of the statement that initialises them (this is also what makes the kernel descriptor allocate them)
#define A3V_CL8(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
#define A3V_AGPR_0_255 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A3V_CL8(1), A3V_CL8(2), A3V_CL8(3), A3V_CL8(4), A3V_CL8(5), A3V_CL8(6), \
    A3V_CL8(7), A3V_CL8(8), A3V_CL8(9), A3V_CL8(10), A3V_CL8(11), A3V_CL8(12), A3V_CL8(13), A3V_CL8(14), A3V_CL8(15), A3V_CL8(16), A3V_CL8(17), \
    A3V_CL8(18), A3V_CL8(19), A3V_CL8(20), A3V_CL8(21), A3V_CL8(22), A3V_CL8(23), A3V_CL8(24), "a250", "a251", "a252", "a253", "a254", "a255"
// acc a[ACC:ACC+15] (+)= A (VGPRs) x B (VGPRs); the s_nop covers a VALU write of A / B right in front (hipcc pads nothing inside asm)
template <int ACC>
__device__ __forceinline__ void mfma_vv(const bf16x8& a, const bf16x8& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(ACC), "i"(ACC + 15) : A3V_AGPR_0_255);
}
// acc a[ACC:ACC+15] (ZERO ? = : +=) A (VGPRs) x a[BR:BR+3]
template <int ACC, int BR, bool ZERO>
__device__ __forceinline__ void mfma_va(const bf16x8& a) {
  if constexpr (ZERO)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, a[%c3:%c4], 0" ::"v"(a), "i"(ACC), "i"(ACC + 15), "i"(BR), "i"(BR + 3) : A3V_AGPR_0_255);
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, a[%c3:%c4], a[%c1:%c2]" ::"v"(a), "i"(ACC), "i"(ACC + 15), "i"(BR), "i"(BR + 3) : A3V_AGPR_0_255);
}
// VGPR accumulator (+)= A (VGPRs) x a[BR:BR+3]: for dP, which the VALU reads element by element (no v_accvgpr_read per element)
template <int BR, bool ZERO>
__device__ __forceinline__ void mfma_va_v(f32x16& acc, const bf16x8& a) {
  if constexpr (ZERO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(acc) : "v"(a), "i"(BR), "i"(BR + 3) : A3V_AGPR_0_255);
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(acc) : "v"(a), "i"(BR), "i"(BR + 3) : A3V_AGPR_0_255);
}
template <int R>
__device__ __forceinline__ float agpr_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(R) : A3V_AGPR_0_255);
  return v;
}
template <int R>
__device__ __forceinline__ void agpr_write(unsigned v) {
  asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "i"(R) : A3V_AGPR_0_255);
}
template <int R0>
__device__ __forceinline__ f32x16 agpr_read16() {
  f32x16 v;
  sfor<0, 16>([&](auto i) { v[decltype(i)::value] = agpr_read<R0 + decltype(i)::value>(); });
  return v;
}

// LDS-DMA as asm statements: hipcc orders every LDS read it cannot prove disjoint from an outstanding `buffer_load .. lds` BUILTIN behind
// `s_waitcnt vmcnt(0)` -- with a ring buffer chosen at run time that is a full memory round trip in front of the first fragment read of every
// step.  The asm form is invisible to that bookkeeping (completion = this kernel's own `s_waitcnt vmcnt` + barrier); M0 (the LDS destination)
// is written in the statement that reads it; `s_nop 4` in front covers descriptor words fresh from a v_readfirstlane.
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__device__ __forceinline__ void dma16_asm(i32x4_t rs, unsigned lds_addr, unsigned voff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", A3V_AGPR_0_255);
}
__device__ __forceinline__ void dma4_asm(i32x4_t rs, unsigned lds_addr, unsigned voff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", A3V_AGPR_0_255);
}
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, int bytes) {
  const uint64_t a = (uint64_t)base;
  i32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000;
  return r;
}
template <int HD>
__device__ __forceinline__ void stage_rows_dma_asm(const bf16_t* base, int64_t row_stride, int row0, int nrows, unsigned lds_addr, unsigned voff0, int wave) {
  int64_t left = ((int64_t)(nrows - 1 - row0) * row_stride + HD) * 2;                // bytes from the tile's first row to the end of the last valid row
  if (left < 0) left = 0;                                                            // the whole tile lies past the last row: zeros, no access
  const i32x4_t rs = make_rsrc(base + (int64_t)(left ? row0 : 0) * row_stride, (int)(left < 0x7fffffff ? left : 0x7fffffff));
  const unsigned step = dma_step<HD>(row_stride);
#pragma unroll
  for (int i = 0; i < 64 * (HD / 8) / 256; ++i) dma16_asm(rs, lds_addr + (wave * 64 + i * 256) * 16, voff0 + i * step);
}
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
template <int HD>
__device__ __forceinline__ bf16x8 frag_rows_tr2(const char* lds, TrBase tb0, int db, int tb, int c) {
  const char* blk = lds + (32 * tb + 16 * c) * (HD * 2);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(blk + (tb0.lo ^ (db << 6))));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(blk + (tb0.hi ^ (db << 6))));
  const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int HD, bool PACKED>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv2_kernel(BwdArgs p) {
  constexpr int TILE = 64 * HD * 2;
  constexpr int BUFB = 2 * TILE + 1024;            // [Q rows | dO rows | lse[64] D[64] ...]
  constexpr int NKS = HD / 16, NDB = HD / 32;
  constexpr int RV = 0, RK = 64, RKF = 128, RVF = 160, RS = 192, RDP = 208;      // the owned AGPR ranges (see above); S / dP: set (unit & 1) at + 32
  constexpr int NB = 4;                            // ring depth: pairs are issued two iterations ahead of their use
  __shared__ __attribute__((aligned(1024))) char lds[NB * BUFB + 4096];    // + a 1-KiB landing patch per wave for the pieces of an iteration that has no pair to fetch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AB_STAMP
  unsigned long long* tl = (p.stamps && tid == 0) ? p.stamps + 4096 + 8 * blockIdx.x : nullptr;
  if (tl) { tl[0] = __builtin_amdgcn_s_memrealtime(); tl[4] = __builtin_amdgcn_s_getreg(0x0004 | (0 << 6) | (31 << 11)); tl[5] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)); }
#endif
  int head_slot, kt_;
  xcd_head_tile((p.S + 127) / 128, head_slot, kt_, p.group_kv);
  const int b = head_slot / p.Hkv, hk = head_slot - b * p.Hkv;
  const int nrep = p.H / p.Hkv;
  const int kl = lane & 31, hh = lane >> 5;
  const int kv_lo = kt_ * 128 + wave * 32;         // wave-uniform: first key of this wave
  const int kvrow = kv_lo + kl;
  const int kc = kvrow < p.S ? kvrow : p.S - 1;
  asm volatile("" ::: A3V_AGPR_0_255);
  const float sl2 = p.scale * 1.4426950408889634f;
  const int q_begin = p.causal ? kt_ * 2 : 0;      // first 64-query tile that can see this block's keys
  const int n_qt = (p.S + 63) / 64;
  const int n_it = n_qt - q_begin, total = nrep * n_it;
  const unsigned qoff = dma_lane_offset0<HD>((int64_t)p.H * HD, tid);
  const int fbase = frag_rows_base<HD>(kl, hh);
  const TrBase trb = tr_bases<HD>(lane);

  int rep_n = 0, qi_n = 0, buf_n = 0;              // (head repetition, query tile, ring buffer) of the next pair to issue
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
  // One pair = PP / 2 pieces of the Q tile + PP / 2 of the dO tile + one 256-byte piece (lse: wave 0, D: wave 1) per wave.  EVERY iteration
  // issues exactly PP + 1 pieces per wave -- an iteration with no pair left to fetch sends them through a zero-length descriptor (no memory
  // request, zeros) into the wave's landing patch -- so "pair i landed" is always "at most (PP + 1) x (pairs issued after i) outstanding", and
  // the pieces are issued ONE PER K-STEP between the MFMAs of an iteration's first step (a burst of nine costs the lone wave ~800 cycles).
  constexpr int PP = HD / 16;
  i32x4_t rsQ, rsO, rsX;
  unsigned dmaQ, dmaO, dmaX, dma_step_i, voffX;
  const unsigned qstep = dma_step<HD>((int64_t)p.H * HD);
  // descriptor state of the next pair to issue, advanced incrementally (the from-scratch form cost ~500 cycles of 64-bit scalar multiplies per pair)
  const int64_t rstride = (int64_t)p.H * HD;
  const uint64_t pairB = (uint64_t)(64 * rstride * 2);                                      // bytes from one 64-query tile to the next (Q, dO)
  uint64_t aQ = (uint64_t)(p.q + ((int64_t)b * p.S * p.H + hk * nrep) * HD + (int64_t)q_begin * 64 * rstride);
  uint64_t aO = (uint64_t)(p.dout + ((int64_t)b * p.S * p.H + hk * nrep) * HD + (int64_t)q_begin * 64 * rstride);
  uint64_t aL = (uint64_t)(p.lse + ((int64_t)b * p.H + hk * nrep) * p.S + q_begin * 64);
  uint64_t aD = (uint64_t)(p.D + ((int64_t)b * p.S + q_begin * 64) * p.H + hk * nrep);
  int64_t leftQ = ((int64_t)(p.S - 1 - q_begin * 64) * rstride + HD) * 2;                    // bytes from the tile's first row to the end of the last valid row
  int rowsL = p.S - q_begin * 64;                                                            // query rows left from the tile's first row
  voffX = (wave == 1) ? lane * p.H * 4 : lane * 4;
  auto mk = [&](uint64_t a, int bytes) __attribute__((always_inline)) {
    i32x4_t r;
    r[0] = (int)(unsigned)a; r[1] = (int)(unsigned)(a >> 32) & 0xffff; r[2] = bytes; r[3] = 0x00020000;
    return r;
  };
  auto prep_issue = [&](bool real) __attribute__((always_inline)) {
    const unsigned buf = lds0 + buf_n * BUFB, patch = lds0 + NB * BUFB + wave * 1024;
    const int nb = real ? (int)(leftQ < 0x7fffffff ? leftQ : 0x7fffffff) : 0;
    rsQ = mk(aQ, nb);
    rsO = mk(aO, nb);
    dmaQ = real ? buf + wave * 1024 : patch;
    dmaO = real ? buf + TILE + wave * 1024 : patch;
    dma_step_i = real ? 4096u : 0u;
    if (wave == 0) { rsX = mk(aL, real ? rowsL * 4 : 0); dmaX = real ? buf + 2 * TILE : patch; }
    else if (wave == 1) { rsX = mk(aD, real ? ((rowsL - 1) * p.H + 1) * 4 : 0); dmaX = real ? buf + 2 * TILE + 256 : patch; }
    else { rsX = mk(aL, 0); dmaX = patch; }
    if (real) {
      if (++buf_n == NB) buf_n = 0;
      if (++qi_n == n_it) {                        // next head repetition of the group: back to the first query tile
        qi_n = 0; ++rep_n;
        const int64_t back = (int64_t)(n_it - 1) * 64;
        aQ += (uint64_t)(HD * 2) - (uint64_t)(back * rstride * 2); aO += (uint64_t)(HD * 2) - (uint64_t)(back * rstride * 2);
        aL += (uint64_t)((int64_t)p.S * 4) - (uint64_t)(back * 4); aD += 4 - (uint64_t)(back * p.H * 4);
        leftQ += back * rstride * 2; rowsL += (int)back;
      } else {
        aQ += pairB; aO += pairB; aL += 256; aD += (uint64_t)(64 * p.H * 4);
        leftQ -= (int64_t)pairB; rowsL -= 64;
      }
    }
  };
  auto piece = [&](auto i_) __attribute__((always_inline)) {
    constexpr int i = decltype(i_)::value;
    if constexpr (i < PP / 2) dma16_asm(rsQ, dmaQ + i * dma_step_i, qoff + i * qstep);
    else if constexpr (i < PP) dma16_asm(rsO, dmaO + (i - PP / 2) * dma_step_i, qoff + (i - PP / 2) * qstep);
    else dma4_asm(rsX, dmaX, voffX);
  };
  auto issue = [&](bool real) __attribute__((always_inline)) {            // the whole pair in one burst (prologue)
    prep_issue(real);
    sfor<0, PP + 1>([&](auto i_) { piece(i_); });
  };

  bf16x8 qa[NKS], da[NKS];                         // row fragments (A operands) of the NEXT unit's S / dP products
  auto fetch = [&](const char* buf, int tb) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qa[ks] = frag_rows<HD>(buf, fbase, tb, ks);
      da[ks] = frag_rows<HD>(buf + TILE, fbase, tb, ks);
    }
  };
  auto dead = [&](int qr0) { return qr0 >= p.S || (p.causal && qr0 + 31 < kv_lo); };        // no (query, key) of the unit is visible
  auto interior = [&](int qr0) { return qr0 + 32 <= p.S && kv_lo + 31 < p.S && (!p.causal || qr0 >= kv_lo + 31); };

#ifdef AB_STAMP
  unsigned long long* stamps = (p.stamps && blockIdx.x == AB_STAMP && tid == 0) ? p.stamps : nullptr;
  int cur_it = 0;
#endif
  // One pipeline step.  DA: run A (S, dP of the next unit, from qa / da).  DBC: run B and C of the current unit (its pair in `cb`, half
  // `tbc`, first query row qr0).  MASK: the unit has invisible (query, key) pairs.  Then fetch the fragments of the unit after next.
  // Instruction ORDER is the whole game for a lone wave: it issues in order, so an MFMA behind an MFMA stalls the wave until the pipe takes
  // it (32 cycles) and nothing else issues meanwhile, and an LDS read costs its share of the CU's 256 B/clk while the other three waves read
  // too (~16 cycles per b128, ~8 per transpose read).  Two MFMAs back to back + 16 other instructions per k-step measured 62 cycles per MFMA
  // in A || B and 69 in C (tools/attn_bwd_stamps.py); the stream below alternates ONE MFMA with its share of the other work everywhere:
  //   A || B, per k-step:  S MFMA | DMA piece, AGPR reads + fma + exp of element pair ks | dP MFMA | mul + cvt of pair ks - 1 (its
  //                        exponentials were issued a whole MFMA earlier), one transpose-read pair of C's first half, lse / D of the next group
  //   C, first half:       MFMA | 2 transpose reads of the second half + 1 row fragment of A(u + 2)      (x 2 NDB)
  //   C, second half:      MFMA | 1 row fragment of A(u + 2)                                               (x 2 NDB)
  // asm volatile statements (MFMAs, AGPR reads, DMA pieces, the empty `pin`s) keep their source order; pin(x) forces x to be computed in
  // front of the next MFMA, a "memory" clobber keeps LDS reads on their side of it.
  f32x16 dpv[2];                                   // dP of the even / odd units: VGPR accumulators (B reads them directly)
  auto step = [&](auto DA_, auto DBC_, auto MASK_, auto TBC_, const char* cb, int qr0, const char* fb, int tbf) __attribute__((always_inline)) {
    constexpr bool DA = decltype(DA_)::value, DBC = decltype(DBC_)::value, MASK = decltype(MASK_)::value;
    constexpr int tbc = decltype(TBC_)::value;          // half of the current unit = its S / dP register set; A writes the other set
    constexpr int RSC = RS + 16 * tbc, RSN = RS + 16 * (1 - tbc);
    constexpr int EPK = 16 / NKS;                  // elements of B per k-step of A (2 at hd 128, 4 at hd 64)
    bf16x8 trO[2][NDB], trQ[2][NDB], bP[2], bS[2];
    f32x4 l4[4], d4[4];
    float ex[EPK], dd[EPK];                        // exponentials and (dP - D) of the element group whose mul + cvt half is still to come
    const float* sm = reinterpret_cast<const float*>(cb + 2 * TILE) + tbc * 32 + 4 * hh;
    if constexpr (DBC) {                           // lse / D of query rows 8 g + 4 hh + 0..3 of the unit; group g + 1 is requested while group g is used
      l4[0] = *reinterpret_cast<const f32x4*>(sm);
      d4[0] = *reinterpret_cast<const f32x4*>(sm + 64);
    }
    auto finish = [&](auto kp_) __attribute__((always_inline)) {      // mul + cvt half of element group kp
      constexpr int kp = decltype(kp_)::value;
#pragma unroll
      for (int j = 0; j < EPK; ++j) {
        const int r = kp * EPK + j;
        bP[r >> 3][r & 7] = f2bf(ex[j]);
        bS[r >> 3][r & 7] = f2bf(ex[j] * dd[j]);
      }
      // the packed words of this group (EPK / 2 dwords of each operand) must exist in front of the next MFMA
      u32x4 wP, wS;
      __builtin_memcpy(&wP, &bP[(kp * EPK) >> 3], 16);
      __builtin_memcpy(&wS, &bS[(kp * EPK) >> 3], 16);
      constexpr int w0 = ((kp * EPK) & 7) / 2;
      if constexpr (EPK == 2) asm volatile("" : "+v"(wP[w0]), "+v"(wS[w0]));
      else asm volatile("" : "+v"(wP[w0]), "+v"(wP[w0 + 1]), "+v"(wS[w0]), "+v"(wS[w0 + 1]));
      __builtin_memcpy(&bP[(kp * EPK) >> 3], &wP, 16);
      __builtin_memcpy(&bS[(kp * EPK) >> 3], &wS, 16);
    };
    sfor<0, NKS>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      if constexpr (DA) mfma_va<RSN, RKF + 4 * ks, ks == 0>(qa[ks]);
      if constexpr (tbc == 0) piece(ks_);          // this iteration's DMA pieces: one per k-step of its first step (PP == NKS)
      float en[EPK], dn[EPK];
      if constexpr (DBC) {
        constexpr int g4 = (ks * EPK) >> 2;
        sfor<0, EPK>([&](auto j_) {
          constexpr int j = decltype(j_)::value, r = ks * EPK + j, e = r & 3;
          const float sv = agpr_read<RSC + r>();
          const float dv = dpv[tbc][r];
          float a = fmaf(sv, sl2, l4[g4][e]);        // l4 = -lse log2(e): pre-scaled in LDS by wave 3 one iteration ahead
          if constexpr (MASK) {
            const int qg = qr0 + 8 * g4 + 4 * hh + e;
            const bool ok = (qg < p.S) && (kvrow < p.S) && (!p.causal || kvrow <= qg);
            a = ok ? a : -INFINITY;                // exp2(-inf) = 0: the select sits in front of the exponential (no exec branch per element)
          }
          en[j] = __builtin_amdgcn_exp2f(a);
          dn[j] = dv - d4[g4][e];
        });
        if constexpr (EPK == 2) asm volatile("" : "+v"(en[0]), "+v"(en[1]), "+v"(dn[0]), "+v"(dn[1]));
        else asm volatile("" : "+v"(en[0]), "+v"(en[1]), "+v"(en[2]), "+v"(en[3]), "+v"(dn[0]), "+v"(dn[1]), "+v"(dn[2]), "+v"(dn[3]));
      }
      if constexpr (DA) mfma_va_v<RVF + 4 * ks, ks == 0>(dpv[1 - tbc], da[ks]);
      if constexpr (DBC) {
        if constexpr (ks > 0) finish(std::integral_constant<int, ks - 1>{});
#pragma unroll
        for (int j = 0; j < EPK; ++j) { ex[j] = en[j]; dd[j] = dn[j]; }
        if constexpr (ks < NDB) {                  // the transpose-read fragments of C's first half (query rows 0-15 of the unit), one (dO^T, Q^T) pair per k-step
          trO[0][ks] = frag_rows_tr2<HD>(cb + TILE, trb, ks, tbc, 0);
          trQ[0][ks] = frag_rows_tr2<HD>(cb, trb, ks, tbc, 0);
        }
        constexpr int gn = ((ks * EPK) >> 2) + 1;  // lse / D of the next group, requested when the current group's first elements are read
        if constexpr (((ks * EPK) & 3) == 0 && gn < 4) {
          l4[gn] = *reinterpret_cast<const f32x4*>(sm + 8 * gn);
          d4[gn] = *reinterpret_cast<const f32x4*>(sm + 64 + 8 * gn);
        }
        asm volatile("" ::: "memory");             // pins the LDS reads above to this k-step (hipcc would hoist them all to the top: +60 VGPRs)
      }
    });
    if constexpr (tbc == 0) piece(std::integral_constant<int, PP>{});
    if constexpr (DBC) finish(std::integral_constant<int, NKS - 1>{});
    asm volatile("" ::: "memory");
#ifdef AB_STAMP
    if (stamps && cur_it < 64) stamps[(64 + cur_it) * 8 + 6 + tbc] = __builtin_amdgcn_s_memtime();
#endif
    // C: dV^T += dO^T P, dK^T += Q^T dS
    constexpr int FPD = NKS / NDB;                 // row-fragment k-steps per d block (2)
    sfor<0, NDB>([&](auto d_) {
      constexpr int d = decltype(d_)::value;
      if constexpr (DBC) {
        mfma_vv<RV + 16 * d>(trO[0][d], bP[0]);
        trO[1][d] = frag_rows_tr2<HD>(cb + TILE, trb, d, tbc, 1);
      }
      qa[d * FPD] = frag_rows<HD>(fb, fbase, tbf, d * FPD);
      asm volatile("" ::: "memory");
      if constexpr (DBC) {
        mfma_vv<RK + 16 * d>(trQ[0][d], bS[0]);
        trQ[1][d] = frag_rows_tr2<HD>(cb, trb, d, tbc, 1);
      }
      da[d * FPD] = frag_rows<HD>(fb + TILE, fbase, tbf, d * FPD);
      asm volatile("" ::: "memory");
    });
    sfor<0, NDB>([&](auto d_) {
      constexpr int d = decltype(d_)::value;
      if constexpr (DBC) mfma_vv<RV + 16 * d>(trO[1][d], bP[1]);
#pragma unroll
      for (int ks = d * FPD + 1; ks < (d + 1) * FPD; ++ks) qa[ks] = frag_rows<HD>(fb, fbase, tbf, ks);
      asm volatile("" ::: "memory");
      if constexpr (DBC) mfma_vv<RK + 16 * d>(trQ[1][d], bS[1]);
#pragma unroll
      for (int ks = d * FPD + 1; ks < (d + 1) * FPD; ++ks) da[ks] = frag_rows<HD>(fb + TILE, fbase, tbf, ks);
      asm volatile("" ::: "memory");
    });
    if constexpr (DA && !DBC) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory", A3V_AGPR_0_255);   // A's results -> the next step's AGPR reads with no C phase in between
  };
  using T_ = std::true_type; using F_ = std::false_type;
  // cur live (+ next live or dead): the full step (A on a dead next unit multiplies fragments nobody reads: harmless, and only at the end of a
  // block); cur dead + next live: A only (the first units of waves 1-3 of a causal block); both dead: fetch only.
  auto run = [&](bool lc, bool ln, bool mk, auto tbc, const char* cb, int qr0, const char* fb, int tbf) __attribute__((always_inline)) {
    if (lc) {
      if (mk) step(T_{}, T_{}, T_{}, tbc, cb, qr0, fb, tbf);
      else step(T_{}, T_{}, F_{}, tbc, cb, qr0, fb, tbf);
    } else if (ln) {
      step(T_{}, F_{}, F_{}, tbc, cb, qr0, fb, tbf);
    } else {
      step(F_{}, F_{}, F_{}, tbc, cb, qr0, fb, tbf);
    }
  };

  // lse of a landed pair -> -lse log2(e) in place (64 floats, wave 3, one iteration before the pair's units read it: the barrier in
  // between publishes it): one multiply per query row instead of one per (query, key) element
  auto prescale_lse = [&](const char* buf) __attribute__((always_inline)) {
    if (wave == 3) {
      float* sm = const_cast<float*>(reinterpret_cast<const float*>(buf + 2 * TILE));
      sm[lane] = -1.4426950408889634f * sm[lane];
    }
  };
  // ---- prologue: pairs 0, 1, 2 in flight; A(0) and the fragments of A(1) from pair 0
#ifdef AB_STAMP
#define AB_PE(k) do { if (stamps) stamps[2 * 64 * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AB_PE(k) do {} while (0)
#endif
  // K and V rows of the block's 128 keys go through LDS like every other tile -- ring buffers 2 and 3, free until pairs 2 / 3 are issued --
  // by coalesced LDS-DMA, and each wave reads its keys' fragments (B operands of S = Q K^T and dP = dO V^T) back with ds_read_b128.  The
  // first build loaded them straight into registers, 16 bytes per lane at a 256-byte stride: 1024 sector requests per wave, ~4 us per
  // block of pure address-unit time in front of the first MFMA (tools/attn_bwd_timeline.py).  Pair 0 is issued first, pair 1 once the
  // fragments sit in their AGPRs, pair 2 between the MFMAs of A(0).
  AB_PE(0);
#ifdef AB_STAMP
  if (tl) tl[6] = __builtin_amdgcn_s_memrealtime();
#endif
  issue(true);
#ifdef AB_STAMP
  if (tl) tl[7] = __builtin_amdgcn_s_memrealtime();
#endif
  {
    const bf16_t* Kb = p.k + b * p.k_sb + hk * p.k_sh;
    const bf16_t* Vb = p.v + b * p.v_sb + hk * p.v_sh;
    const unsigned kvoffK = dma_lane_offset0<HD>(HD, tid), kvoffV = dma_lane_offset0<HD>(p.v_ss, tid);
#pragma unroll
    for (int t = 0; t < 2; ++t) {                    // two 64-key tiles each
      stage_rows_dma_asm<HD>(Kb, HD, kt_ * 128 + 64 * t, p.S, lds0 + 2 * BUFB + t * TILE, kvoffK, wave);
      stage_rows_dma_asm<HD>(Vb, p.v_ss, kt_ * 128 + 64 * t, p.S, lds0 + 3 * BUFB + t * TILE, kvoffV, wave);
    }
    sfor<0, 128>([&](auto i) { agpr_write<decltype(i)::value>(0u); });       // dV^T / dK^T = 0 while the tiles fly
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", A3V_AGPR_0_255);
    __syncthreads();
    prescale_lse(lds);                                                        // pair 0 (published by the barrier below)
    const char* Kt = lds + 2 * BUFB + (wave >> 1) * TILE;                     // this wave's 32 keys: rows 32 (wave & 1) .. of tile wave >> 1
    const char* Vt = lds + 3 * BUFB + (wave >> 1) * TILE;
    sfor<0, NKS>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      const bf16x8 kq = frag_rows<HD>(Kt, fbase, wave & 1, ks), vq = frag_rows<HD>(Vt, fbase, wave & 1, ks);
      u32x4 ku, vu;
      __builtin_memcpy(&ku, &kq, 16);
      __builtin_memcpy(&vu, &vq, 16);
      sfor<0, 4>([&](auto j) {
        agpr_write<RKF + 4 * ks + decltype(j)::value>(ku[decltype(j)::value]);
        agpr_write<RVF + 4 * ks + decltype(j)::value>(vu[decltype(j)::value]);
      });
    });
    __syncthreads();                                  // every wave has its fragments: buffers 2 / 3 may take pairs 2 / 3
  }
  AB_PE(1);
  // (pair 0 landed with the K / V tiles: one wait above)
  AB_PE(2);
  issue(total > 1);
  AB_PE(3);
  fetch(lds, 0);
  prep_issue(total > 2);
  sfor<0, NKS>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    mfma_va<RS, RKF + 4 * ks, ks == 0>(qa[ks]);
    piece(ks_);
    mfma_va_v<RVF + 4 * ks, ks == 0>(dpv[0], da[ks]);
  });
  piece(std::integral_constant<int, PP>{});
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory", A3V_AGPR_0_255);
  fetch(lds, 1);
  AB_PE(4);
  int qi_c = 0, buf_c = 0;
#ifdef AB_STAMP
  if (tl) tl[1] = __builtin_amdgcn_s_memrealtime();
#endif
  for (int it = 0; it < total; ++it) {
    const int q0 = (q_begin + qi_c) * 64;
    if (++qi_c == n_it) qi_c = 0;
    const int q0x = (q_begin + qi_c) * 64;                        // first query row of the next pair (if there is one)
    const char* cb = lds + buf_c * BUFB;
    if (++buf_c == NB) buf_c = 0;
    const char* xb = lds + buf_c * BUFB;                          // the next pair's buffer
    const bool more = it + 1 < total;
#ifdef AB_STAMP
    cur_it = it;
#endif
    AB_ST(1, it, 0);
    // pair it + 1 (issued two iterations ago) has landed -- the pieces of pair it + 2 may still be in flight; every wave is done with pair
    // it - 1, whose buffer pair it + 3 goes to (its pieces ride between the MFMAs of the step below)
    if constexpr (PP == 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory", A3V_AGPR_0_255); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory", A3V_AGPR_0_255);
    AB_ST(1, it, 1);
    __syncthreads();
    AB_ST(1, it, 2);
    prep_issue(it + 3 < total);
    if (more) prescale_lse(xb);                                   // pair it + 1 has landed; its units run after the next barrier
    AB_ST(1, it, 3);
    const bool l0 = !dead(q0), l1 = !dead(q0 + 32), lx = more && !dead(q0x);
    run(l0, l1, !interior(q0), std::integral_constant<int, 0>{}, cb, q0, xb, 0);
    AB_ST(1, it, 4);
    run(l1, lx, !interior(q0 + 32), std::integral_constant<int, 1>{}, cb, q0 + 32, xb, 1);
    AB_ST(1, it, 5);
  }
#ifdef AB_STAMP
  if (tl) tl[2] = __builtin_amdgcn_s_memrealtime();
#endif
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory", A3V_AGPR_0_255);               // the last MFMAs -> the AGPR reads below
  AB_PE(5);
  // Epilogue: both 32 x HD tiles of the wave leave through private LDS patches (the ring is free after one more barrier) as whole rows, 16
  // bytes per lane (per-lane row stores put 16 bytes into each of 32 rows per instruction: ~7 B/clk/CU, 4.6 us per block in the first
  // build); packed mode: dK rotated back by -theta (LLM/llama_ens5.py:123-135 backward), its cos / sin rows requested before anything else
  const float* CS = PACKED ? p.cos_sin + (int64_t)(p.rope_pos0 + kc) * HD : nullptr;
  f32x4 cs[NDB][4];
  if constexpr (PACKED) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) cs[d][g4] = *reinterpret_cast<const f32x4*>(CS + d * 32 + g4 * 8 + hh * 4);
  }
  __syncthreads();
  AB_PE(6);
  char* patchV = lds + wave * (32 * HD * 2);
  char* patchK = lds + 4 * (32 * HD * 2) + wave * (32 * HD * 2);
  const int wx = (kl & (HD / 8 - 1)) << 1;
  sfor<0, NDB>([&](auto d_) {
    constexpr int d = decltype(d_)::value;
    const f32x16 av = agpr_read16<RV + 16 * d>();
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = f2bf(av[g4 * 4 + e]);
      *reinterpret_cast<bf16x4*>(patchV + kl * (HD * 2) + (((d * 8 + g4 * 2 + hh) ^ wx) << 3)) = ov;
    }
  });
  bf16_t* baseV = PACKED ? p.dqkv + (int64_t)b * p.S * p.ld_qkv + (int64_t)(p.H + p.Hkv + hk) * HD : p.dv + ((int64_t)b * p.Hkv + hk) * p.S * HD;
  bf16_t* baseK = PACKED ? p.dqkv + (int64_t)b * p.S * p.ld_qkv + (int64_t)(p.H + hk) * HD : p.dk + ((int64_t)b * p.Hkv + hk) * p.S * HD;
  const int64_t ldo = PACKED ? p.ld_qkv : HD;
  store_patch_rows<HD>(patchV, baseV, ldo, kv_lo, p.S, lane);
  AB_PE(7);
  sfor<0, NDB>([&](auto d_) {
    constexpr int d = decltype(d_)::value;
    f32x16 ak = agpr_read16<RK + 16 * d>();
#pragma unroll
    for (int r = 0; r < 16; ++r) ak[r] *= p.scale;          // dS was formed without the 1 / sqrt(hd) factor (see the dQ kernel)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float v0 = ak[g4 * 4 + 0], v1 = ak[g4 * 4 + 1], v2 = ak[g4 * 4 + 2], v3 = ak[g4 * 4 + 3];
      bf16x4 ok;
      if constexpr (PACKED) {
        const f32x4 c4 = cs[d][g4];
        ok[0] = f2bf(v0 * c4[0] + v1 * c4[1]); ok[1] = f2bf(-v0 * c4[1] + v1 * c4[0]);
        ok[2] = f2bf(v2 * c4[2] + v3 * c4[3]); ok[3] = f2bf(-v2 * c4[3] + v3 * c4[2]);
      } else {
        ok[0] = f2bf(v0); ok[1] = f2bf(v1); ok[2] = f2bf(v2); ok[3] = f2bf(v3);
      }
      *reinterpret_cast<bf16x4*>(patchK + kl * (HD * 2) + (((d * 8 + g4 * 2 + hh) ^ wx) << 3)) = ok;
    }
  });
  store_patch_rows<HD>(patchK, baseK, ldo, kv_lo, p.S, lane);
  AB_PE(8);
#ifdef AB_STAMP
  if (tl) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
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
  // A3V_ATTN_BWD_V2 (default 1): dK and dV by the one-pass, one-wave-per-SIMD kernel (0: the two round-1 launches; A/B and equality tests)
  const int v2 = A3V_ENV_INT("A3V_ATTN_BWD_V2", 1);
#define A3V_BWD_LAUNCH(HDV, PK)                                                                  \
  do {                                                                                           \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HDV, PK>), gq, dim3(256), 0, st, p);                  \
    if (v2 & 1) {                                                                                \
      hipLaunchKernelGGL((attn_bwd_dkv2_kernel<HDV, PK>), gk, dim3(256), 0, st, p);              \
    } else {                                                                                     \
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 0, PK>), gk, dim3(256), 0, st, p);            \
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDV, 1, PK>), gk, dim3(256), 0, st, p);            \
    }                                                                                            \
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
extern "C" __attribute__((visibility("hidden"))) int a3v_attention_bwd_mfma_packed(const void* q, const void* k, int64_t k_sb, int64_t k_sh, const void* v, int64_t v_sb,
                                             int64_t v_ss, int64_t v_sh, const void* dout, const float* lse, const float* D, void* dqkv,
                                             int64_t ld_qkv, const float* cos_sin, int rope_pos0, int B, int S, int H, int Hkv, int hd,
                                             int causal, void* stream) {
  if (!dqkv || !cos_sin || ld_qkv % 4 || rope_pos0 < 0) return A3V_ERR_ARG;
  return attention_bwd_mfma_impl(q, k, k_sb, k_sh, v, v_sb, v_ss, v_sh, dout, lse, D, nullptr, nullptr, nullptr, dqkv, ld_qkv, cos_sin, rope_pos0,
                                 B, S, H, Hkv, hd, causal, stream);
}

