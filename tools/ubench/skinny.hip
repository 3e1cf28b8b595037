// Decode-GEMV structure experiments: C[M<=16, N] = A[M,K] . W[N,K]^T with W streamed once from HBM.
// Weight copies rotate through a > 1 GB buffer so the 256 MB Infinity Cache never serves a launch.
//   V0: gemm_skinny1 structure (8 waves split K, batch of U loads -> U MFMAs, LDS reduce)
//   V1: V0 + register double buffering (next batch issued before the current MFMAs)
//   V2: pure W read in V0's structure (no A, no MFMA, no reduce)  -> upper bound of the structure
//   V3: V1 with ROWS tiles per block processed sequentially by each wave (longer wave lifetime)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
typedef uint16_t bf16_t;

struct Args { const bf16_t* A; const bf16_t* W; float* C; int M, N, K, kslice; };

template <int V, int U, int TPB>   // TPB: 16-row tiles per block (sequential)
__global__ __launch_bounds__(512) void k(Args p) {
  __shared__ float red[8][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 15, kq = (lane >> 4) * 8;
  const int ar = row < p.M ? row : p.M - 1;
  const bf16_t* ap = p.A + (int64_t)ar * p.K + wave * p.kslice + kq;
  const int kend = p.kslice;
  for (int t = 0; t < TPB; ++t) {
    const int n0 = (blockIdx.x * TPB + t) * 16;
    if (n0 >= p.N) break;
    const bf16_t* wp = p.W + (int64_t)(n0 + row) * p.K + wave * p.kslice + kq;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    if (V == 2) {
      u4 x = {0, 0, 0, 0};
      for (int k = 0; k < kend; k += U * 32) {
        u4 w[U];
#pragma unroll
        for (int q = 0; q < U; ++q) w[q] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(wp + k + q * 32));
#pragma unroll
        for (int q = 0; q < U; ++q) x ^= w[q];
      }
      if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345u) p.C[lane] = 1;
      continue;
    }
    if (V == 6) {      // coalesced pure read: wave w of the block reads rows 2w, 2w+1 of the tile over the full K
      u4 x = {0, 0, 0, 0};
      for (int r = 0; r < 2; ++r) {
        const char* pr = reinterpret_cast<const char*>(p.W + (int64_t)(n0 + wave * 2 + r) * p.K) + lane * 16;
        for (int kb = 0; kb < p.K * 2; kb += U * 1024) {
          u4 w[U];
#pragma unroll
          for (int q = 0; q < U; ++q) w[q] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(pr + kb + q * 1024));
#pragma unroll
          for (int q = 0; q < U; ++q) x ^= w[q];
        }
      }
      if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345u) p.C[lane] = 1;
      continue;
    }
    if (V == 4 || V == 5) {
      bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap);
      for (int k = 0; k < kend; k += U * 32) {
        bf16x8 w[U], a[U];
#pragma unroll
        for (int q = 0; q < U; ++q) w[q] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k + q * 32));
#pragma unroll
        for (int q = 0; q < U; ++q) a[q] = V == 4 ? a0 : *reinterpret_cast<const bf16x8*>(ap + k + q * 32);
#pragma unroll
        for (int q = 0; q < U; ++q) {
          if (q & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q], a[q], acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q], a[q], acc0, 0, 0, 0);
        }
      }
      if (V == 5) {
        if (acc0[0] + acc1[0] == 1.2345f) p.C[lane] = 1;
        continue;
      }
    }
    if (V == 0) {
      for (int k = 0; k < kend; k += U * 32) {
        bf16x8 w[U], a[U];
#pragma unroll
        for (int q = 0; q < U; ++q) w[q] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k + q * 32));
#pragma unroll
        for (int q = 0; q < U; ++q) a[q] = *reinterpret_cast<const bf16x8*>(ap + k + q * 32);
#pragma unroll
        for (int q = 0; q < U; ++q) {
          if (q & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q], a[q], acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q], a[q], acc0, 0, 0, 0);
        }
      }
    } else if (V == 1) {
      bf16x8 w[2][U], a[2][U];
#pragma unroll
      for (int q = 0; q < U; ++q) w[0][q] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + q * 32));
#pragma unroll
      for (int q = 0; q < U; ++q) a[0][q] = *reinterpret_cast<const bf16x8*>(ap + q * 32);
      const int nb = kend / (U * 32);
#pragma unroll 2
      for (int b = 0; b < nb; ++b) {
        const int cur = b & 1, nxt = cur ^ 1;
        if (b + 1 < nb) {
#pragma unroll
          for (int q = 0; q < U; ++q) w[nxt][q] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (b + 1) * U * 32 + q * 32));
#pragma unroll
          for (int q = 0; q < U; ++q) a[nxt][q] = *reinterpret_cast<const bf16x8*>(ap + (b + 1) * U * 32 + q * 32);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
          if (q & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cur][q], a[cur][q], acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cur][q], a[cur][q], acc0, 0, 0, 0);
        }
      }
    }
    if (t) __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc0[r] + acc1[r];
    __syncthreads();
    if (wave == 0) {
      const int m = lane & 15;
      if (m < p.M) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = 0.f;
#pragma unroll
          for (int w8 = 0; w8 < 8; ++w8) s += red[w8][lane][r];
          o[r] = s;
        }
        *reinterpret_cast<f32x4*>(p.C + (int64_t)m * p.N + n0 + (lane >> 4) * 4) = o;
      }
    }
  }
}


// V7: LDS-DMA streaming.  Block = WAVES waves, all on K-slice `sl`, wave w on row tile tg*WAVES + w.  A slice [16][kslice]
// DMA'd once per block into LDS (shared); each wave streams its 16 rows x kslice through a private ring of NS stages of
// 16 rows x 128 k (4 KB, 4 DMA instructions of 4 rows x 256 B), fragments by ds_read_b128 with an XOR swizzle.
template <int WAVES, int NS, int KS_MAX>
__global__ __launch_bounds__(WAVES * 64) void k7(Args p, int S) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tg = blockIdx.x / S, sl = blockIdx.x % S;
  const int kslice = p.kslice, nkb = kslice / 128;
  char* Alds = lds;                                   // nkb x 4 KB
  char* Wring = lds + nkb * 4096 + wave * NS * 4096;  // NS x 4 KB per wave
  const int n0 = (tg * WAVES + wave) * 16;
  // DMA source offsets: instruction i covers rows 4i..4i+3; lane l -> row 4i + (l>>4), slot l&15 holds chunk slot ^ row
  const int dr = lane >> 4, dslot = lane & 15;
  // A slice: nkb*4 instructions spread over the waves
  for (int j = wave; j < nkb * 4; j += WAVES) {
    const int kb = j >> 2, i = j & 3;
    const int row = 4 * i + dr;
    const int ar = row < p.M ? row : p.M - 1;
    const bf16_t* src = p.A + (int64_t)ar * p.K + sl * kslice + kb * 128 + ((dslot ^ row) & 15) * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(Alds + kb * 4096 + i * 1024), 16, 0, 0);
  }
  const bf16_t* wbase = p.W + (int64_t)n0 * p.K + sl * kslice;
  auto dma_stage = [&](int kb, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * i + dr;
      const bf16_t* src = wbase + (int64_t)row * p.K + kb * 128 + ((dslot ^ row) & 15) * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(Wring + slot * 4096 + i * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int st = 0; st < NS; ++st) if (st < nkb) dma_stage(st, st);
  // A landed (in-order completion: everything before the W prologue)
  if (NS == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (NS == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __syncthreads();
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const int fr = lane & 15, fg = lane >> 4;
  for (int kb = 0; kb < nkb; ++kb) {
    const int slot = kb % NS;
    // this stage's 4 DMAs are the oldest outstanding
    if (kb + NS <= nkb) {
      if (NS == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (NS == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* Ws = Wring + slot * 4096 + fr * 256;
    const char* As = Alds + kb * 4096 + fr * 256;
    bf16x8 wf[4], af[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int off = (((4 * s4 + fg) ^ fr) & 15) * 16;
      wf[s4] = *reinterpret_cast<const bf16x8*>(Ws + off);
      af[s4] = *reinterpret_cast<const bf16x8*>(As + off);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (kb + NS < nkb) dma_stage(kb + NS, slot);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if (s4 & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s4], af[s4], acc0, 0, 0, 0);
    }
  }
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = acc0[r] + acc1[r];
  *reinterpret_cast<f32x4*>(p.C + ((int64_t)(blockIdx.x * WAVES + wave) * 64 + lane) * 4) = o;
}
template <int WAVES, int NS>
void run7(const bf16_t* A, const bf16_t* Wbase, size_t wbytes_total, float* C, int M, int N, int K, int S) {
  const size_t wb = (size_t)N * K * 2;
  const int copies = (int)(wbytes_total / wb);
  if (K % (S * 128) || N % (16 * WAVES)) { printf("N %d K %d V7 W%d S%d: skip\n", N, K, WAVES, S); return; }
  Args p{A, Wbase, C, M, N, K, K / S};
  const int blocks = N / (16 * WAVES) * S;
  const size_t ldsb = (size_t)(K / S / 128) * 4096 + (size_t)WAVES * NS * 4096;
  (void)hipFuncSetAttribute((const void*)k7<WAVES, NS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { p.W = Wbase + (size_t)(i % copies) * N * K; hipLaunchKernelGGL((k7<WAVES, NS, 0>), dim3(blocks), dim3(WAVES * 64), ldsb, 0, p, S); }
  (void)hipEventRecord(e0);
  const int reps = 24;
  for (int i = 0; i < reps; ++i) { p.W = Wbase + (size_t)((i + 3) % copies) * N * K; hipLaunchKernelGGL((k7<WAVES, NS, 0>), dim3(blocks), dim3(WAVES * 64), ldsb, 0, p, S); }
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("N %6d K %6d V7 dma waves %d stages %d split %2d lds %3zu KB blocks %5d: %7.1f us  %6.2f TB/s\n", N, K, WAVES, NS, S, ldsb / 1024, blocks,
         ms * 1e3 / reps, (double)wb * reps / ms / 1e9);
}

template <int V, int U, int TPB>
void run(const bf16_t* A, const bf16_t* Wbase, size_t wbytes_total, float* C, int M, int N, int K, const char* name) {
  const size_t wb = (size_t)N * K * 2;
  const int copies = (int)(wbytes_total / wb);
  Args p{A, Wbase, C, M, N, K, K / 8};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = (N / 16 + TPB - 1) / TPB;
  for (int i = 0; i < 3; ++i) { p.W = Wbase + (size_t)(i % copies) * N * K; hipLaunchKernelGGL((k<V, U, TPB>), dim3(blocks), dim3(512), 0, 0, p); }
  (void)hipEventRecord(e0);
  const int reps = 24;
  for (int i = 0; i < reps; ++i) { p.W = Wbase + (size_t)((i + 3) % copies) * N * K; hipLaunchKernelGGL((k<V, U, TPB>), dim3(blocks), dim3(512), 0, 0, p); }
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("N %6d K %6d %-22s U %d TPB %d copies %d: %7.1f us  %6.2f TB/s\n", N, K, name, U, TPB, copies, ms * 1e3 / reps, (double)wb * reps / ms / 1e9);
}
int main() {
  const size_t total = (size_t)3 << 30;
  bf16_t *W, *A; float* C;
  if (hipMalloc(&W, total) != hipSuccess) return 1;
  (void)hipMemset(W, 0x3c, total);
  (void)hipMalloc(&A, 16 * 16384 * 2); (void)hipMemset(A, 0x3c, 16 * 16384 * 2);
  (void)hipMalloc(&C, (size_t)64 << 20);
  for (auto nk : {std::pair<int,int>{22016, 4096}, {12288, 4096}, {4096, 4096}, {4096, 11008 - 11008 % 2048}}) {
    const int N = nk.first, K = nk.second, M = 8;
    run<2, 8, 1>(A, W, total, C, M, N, K, "V2 pure read mfma16");
    run<2, 4, 1>(A, W, total, C, M, N, K, "V2 pure read mfma16");
    run<6, 8, 1>(A, W, total, C, M, N, K, "V6 pure read coalesced");
    run<6, 4, 1>(A, W, total, C, M, N, K, "V6 pure read coalesced");
    run<0, 8, 1>(A, W, total, C, M, N, K, "V0 skinny1");
    run<4, 8, 1>(A, W, total, C, M, N, K, "V4 no A loads");
    run<5, 8, 1>(A, W, total, C, M, N, K, "V5 no reduce");
    for (int S : {4, 8, 16}) {
      run7<4, 2>(A, W, total, C, M, N, K, S);
      run7<4, 3>(A, W, total, C, M, N, K, S);
      run7<4, 4>(A, W, total, C, M, N, K, S);
      run7<8, 2>(A, W, total, C, M, N, K, S);
      run7<8, 3>(A, W, total, C, M, N, K, S);
    }
  }
  return 0;
}
