// Store-issue microbenchmark on gfx950: how fast can the 8 waves of a GEMM block push a 256 x 256 bf16 output tile (128 KB) out?
// The patterns are the ones the GEMM epilogues use.  Every block owns tiles of a [rows][ldc] byte matrix like the persistent
// GEMM (tile t of block b = tile b + t * gridDim.x of a tiles_n-wide grid).
//   mode 0: 16 x dwordx4 per wave, an instruction covers 8 rows x 128 B (LDS-staged epilogue)
//   mode 1: 32 x dwordx2 per wave, an instruction covers 16 rows x 32 B (direct 16x16 accumulator layout)
//   mode 2: 16 x dwordx4 per wave, an instruction covers 1 KB contiguous (row-major tile image: the bound of the store path)
//   mode 3: mode 0 with non-temporal stores
//   mode 4: 16 x dwordx4 per wave, an instruction covers 4 rows x 256 B
//   mode 5: 16 x dwordx4, 2 rows x 512 B (all four column waves' share of two rows: needs a cross-wave LDS transpose)
// build: hipcc --offload-arch=gfx950 -O3 -o bin/stores stores.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <int MODE>
__global__ __launch_bounds__(512) void k(char* C, long ldc, int tiles_n, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int tile = blockIdx.x + it * gridDim.x;
    const long tm = tile / tiles_n, tn = tile % tiles_n;
    char* base = C + (tm * 256) * ldc + tn * 512;
    if (MODE == 0 || MODE == 3) {
      char* p = base + (long)(wr * 128 + (lane >> 3)) * ldc + wc * 128 + (lane & 7) * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (MODE == 3) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + (long)c * 8 * ldc));
        else *reinterpret_cast<u32x4*>(p + (long)c * 8 * ldc) = v;
      }
    } else if (MODE == 1) {
      char* p = base + (long)(wr * 128 + (lane & 15)) * ldc + wc * 128 + (lane >> 4) * 8;
      u32x2 w = {v[0], v[1]};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x2*>(p + (long)i * 16 * ldc + j * 32) = w;
    } else if (MODE == 2) {
      char* p = C + (long)tile * 131072 + wave * 16384 + lane * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) *reinterpret_cast<u32x4*>(p + c * 1024) = v;
    } else if (MODE == 4) {
      char* p = base + (long)(wr * 128 + (wc >> 1) * 64 + (lane >> 4)) * ldc + (wc & 1) * 256 + (lane & 15) * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) *reinterpret_cast<u32x4*>(p + (long)c * 4 * ldc) = v;
    } else {
      char* p = base + (long)(wave * 32 + (lane >> 5)) * ldc + (lane & 31) * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) *reinterpret_cast<u32x4*>(p + (long)c * 2 * ldc) = v;
    }
    v[2] += 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

template <int MODE>
void run(char* C, unsigned long long* cyc, long ldc, int nblk, int iters) {
  const int tiles_n = (int)(ldc / 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(512), 0, 0, C, ldc, tiles_n, 2, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(512), 0, 0, C, ldc, tiles_n, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long issue_max = 0, done_max = 0;
  for (int w = 0; w < 8; ++w) { if (h[w * 2] > issue_max) issue_max = h[w * 2]; if (h[w * 2 + 1] > done_max) done_max = h[w * 2 + 1]; }
  const double bytes = (double)nblk * iters * 131072;
  printf("mode %d ldc %6ld blocks %3d tiles/block %2d: %6.2f TB/s  block 0: issue %7llu cyc  drained %7llu cyc  = %5.1f B/clk/CU, %6.0f cyc per 128-KB tile\n",
         MODE, ldc, nblk, iters, bytes / ms / 1e9, issue_max, done_max, (double)iters * 131072 / done_max, (double)done_max / iters);
}

int main(int argc, char** argv) {
  char* C; unsigned long long* cyc;
  const size_t cap = (size_t)3 << 30;
  hipMalloc(&C, cap); hipMalloc(&cyc, 256 * 8 * 2 * 8);
  hipMemset(C, 0, cap);
  const long ldcs[] = {8192, 24576, 16384, 44032};
  for (long ldc : ldcs)
    for (int nblk : {256, 32}) {
      for (int iters : {1, 4}) {
        if ((size_t)(nblk * iters / (ldc / 512) + 1) * 256 * ldc > cap) continue;
        run<0>(C, cyc, ldc, nblk, iters); run<1>(C, cyc, ldc, nblk, iters); run<2>(C, cyc, ldc, nblk, iters);
        run<3>(C, cyc, ldc, nblk, iters); run<4>(C, cyc, ldc, nblk, iters); run<5>(C, cyc, ldc, nblk, iters);
      }
    }
  return 0;
}
