// Streaming LDS-DMA bandwidth per CU with GEMM-like sharing: every iteration each block DMA-loads a
// 32 KB "A piece" shared by SA blocks of its XCD and a 32 KB "B piece" shared by SB blocks, then moves on
// (no temporal reuse).  Reports bytes landed in LDS per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t total, unsigned* out, int iters, int SA, int SB, int waitmode) {
  __shared__ __attribute__((aligned(1024))) char lds[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;          // 32 blocks per XCD
  const int ia = loc / SA, ib = loc % SB;                          // panel ids inside the XCD
  const int nA = 32 / SA, nB = SB;                                 // distinct panels per XCD per step
  const size_t step_bytes = (size_t)8 * (nA + nB) * 32 * 1024;     // all XCDs
  for (int it = 0; it < iters; ++it) {
    const size_t s0 = ((size_t)it * step_bytes) % (total - step_bytes);
    const char* pa = src + s0 + ((size_t)xcd * (nA + nB) + ia) * 32 * 1024;
    const char* pb = src + s0 + ((size_t)xcd * (nA + nB) + nA + ib) * 32 * 1024;
    char* dst = lds + (it & 1) * 65536;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int off = (wave * 4 + c) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + off + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + off), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + off + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + 32768 + off), 16, 0, 0);
    }
    if (waitmode == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (waitmode == 2) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = *(unsigned*)(lds + threadIdx.x * 4);
}
int main() {
  for (size_t mb : {128, 1024}) {
    const size_t total = mb << 20;
    char* src; unsigned* out;
    if (hipMalloc(&src, total) != hipSuccess) return 1;
    (void)hipMemset(src, 1, total);
    (void)hipMalloc(&out, 256 * 512 * 4);
    for (int wm : {0, 1, 2})
      for (auto sh : {std::pair<int,int>{1, 32}, {4, 8}, {8, 4}}) {
        const int iters = 4000;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, src, total, out, 50, sh.first, sh.second == 32 ? 1 : sh.second, wm);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, src, total, out, iters, sh.first, sh.second == 32 ? 1 : sh.second, wm);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * iters * 65536;
        printf("buf %4zu MB wait %d shareA %d shareB %2d: %6.2f TB/s into LDS, %5.1f B/clk/CU @2.1GHz, %6.0f cycles per 64KB\n", mb, wm, sh.first,
               sh.second == 32 ? 1 : sh.second, bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.1, 65536.0 / (bytes / ms / 1e6 / 256 / 2.1));
      }
    (void)hipFree(src); (void)hipFree(out);
  }
  return 0;
}
