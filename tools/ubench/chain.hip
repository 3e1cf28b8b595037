// Gate for VERDICT r5 item 3 (decode: overlap a GEMV's fill with its predecessor's drain): two LDS-DMA weight-stream kernels of
// ~100 MB each, (a) back to back on ONE stream, (b) alternating over TWO streams with the dependency carried by a device flag that the
// predecessor's last-arriving block sets -- the successor's blocks put their first U pieces per wave in flight BEFORE they poll it.
// Go only if (b) is >= 8 % faster per kernel than (a).     hipcc --offload-arch=gfx950 -O3 chain.hip -o bin/chain && bin/chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NW = 4, U = 8;                     // waves per block, 1-KiB pieces in flight per wave (the GEMVs' private rings)
// each wave streams RPW rows of K2 bytes; flags: wait_flag >= epoch before the stream runs on (nullptr: no wait); the last block to
// arrive on `arrive` publishes epoch in set_flag
template <bool CHAIN>
__global__ __launch_bounds__(NW * 64) void stream_k(const char* __restrict__ W, int rows, int K2, const unsigned* wait_flag, unsigned* arrive,
                                                    unsigned* set_flag, unsigned epoch, unsigned* out, unsigned my_epoch = 0) {
  __shared__ __attribute__((aligned(1024))) char lds[NW * U * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * NW + wave;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)(row < rows ? row : 0) * K2), 0, K2, 0x00020000);
  const unsigned vo = lane * 16;
  int kb = 0;
  auto burst = [&]() {
#pragma unroll
    for (int q = 0; q < U; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * U + q) * 1024), 16, vo, kb + q * 1024, 0, 2);
    kb += U * 1024;
  };
  if (row < rows) burst();                       // weights do not depend on the predecessor: in flight before the poll
  if (CHAIN && wait_flag) {
    if (threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(8);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  if (row < rows) {
    while (kb < K2) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); burst(); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 77 && lds[1023] == 33) out[threadIdx.x] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(set_flag, my_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
int main(int argc, char** argv) {
  const int K2 = 8192 * 2;                       // bytes per row (K = 8192 bf16: 16 pieces of 1 KiB per wave)
  for (int rows : {3072, 6144, 1024}) {           // 50 / 100 / 17 MB per kernel: 768 / 1536 / 256 blocks of 256 threads
    const size_t bytes = (size_t)rows * K2;
    char *W1, *W2; unsigned *flags, *out;
    CK(hipMalloc(&W1, bytes)); CK(hipMalloc(&W2, bytes)); CK(hipMalloc(&flags, 64 * 4)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(W1, 1, bytes)); CK(hipMemset(W2, 2, bytes)); CK(hipMemset(flags, 0, 256));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
    const int blocks = (rows + NW - 1) / NW, reps = 40;
    unsigned epoch = 0;
    float ms_serial = 0, ms_chain = 0, ms_noflag = 0;
    for (int round = 0; round < 3; ++round) {
      // (a) one stream, no flags
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL((stream_k<false>), dim3(blocks), dim3(NW * 64), 0, s1, (i & 1) ? W2 : W1, rows, K2, nullptr, flags + 16, flags + 32, 0u, out);
      }
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_serial, e0, e1));
      // (b) two streams, flag-chained: kernel i waits for kernel i-1's flag
      CK(hipMemset(flags, 0, 256)); CK(hipDeviceSynchronize());
      epoch = 0;
      CK(hipEventRecord(e0, s1)); CK(hipStreamWaitEvent(s2, e0, 0));
      for (int i = 0; i < reps; ++i) {
        ++epoch;
        hipStream_t s = (i & 1) ? s2 : s1;
        // flag i&1 is set by kernel i; kernel i waits for flag (i-1)&1 >= epoch-1 (epoch 0 = nothing to wait for)
        hipLaunchKernelGGL((stream_k<true>), dim3(blocks), dim3(NW * 64), 0, s, (i & 1) ? W2 : W1, rows, K2, i ? flags + ((i - 1) & 1) : nullptr,
                           flags + 16 + (i & 1), flags + (i & 1), i ? epoch - 1 : 0u, out, epoch);
      }
      CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_chain, e0, e1));
      // (c) two streams, no dependency at all (upper bound of what overlap can give)
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1)); CK(hipStreamWaitEvent(s2, e0, 0));
      for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((stream_k<false>), dim3(blocks), dim3(NW * 64), 0, (i & 1) ? s2 : s1, (i & 1) ? W2 : W1, rows, K2, nullptr, flags + 16 + (i & 1), flags + 32, 0u, out);
      CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_noflag, e0, e1));
    }
    printf("%6.1f MB per kernel, %4d blocks: one stream %6.2f us/kernel (%5.2f TB/s) | two streams flag-chained %6.2f us (%5.2f TB/s, %+5.1f %%) | two streams independent %6.2f us (%5.2f TB/s)\n",
           bytes / 1e6, blocks, ms_serial * 1e3 / reps, bytes * reps / ms_serial / 1e9, ms_chain * 1e3 / reps, bytes * reps / ms_chain / 1e9,
           (ms_serial / ms_chain - 1) * 100, ms_noflag * 1e3 / reps, bytes * reps / ms_noflag / 1e9);
    CK(hipFree(W1)); CK(hipFree(W2)); CK(hipFree(flags)); CK(hipFree(out));
  }
  return 0;
}
