// Measurement / self-check entries, built into tools/ubench/liba3v_probe.so (NOT part of the product C-ABI; round 5 moved them out of
// liba3vlm_hip.so): what the MFMA pipe of THIS device delivers on a bare stream of
// v_mfma_f32_16x16x32_bf16 (the instruction of the GEMM kernels), once on constant operands and once on random operands.  The chip
// clocks to its power budget: on random bf16 operands the same instruction stream runs 15-25 % slower than on constants
// (tools/ubench/mfmarate.hip, profiles/r04g_mfma_rate.txt), so "fraction of 2.5 PF" and "fraction of what the pipe can deliver on
// this data" are different numbers; bench.py prints both.
#include "../../a3vlm_amd/csrc/a3v_common.h"

int a3v_env_generation() { return 0; }     // (a3v_common.h declares it for A3V_ENV_INT; unused here)

namespace {
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* out, int iters, int random) {
  bf16x8 a, b;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u;
    const unsigned short ua = random ? (unsigned short)(0x3c00 | ((h >> 9) & 0x83ff)) : (unsigned short)0x3f80;
    h = h * 1664525u + 1013904223u;
    const unsigned short ub = random ? (unsigned short)(0x3c00 | ((h >> 9) & 0x83ff)) : (unsigned short)0x3f80;
    a[i] = __builtin_bit_cast(bf16_t, ua);
    b[i] = __builtin_bit_cast(bf16_t, ub);
  }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    // keep the operands changing a little so that nothing is hoisted and the accumulators stay finite
    if ((it & 255) == 255) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

// tflops[0]: constant operands, tflops[1]: random operands (2 M N K per MFMA, dense).  `scratch`: device memory for
// 8 * CUs * 256 floats (the kernel's sink).  Synchronises the stream (HIP events).
extern "C" int a3v_probe_mfma_tflops(int iters, float* scratch, float* tflops, void* stream) {
  if (iters <= 0 || !scratch || !tflops) return A3V_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
  const int blocks = cus * 8;                       // 8 waves per SIMD: the pipe never waits for an issuer
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return A3V_ERR_ARG;
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, st, scratch, 64, r);     // warm-up
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, st, scratch, iters, r);
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return A3V_ERR_ARG; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4.0 * 16.0 * (double)iters * (2.0 * 16 * 16 * 32);
    tflops[r] = ms > 0.f ? (float)(fl / (ms * 1e-3) / 1e12) : 0.f;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}

// Self-check entry: the VALU butterfly reductions of a3v_common.h (v_permlane32/16_swap + DPP) against the __shfl_xor (ds_bpermute)
// forms they replaced, on `n_waves` waves of values from `x` (64 floats per wave): *mismatches = lanes whose sum or max differs in
// ANY bit.  The two butterflies pair the same lanes in the same order, so the count must be zero.
namespace {
__global__ __launch_bounds__(256) void wave_reduce_probe_kernel(const float* __restrict__ x, int n_waves, int* __restrict__ bad) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_waves) return;                                   // (wave-uniform: whole waves leave)
  const float v = x[(int64_t)w * 64 + (threadIdx.x & 63)];
  const float s0 = wave_sum(v), s1 = wave_sum_shfl(v), m0 = wave_max(v), m1 = wave_max_shfl(v);
  if (__builtin_bit_cast(unsigned, s0) != __builtin_bit_cast(unsigned, s1) || __builtin_bit_cast(unsigned, m0) != __builtin_bit_cast(unsigned, m1))
    atomicAdd(bad, 1);
}
}  // namespace

extern "C" int a3v_probe_wave_reduce(const float* x, int n_waves, int* mismatches, void* stream) {
  if (!x || n_waves <= 0 || !mismatches) return A3V_ERR_ARG;
  hipLaunchKernelGGL(wave_reduce_probe_kernel, dim3((n_waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, n_waves, mismatches);
  A3V_LAUNCH_CHECK();
  return A3V_OK;
}
