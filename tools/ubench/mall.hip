// Does the 256-MB Infinity Cache (memory-side) keep a weight matrix between two kernels, and how fast is a streaming read that hits it?
// Decode reads every weight once per step from HBM; between consecutive kernels HBM idles (ramp / tail).  If a small co-resident
// kernel can pull the NEXT matrix into the Infinity Cache while the current GEMV runs, the GEMV streams from there.
//   cold  : after flushing with a 2-GB read;   warm : immediately after a read of the same buffer (plain or nt first touch)
//   co-run: a warm read of X on one stream while another stream streams a cold buffer Y (HBM busy): does the hit rate survive?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int NT>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ p, size_t bytes, unsigned* out) {
  const size_t n = bytes / 16, stride = (size_t)gridDim.x * 256 * 4;
  u4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
    u4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t j = i + q * 256;
      if (j < n) v[q] = NT ? __builtin_nontemporal_load((const u4*)p + j) : ((const u4*)p)[j]; else v[q] = u4{0, 0, 0, 0};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc ^= v[q];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

static float timed(void (*kern)(const char*, size_t, unsigned*), const char* p, size_t bytes, unsigned* out, int blocks, hipStream_t st) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, p, bytes, out);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  char *X, *Y, *F; unsigned* out;
  const size_t MB = 1 << 20;
  hipMalloc(&X, 256 * MB); hipMalloc(&Y, 256 * MB); hipMalloc(&F, 2048 * MB); hipMalloc(&out, 64);
  hipMemset(X, 1, 256 * MB); hipMemset(Y, 2, 256 * MB); hipMemset(F, 3, 2048 * MB);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  for (size_t mb : {32, 90, 128, 180, 230}) {
    for (int first_nt = 0; first_nt < 2; ++first_nt)
      for (int second_nt = 0; second_nt < 2; ++second_nt) {
        timed(rd<0>, F, 2048 * MB, out, 2048, s1);                              // flush
        const float cold = timed(first_nt ? rd<1> : rd<0>, X, mb * MB, out, 2048, s1);
        const float warm = timed(second_nt ? rd<1> : rd<0>, X, mb * MB, out, 2048, s1);
        const float warm2 = timed(second_nt ? rd<1> : rd<0>, X, mb * MB, out, 2048, s1);
        printf("%3zu MB first %s then %s: cold %6.1f us (%5.2f TB/s)  warm %6.1f us (%5.2f TB/s)  warm again %6.1f us (%5.2f TB/s)\n", mb,
               first_nt ? "nt   " : "plain", second_nt ? "nt   " : "plain", cold, mb * MB / cold / 1e6, warm, mb * MB / warm / 1e6, warm2, mb * MB / warm2 / 1e6);
      }
    // co-run: X warm (plain first touch), consumer reads X with nt on s1 while a 64-block prefetcher streams cold Y on s2
    timed(rd<0>, F, 2048 * MB, out, 2048, s1);
    timed(rd<0>, X, mb * MB, out, 2048, s1);
    hipDeviceSynchronize();
    hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    hipEventRecord(b0, s2);
    hipLaunchKernelGGL(rd<0>, dim3(256), dim3(256), 0, s2, (const char*)Y, mb * MB, out);       // the "prefetcher": one small block per CU
    hipEventRecord(b1, s2);
    hipEventRecord(a0, s1);
    hipLaunchKernelGGL(rd<1>, dim3(2048), dim3(256), 0, s1, (const char*)X, mb * MB, out);
    hipEventRecord(a1, s1);
    hipDeviceSynchronize();
    float ta, tb; hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb, b0, b1);
    printf("%3zu MB co-run: warm consumer %6.1f us (%5.2f TB/s)   cold prefetcher (256 blocks) %6.1f us (%5.2f TB/s)\n", mb, ta * 1e3, mb * MB / ta / 1e9, tb * 1e3,
           mb * MB / tb / 1e9);
  }
  return 0;
}
