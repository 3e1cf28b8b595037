// Are 16-byte global stores / loads at 2-byte-aligned addresses legal on this box (unaligned access mode), and what do they cost?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(char* p, int off, int stride) {
  u32x4 v = {threadIdx.x, 1u, 2u, 3u};
  char* q = p + off + (size_t)(blockIdx.x * 256 + threadIdx.x) * stride;
  __builtin_memcpy(q, &v, 16);   // compiler may split
}
__global__ void k2(char* p, int off, int stride) {
  u32x4 v = {threadIdx.x, 1u, 2u, 3u};
  char* q = p + off + (size_t)(blockIdx.x * 256 + threadIdx.x) * stride;
  asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(q), "v"(v) : "memory");
}
int main() {
  char* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0xff, 64 << 20);
  for (int off : {0, 2, 6, 14}) {
    hipLaunchKernelGGL(k2, dim3(1024), dim3(256), 0, 0, d, off, 32);
    hipError_t e = hipDeviceSynchronize();
    unsigned h[12]; hipMemcpy(h, d + off + 5 * 32, 16, hipMemcpyDeviceToHost);
    printf("offset %2d: sync %s  thread5 wrote %u %u %u %u\n", off, hipGetErrorString(e), h[0], h[1], h[2], h[3]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k2, dim3(65536), dim3(256), 0, 0, d, off, 2);   // overlapping, dense
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("   dense 16-B stores at 2-B stride: %.1f us per launch\n", ms * 1e3 / 20);
  }
  return 0;
}
