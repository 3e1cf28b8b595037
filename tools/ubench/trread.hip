// What does ds_read_b64_tr_b16 return?  LDS holds element i at index i (16-bit); lane l passes byte address A(l); we print the
// four 16-bit values each lane receives for two address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned short* out, int mode) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = l * 8;                                   // lane l -> its own 8 contiguous bytes
  else addr = ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 1024;   // [4 rows of 128 elems][16 cols]: lane i -> row i/4, cols 4(i%4)..; groups 1 KB apart
  unsigned lo, hi;
  unsigned base = (unsigned)(uintptr_t)lds + addr;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base) : "memory");
  lo = (unsigned)v; hi = (unsigned)(v >> 32);
  out[l * 4 + 0] = lo & 0xffff; out[l * 4 + 1] = lo >> 16; out[l * 4 + 2] = hi & 0xffff; out[l * 4 + 3] = hi >> 16;
}
int main() {
  unsigned short* d; (void)hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : "   ");
  }
  return 0;
}
