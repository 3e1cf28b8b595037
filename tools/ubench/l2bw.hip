// Per-CU L2->CU bandwidth microbenchmark on gfx950: LDS-DMA vs register loads, L2-resident data.
// build: hipcc --offload-arch=gfx950 -O3 -o l2bw l2bw.hip ; run: ./l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int REGION_KB>   // MODE 0: buffer_load..lds ; 1: global_load_dwordx4 to registers
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, unsigned* out, int iters, int share) {
  __shared__ __attribute__((aligned(1024))) char lds[REGION_KB * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // `share` blocks read the same region (models panel sharing inside an XCD: blockIdx/8 picks the region so
  // that the sharers sit on one XCD)
  const int region = ((blockIdx.x >> 3) / share) * 8 + (blockIdx.x & 7);
  const char* base = src + (size_t)region * REGION_KB * 1024;
  constexpr int PIECES = REGION_KB / 8;   // 1-KiB pieces per wave (8 waves)
  unsigned acc = 0;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, REGION_KB * 1024, 0x00020000);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < PIECES; ++c) {
        const int off = (wave * PIECES + c) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + off), 16, lane * 16 + off, 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[PIECES];
#pragma unroll
      for (int c = 0; c < PIECES; ++c) {
        const int off = (wave * PIECES + c) * 1024;
        v[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + off + lane * 16));
      }
#pragma unroll
      for (int c = 0; c < PIECES; ++c) acc += v[c][0] ^ v[c][3];
      asm volatile("" ::: "memory");
    }
  }
  if (MODE == 0) acc = *(unsigned*)(lds + threadIdx.x * 4);
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MODE, int KB>
void run(const char* name, const char* src, unsigned* out, int share) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, KB>), dim3(256), dim3(512), 0, 0, src, out, 10, share);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, KB>), dim3(256), dim3(512), 0, 0, src, out, iters, share);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * iters * KB * 1024;
  printf("%-28s region %3d KB share %d: %7.2f TB/s aggregate, %6.1f GB/s per CU, %5.1f B/clk/CU @2.1GHz\n", name, KB, share,
         bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.1);
}

int main() {
  char* src; unsigned* out;
  hipMalloc(&src, 256 * 64 * 1024);
  hipMemset(src, 1, 256 * 64 * 1024);
  hipMalloc(&out, 256 * 512 * 4);
  for (int share : {1, 4, 8}) {
    run<0, 64>("lds-dma", src, out, share);
    run<1, 64>("global_load->regs", src, out, share);
    run<0, 32>("lds-dma", src, out, share);
    run<1, 32>("global_load->regs", src, out, share);
  }
  return 0;
}
