// HBM streaming-read rate of one [N, K] bf16 matrix (the decode weight stream) under different lane->address maps:
//   mode 0  "mfma16": lane (row = l&15, chunk = l>>4): a wave instruction touches 16 rows x 64 B  (gemm_skinny1 today)
//   mode 1  "row4":   lane (row = l>>2, chunk = l&3):  same bytes, but a quarter-wave covers 4 rows x 64 B
//   mode 2  "coal":   a wave instruction reads 1 KB contiguous of ONE row (lane l -> 16 B at l*16)
//   mode 3  "coal2":  2 rows x 512 B per instruction
// nt = non-temporal loads.  Block = 8 waves; block b owns RB rows; waves split K (mode 0/1) or rows (mode 2/3).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int MODE, int NT, int U>
__global__ __launch_bounds__(512) void k(const char* __restrict__ W, int N, int K, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ldw = (size_t)K * 2;
  u4 acc = {0, 0, 0, 0};
  if (MODE <= 1) {
    const int row = blockIdx.x * 16 + (MODE == 0 ? (lane & 15) : (lane >> 2));
    const int ch = MODE == 0 ? (lane >> 4) : (lane & 3);
    const int kslice = K / 8;                                // elements per wave
    const char* p = W + (size_t)row * ldw + (size_t)wave * kslice * 2 + ch * 16;
    for (int kb = 0; kb < kslice * 2; kb += U * 64) {
      u4 v[U];
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = NT ? __builtin_nontemporal_load((const u4*)(p + kb + q * 64)) : *(const u4*)(p + kb + q * 64);
#pragma unroll
      for (int q = 0; q < U; ++q) acc ^= v[q];
    }
  } else {
    // block owns 16 rows; wave w owns rows 2w, 2w+1
    const int rpi = MODE == 2 ? 1 : 2;                       // rows per instruction
    const int per = 1024 / rpi;                              // bytes per row per instruction
    for (int r0 = 0; r0 < 2; r0 += rpi) {
      const int row = blockIdx.x * 16 + wave * 2 + r0 + (rpi == 2 ? (lane >> 5) : 0);
      const char* p = W + (size_t)row * ldw + (rpi == 2 ? (lane & 31) : lane) * 16;
      for (size_t kb = 0; kb < ldw; kb += (size_t)U * per) {
        u4 v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = NT ? __builtin_nontemporal_load((const u4*)(p + kb + q * per)) : *(const u4*)(p + kb + q * per);
#pragma unroll
        for (int q = 0; q < U; ++q) acc ^= v[q];
      }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[threadIdx.x] = 1;
}
template <int MODE, int NT, int U>
void run(const char* W, int N, int K, unsigned* out, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NT, U>), dim3(N / 16), dim3(512), 0, 0, W, N, K, out);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, NT, U>), dim3(N / 16), dim3(512), 0, 0, W, N, K, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)N * K * 2;
  printf("N %6d K %6d %-8s nt %d U %d: %7.1f us  %6.2f TB/s\n", N, K, name, NT, U, ms * 1e3 / reps, bytes * reps / ms / 1e9);
}
int main() {
  const size_t maxb = (size_t)22016 * 4096 * 2 * 4;
  char* W; unsigned* out;
  if (hipMalloc(&W, maxb) != hipSuccess) return 1;
  (void)hipMemset(W, 1, maxb);
  (void)hipMalloc(&out, 4096);
  for (auto nk : {std::pair<int,int>{22016, 4096}, {12288, 4096}, {4096, 4096}, {4096, 11008 - 11008 % 512}, {88064, 4096}}) {
    const int N = nk.first, K = nk.second;
    run<0, 1, 8>(W, N, K, out, "mfma16");
    run<0, 0, 8>(W, N, K, out, "mfma16");
    run<1, 1, 8>(W, N, K, out, "row4");
    run<2, 1, 8>(W, N, K, out, "coal");
    run<2, 0, 8>(W, N, K, out, "coal");
    run<2, 1, 4>(W, N, K, out, "coal");
    run<3, 1, 8>(W, N, K, out, "coal2");
  }
  return 0;
}
