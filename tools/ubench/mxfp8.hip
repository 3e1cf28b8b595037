// Operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) A and B, unit scales: which k does byte j of lane l hold?
// C[m][n] = sum_k A[m][k] B[n][k] on small-integer fp8 values; three hypotheses for k(l, j) are tried against a CPU product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __host__ inline int kmap(int h, int g, int j) {
  if (h == 0) return g * 32 + j;                           // 32 contiguous k per lane group
  if (h == 1) return (j / 16) * 64 + g * 16 + (j % 16);    // two 16-byte halves, each spread over the 4 groups
  return (j / 8) * 32 + g * 8 + (j % 8);                   // four 8-byte quarters
}
__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int h) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  union { i32x8 v; uint8_t b[32]; } a, b;
  for (int j = 0; j < 32; ++j) {
    a.b[j] = A[r * 128 + kmap(h, g, j)];
    b.b[j] = B[r * 128 + kmap(h, g, j)];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  // standard 16x16 C layout when called as (a, b): lane holds D[row = 4g + i][col = l & 15] with row from A, col from B
  for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = c[i];
}
static uint8_t enc(int v) {   // e4m3fn of a small integer in [-4, 4]
  static const uint8_t t[5] = {0x00, 0x38, 0x40, 0x44, 0x48};   // 0, 1, 2, 3, 4
  return v < 0 ? (t[-v] | 0x80) : t[v];
}
int main() {
  uint8_t hA[16 * 128], hB[16 * 128]; int iA[16 * 128], iB[16 * 128];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 9 - 4; iB[i] = rand() % 7 - 3; hA[i] = enc(iA[i]); hB[i] = enc(iB[i]); }
  float want[256];
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { int s = 0; for (int kk = 0; kk < 128; ++kk) s += iA[m * 128 + kk] * iB[n * 128 + kk]; want[m * 16 + n] = (float)s; }
  uint8_t *dA, *dB; float* dC;
  (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dC, 1024);
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  for (int h = 0; h < 3; ++h) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, h);
    float got[256];
    (void)hipMemcpy(got, dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { bad += got[m * 16 + n] != want[m * 16 + n]; badT += got[n * 16 + m] != want[m * 16 + n]; }
    printf("hypothesis %d: mismatches %d (transposed C: %d)  sample got %.0f want %.0f\n", h, bad, badT, got[17], want[17]);
  }
  return 0;
}
