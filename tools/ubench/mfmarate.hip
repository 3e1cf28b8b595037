// MFMA issue rate of ONE wave per SIMD on gfx950 by operand placement (v_mfma_f32_32x32x16_bf16):
//   0: accumulators in AGPRs (8 independent), A / B in VGPRs
//   1: accumulators in VGPRs (2 chains), A VGPR, B AGPR
//   2: the attention stream of attn_prefill_w64_kernel: S_A (VGPR acc, B in AGPR), S_B, O_A (AGPR acc), O_B, 8 O accumulators
//   3: as 2 with every accumulator in AGPRs
//   4: as 0 with ONE accumulator (dependent chain)
//   5: accumulators in VGPRs (8 independent), A / B in VGPRs
// build: hipcc --offload-arch=gfx950 -O3 -o bin/mfmarate mfmarate.hip ; run: bin/mfmarate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CL8(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
#define AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", CL8(1), CL8(2), CL8(3), CL8(4), CL8(5), CL8(6), CL8(7), CL8(8), CL8(9), CL8(10), \
    CL8(11), CL8(12), CL8(13), CL8(14), CL8(15), CL8(16), CL8(17), CL8(18), "a190", "a191"

template <int MODE, bool RANDOM>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  extern __shared__ char lds[];
  bf16x8 a, b;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + (unsigned)iters;
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u; const unsigned short ua = RANDOM ? (unsigned short)(0x3c00 | ((h >> 9) & 0x83ff)) : (unsigned short)0x3f80;
    h = h * 1664525u + 1013904223u; const unsigned short ub = RANDOM ? (unsigned short)(0x3c00 | ((h >> 9) & 0x83ff)) : (unsigned short)0x3f80;
    a[i] = __builtin_bit_cast(__bf16, ua); b[i] = __builtin_bit_cast(__bf16, ub);
  }
  if (RANDOM) { for (int r = 0; r < 4; ++r) { unsigned w = (h = h * 1664525u + 1013904223u) & 0x83ff83ffu | 0x3c003c00u; asm volatile("v_accvgpr_write_b32 a[128+%c1], %0" ::"v"(w), "i"(0) : AGPRS); } }
  f32x16 s0, s1, v2, v3, v4, v5, v6, v7;
  for (int i = 0; i < 16; ++i) { s0[i] = 0; s1[i] = 0; v2[i] = 0; v3[i] = 0; v4[i] = 0; v5[i] = 0; v6[i] = 0; v7[i] = 0; }
  asm volatile("" ::: AGPRS);
  float x0 = threadIdx.x * 1e-3f, x1 = 0.5f;
#define MA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 a[" #acc ":" #acc "+15], %0, %1, a[" #acc ":" #acc "+15]" ::"v"(a), "v"(b) : AGPRS)
#define MAQ(acc) asm volatile("v_mfma_f32_32x32x16_bf16 a[" #acc ":" #acc "+15], %0, a[128:131], a[" #acc ":" #acc "+15]" ::"v"(a) : AGPRS)
#define MVQ(accv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[128:131], %0" : "+v"(accv) : "v"(a) : AGPRS)
#define NOP asm volatile("s_nop 0" ::: AGPRS)
#define VAL asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %0" : "+v"(x0), "+v"(x1) :: AGPRS)
#define MVQ2(accv, br) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[" #br ":" #br "+3], %0" : "+v"(accv) : "v"(a) : AGPRS)
#define MVV(accv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv) : "v"(a), "v"(b) : AGPRS)
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { MA(0); MA(16); MA(32); MA(48); MA(64); MA(80); MA(96); MA(112); MA(0); MA(16); MA(32); MA(48); MA(64); MA(80); MA(96); MA(112); }
    if (MODE == 1) { for (int j = 0; j < 8; ++j) { MVQ(s0); MVQ(s1); } }
    if (MODE == 2) { MVQ(s0); MVQ(s1); MA(0); MA(64); MVQ(s0); MVQ(s1); MA(16); MA(80); MVQ(s0); MVQ(s1); MA(32); MA(96); MVQ(s0); MVQ(s1); MA(48); MA(112); }
    if (MODE == 6) { MVQ(s0); NOP; MVQ(s1); NOP; MA(0); NOP; MA(64); NOP; MVQ(s0); NOP; MVQ(s1); NOP; MA(16); NOP; MA(80); NOP; MVQ(s0); NOP; MVQ(s1); NOP; MA(32); NOP; MA(96); NOP; MVQ(s0); NOP; MVQ(s1); NOP; MA(48); NOP; MA(112); NOP; }
    if (MODE == 7) { MVQ2(s0, 128); MVQ2(s1, 160); MA(0); MA(64); MVQ2(s0, 132); MVQ2(s1, 164); MA(16); MA(80); MVQ2(s0, 136); MVQ2(s1, 168); MA(32); MA(96); MVQ2(s0, 140); MVQ2(s1, 172); MA(48); MA(112); }
    if (MODE == 8) { MVQ2(s0, 128); VAL; MVQ2(s1, 160); VAL; MA(0); VAL; MA(64); VAL; MVQ2(s0, 132); VAL; MVQ2(s1, 164); VAL; MA(16); VAL; MA(80); VAL; MVQ2(s0, 136); VAL; MVQ2(s1, 168); VAL; MA(32); VAL; MA(96); VAL; MVQ2(s0, 140); VAL; MVQ2(s1, 172); VAL; MA(48); VAL; MA(112); VAL; }
    if (MODE == 9 || MODE == 10 || MODE == 11) {   // 16 MFMAs of mode 7, then the end-of-phase code of the attention kernel
      MVQ2(s0, 128); MVQ2(s1, 160); MA(0); MA(64); MVQ2(s0, 132); MVQ2(s1, 164); MA(16); MA(80); MVQ2(s0, 136); MVQ2(s1, 168); MA(32); MA(96); MVQ2(s0, 140); MVQ2(s1, 172); MA(48); MA(112);
      if (MODE >= 10) {
        float ya = x0, yb = x0;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ya), "+v"(yb));
        x1 = fmaxf(ya, yb);
      }
      if (MODE == 11) {
        if (__builtin_amdgcn_ballot_w64(x1 > 1e30f) != 0) { x0 = x0 * 2.f; asm volatile("s_nop 15" ::: AGPRS); }
      }
    }
    if (MODE == 3) { MAQ(144); MAQ(160); MA(0); MA(64); MAQ(144); MAQ(160); MA(16); MA(80); MAQ(144); MAQ(160); MA(32); MA(96); MAQ(144); MAQ(160); MA(48); MA(112); }
    if (MODE == 4) { for (int j = 0; j < 16; ++j) MA(0); }
    if (MODE == 5) { MVV(s0); MVV(s1); MVV(v2); MVV(v3); MVV(v4); MVV(v5); MVV(v6); MVV(v7); MVV(s0); MVV(s1); MVV(v2); MVV(v3); MVV(v4); MVV(v5); MVV(v6); MVV(v7); }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: AGPRS);
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = x0 + x1 + s0[0] + s1[3] + v2[1] + v3[1] + v4[1] + v5[1] + v6[1] + v7[1];
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(t)::AGPRS);
  out[blockIdx.x * 256 + threadIdx.x] = r + t;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (16.f * iters);
}
template <int MODE, bool RANDOM>
void run(float* out, const char* what) {
  const int iters = 20000;
  hipFuncSetAttribute((const void*)k<MODE, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(256), dim3(256), 128 * 1024, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(256), dim3(256), 128 * 1024, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, out + (1 << 20), 4, hipMemcpyDeviceToHost);
  const double fl = 256.0 * 4 * 16.0 * iters * 32768.0;
  printf("mode %d random %d %-60s %6.1f memtime ticks / MFMA   %7.1f TF/s   %.1f ns / MFMA\n", MODE, (int)RANDOM, what, cyc, fl / ms / 1e9, ms * 1e6 / (16.0 * iters));
}
int main() {
  float* out; hipMalloc(&out, ((1 << 20) + 16) * 4);
  run<0, false>(out, "AGPR acc x8, A/B VGPR"); run<0, true>(out, "AGPR acc x8, A/B VGPR");
  run<1, false>(out, "VGPR acc x2 chains, B AGPR"); run<1, true>(out, "VGPR acc x2 chains, B AGPR");
  run<2, false>(out, "attention mix: S VGPR acc (B AGPR) + O AGPR acc"); run<2, true>(out, "attention mix: S VGPR acc (B AGPR) + O AGPR acc");
  run<3, false>(out, "attention mix, all acc AGPR"); run<3, true>(out, "attention mix, all acc AGPR");
  run<4, false>(out, "AGPR acc x1 (dependent chain)"); run<4, true>(out, "AGPR acc x1 (dependent chain)");
  run<5, false>(out, "VGPR acc x8, A/B VGPR"); run<5, true>(out, "VGPR acc x8, A/B VGPR");
  run<6, true>(out, "mode 2 + s_nop 0 after every MFMA");
  run<7, true>(out, "mode 2 with a different AGPR B operand per S MFMA");
  run<8, true>(out, "mode 7 + 4 VALU (fma fma exp add) after every MFMA");
  run<9, true>(out, "16 MFMAs of mode 7 per iteration (loop overhead)");
  run<10, true>(out, "... + permlane32 exchange + max");
  run<11, true>(out, "... + ballot + wave-uniform branch (not taken)");
  return 0;
}
