// Lab (not product): slope of the VALU price beside v_mfma_f32_32x32x16_bf16 (one wave per SIMD), and what changes it:
// accumulators in AGPRs or VGPRs, an SGPR operand or not, NV fillers per MFMA gap.  Cycles per MFMA (32 = the MFMA alone).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int KIND, bool AGPR>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.0f + threadIdx.x * 1e-4f * j); }
  unsigned x[8];
  float f[8];
  for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x * 2654435761u + j; f[j] = threadIdx.x * 1e-3f + j; }
  unsigned long long T0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int j = (i * NV + v) & 7;
        if (KIND == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[j]) : "s"(0xFFFF0FFFu));            // SGPR operand, in place
        else if (KIND == 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[j]) : "v"(x[(j + 4) & 7]));    // VGPR operands
        else if (KIND == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[j]) : "v"(f[(j + 4) & 7]));
        else if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[j]) : "v"(f[j]), "v"(f[(j + 1) & 7]));
        else if (KIND == 4) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[j]) : "v"(f[j]), "v"(f[(j + 1) & 7]), "s"(0x07060302u));
        else if (KIND == 5) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x[j]));
      }
    }
  }
  unsigned long long T1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int j = 0; j < 8; ++j) s += (float)x[j] + f[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = T1 - T0;
}

template <int NV, int KIND, bool AGPR>
double run(float* out, unsigned long long* cyc) {
  const int nblk = 256, iters = 2000;
  hipLaunchKernelGGL((k<NV, KIND, AGPR>), dim3(nblk), dim3(256), 0, 0, out, cyc, 10);
  hipLaunchKernelGGL((k<NV, KIND, AGPR>), dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  static unsigned long long h[256];
  (void)hipMemcpy(h, cyc, nblk * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < nblk; ++i) sum += (double)h[i];
  return sum / nblk / (iters * 8.0);
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
  printf("cycles per MFMA with NV fillers per gap:        NV=0    1     2     3     4     5     6\n");
#define ROW(NAME, KIND, AG) printf("%-44s %6.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f\n", NAME, run<0, KIND, AG>(out, cyc), run<1, KIND, AG>(out, cyc), \
    run<2, KIND, AG>(out, cyc), run<3, KIND, AG>(out, cyc), run<4, KIND, AG>(out, cyc), run<5, KIND, AG>(out, cyc), run<6, KIND, AG>(out, cyc));
  ROW("v_and_b32 (SGPR operand), acc in AGPRs", 0, true)
  ROW("v_and_b32 (SGPR operand), acc in VGPRs", 0, false)
  ROW("v_and_b32 (VGPR operands), acc in AGPRs", 1, true)
  ROW("v_sub_f32, acc in AGPRs", 2, true)
  ROW("v_cvt_pk_bf16_f32, acc in AGPRs", 3, true)
  ROW("v_perm_b32, acc in AGPRs", 4, true)
  ROW("v_lshlrev_b32, acc in AGPRs", 5, true)
  ROW("v_sub_f32, acc in VGPRs", 2, false)
  return 0;
}
