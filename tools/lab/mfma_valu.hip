// Lab (not product): do VALU / LDS instructions overlap with v_mfma_f32_32x32x2_f32 on gfx950?
//  intra<NV,KIND>: ONE wave per SIMD; every MFMA is followed by NV independent filler instructions of the same wave
//  cross:          TWO waves per SIMD; waves 0-3 run the MFMA stream, waves 4-7 a filler stream (or nothing)
// Prints shader cycles per MFMA (s_memtime) so the answer does not depend on the clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { K_FMA = 0, K_MULLO = 1, K_DSW = 2, K_CNDMASK = 3 };

template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void intra(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float x[16];
  unsigned xi[16];
  for (int j = 0; j < 16; ++j) { x[j] = a + j; xi[j] = threadIdx.x * 2654435761u + j; }
  const unsigned lp = threadIdx.x * 4;   // LDS byte address (lds is the only __shared__ array: base 0)
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int j = (i * NV + v) & 15;
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[j]) : "v"(a), "v"(b));
        else if (KIND == K_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(xi[j]) : "v"(xi[(j + 1) & 15] | 1u));
        else if (KIND == K_CNDMASK) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[j]) : "v"(b));
        else asm volatile("ds_write_b32 %0, %1" ::"v"(lp), "v"(x[j]) : "memory");
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (NV > 0) __builtin_amdgcn_sched_group_barrier(KIND == K_DSW ? 0x200 : 0x002, NV, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int j = 0; j < 16; ++j) s += x[j] + (float)xi[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// waves 0..3: MFMA stream (8 accumulators); waves 4..7: filler stream of `fill_iters` x 64 instructions
template <int KIND>
__global__ __launch_bounds__(512, 1) void cross(float* out, unsigned long long* cyc, int iters, int fill_iters) {
  __shared__ float lds[8192];
  const int w = threadIdx.x >> 6;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float s = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1;
  if (w < 4) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    float x[16];
    unsigned xi[16];
    for (int j = 0; j < 16; ++j) { x[j] = a + j; xi[j] = threadIdx.x * 2654435761u + j; }
    const unsigned lp = threadIdx.x * 4;   // LDS byte address (lds is the only __shared__ array: base 0)
    for (int it = 0; it < fill_iters; ++it) {
#pragma unroll
      for (int v = 0; v < 64; ++v) {
        const int j = v & 15;
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[j]) : "v"(a), "v"(b));
        else if (KIND == K_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(xi[j]) : "v"(xi[(j + 1) & 15] | 1u));
        else if (KIND == K_CNDMASK) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[j]) : "v"(b));
        else asm volatile("ds_write_b32 %0, %1" ::"v"(lp), "v"(x[j]) : "memory");
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < 16; ++j) s += x[j] + (float)xi[j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

static float* out; static unsigned long long* cyc; static unsigned long long h[256 * 8];

template <int NV, int KIND> void run_intra(const char* kind) {
  const int iters = 20000;
  hipLaunchKernelGGL((intra<NV, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, 200);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((intra<NV, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
  const double nm = 8.0 * iters;
  printf("intra  %-8s NV=%2d : %7.2f cyc/MFMA  (%6.1f TF, %.3f ms; s_memtime ticks are 100 MHz: x clock/100MHz)\n", kind, NV, c / nm,
         256.0 * 4 * nm * 4096 / (ms * 1e-3) / 1e12, ms);
}

template <int KIND> void run_cross(const char* kind, int fill_iters) {
  const int iters = 20000;
  hipLaunchKernelGGL((cross<KIND>), dim3(256), dim3(512), 0, 0, out, cyc, 200, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((cross<KIND>), dim3(256), dim3(512), 0, 0, out, cyc, iters, fill_iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
  double cm = 0, cf = 0; for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? cm : cf) += h[i * 8 + w];
  cm /= 1024; cf /= 1024;
  printf("cross  %-8s fill_iters=%6d : MFMA waves %9.0f ticks (%.2f ticks/MFMA), filler waves %9.0f ticks (%.3f ticks/inst), kernel %.3f ms\n",
         kind, fill_iters, cm, cm / (8.0 * iters), cf, fill_iters ? cf / (64.0 * fill_iters) : 0.0, ms);
}

int main() {
  hipMalloc(&out, 256 * 512 * 4 + 4096); hipMalloc(&cyc, 256 * 8 * 8);
  hipMemset(out, 0, 256 * 512 * 4);
  run_intra<0, K_FMA>("none");
  run_intra<1, K_FMA>("v_fma"); run_intra<2, K_FMA>("v_fma"); run_intra<4, K_FMA>("v_fma"); run_intra<8, K_FMA>("v_fma"); run_intra<12, K_FMA>("v_fma");
  run_intra<1, K_CNDMASK>("v_max"); run_intra<4, K_CNDMASK>("v_max"); run_intra<8, K_CNDMASK>("v_max");
  run_intra<1, K_MULLO>("mul_lo"); run_intra<2, K_MULLO>("mul_lo"); run_intra<4, K_MULLO>("mul_lo");
  run_intra<1, K_DSW>("ds_write"); run_intra<2, K_DSW>("ds_write"); run_intra<4, K_DSW>("ds_write");
  // cross: filler work sized to ~the MFMA stream's duration (160000 MFMAs x 64 cyc = 10.2 M cycles; v_fma ~4-8 cyc each)
  run_cross<K_FMA>("v_fma", 0);
  run_cross<K_FMA>("v_fma", 20000);
  run_cross<K_FMA>("v_fma", 40000);
  run_cross<K_MULLO>("mul_lo", 10000);
  run_cross<K_DSW>("ds_write", 20000);
  run_cross<K_CNDMASK>("v_max", 20000);
  // filler alone (MFMA waves idle): baseline cost per filler instruction
  {
    hipLaunchKernelGGL((cross<K_FMA>), dim3(256), dim3(512), 0, 0, out, cyc, 0, 20000); hipDeviceSynchronize();
    hipMemcpy(h, cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double cf = 0; for (int i = 0; i < 256; ++i) for (int w = 4; w < 8; ++w) cf += h[i * 8 + w];
    printf("filler alone v_fma: %.3f ticks/inst\n", cf / 1024 / (64.0 * 20000));
    hipLaunchKernelGGL((cross<K_MULLO>), dim3(256), dim3(512), 0, 0, out, cyc, 0, 10000); hipDeviceSynchronize();
    hipMemcpy(h, cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    cf = 0; for (int i = 0; i < 256; ++i) for (int w = 4; w < 8; ++w) cf += h[i * 8 + w];
    printf("filler alone mul_lo: %.3f ticks/inst\n", cf / 1024 / (64.0 * 10000));
    hipLaunchKernelGGL((cross<K_DSW>), dim3(256), dim3(512), 0, 0, out, cyc, 0, 20000); hipDeviceSynchronize();
    hipMemcpy(h, cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    cf = 0; for (int i = 0; i < 256; ++i) for (int w = 4; w < 8; ++w) cf += h[i * 8 + w];
    printf("filler alone ds_write: %.3f ticks/inst\n", cf / 1024 / (64.0 * 20000));
  }
  return 0;
}
