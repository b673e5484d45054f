// Lab (not product): what does one wave per SIMD pay for the split-mode cut woven between v_mfma_f32_32x32x16_bf16?
// One group = 3 MFMAs + the 11 VALU instructions that cut one value pair into its three bf16 terms (fused.hpp SPLIT_GROUP).
// Variants of the group, all in asm volatile (emitted order = source order); prints shader cycles per group (s_memtime):
//   0 MFMAs only                               1 the product's group (one dependent chain)
//   2 the same instructions, no dependences     3 two pairs' chains interleaved (6 MFMAs + 22 VALU, counted per 3 MFMAs)
//   4 the group without the three v_cvt_pk (8 VALU: perm-free truncation skeleton)      5 4 VALU (v_and only) per MFMA
//   6 11 VALU first, then the 3 MFMAs (what the compiler's order amounts to)
//   7 the product's chain software-pipelined over three groups: stage 1 of pair g, stage 2 of pair g - 1, stage 3 of pair g - 2
// (variants 1, 3, 6, 7 carry 3 extra compiler-emitted v_xor per group for the sink: compare them with each other)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MF(C) "v_mfma_f32_32x32x16_bf16 %[" #C "], %[a], %[b], %[" #C "]\n"

template <int V>
__global__ __launch_bounds__(256, 1) void weave(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.0f + threadIdx.x * 1e-4f * j); }
  float x0 = threadIdx.x * 1.37e-3f + 0.11f, x1 = threadIdx.x * 2.11e-3f + 0.23f, y0 = x0 * 1.5f, y1 = x1 * 0.7f;
  unsigned h = 0, m = 0, l = 0, h2 = 0, m2 = 0, l2 = 0, t0, t1, u0, u1;
  float r0, r1, s0 = x0 * 0.3f, s1 = x1 * 0.3f;
  unsigned sink = 0;
  unsigned long long T0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; g += 3) {
      const int i0 = g % 8, i1 = (g + 1) % 8, i2 = (g + 2) % 8;
      if (V == 0) {
        asm volatile(MF(c0) MF(c1) MF(c2) : [c0] "+a"(acc[i0]), [c1] "+a"(acc[i1]), [c2] "+a"(acc[i2]) : [a] "v"(a), [b] "v"(b));
      } else if (V == 1 || V == 6) {
        if (V == 1)
          asm volatile(MF(c0) "v_cvt_pk_bf16_f32 %[h], %[x0], %[x1]\n v_lshlrev_b32 %[t0], 16, %[h]\n v_and_b32 %[t1], %[msk], %[h]\n v_sub_f32 %[r0], %[x0], %[t0]\n"
                       MF(c1) "v_sub_f32 %[r1], %[x1], %[t1]\n v_cvt_pk_bf16_f32 %[m], %[r0], %[r1]\n v_lshlrev_b32 %[t0], 16, %[m]\n v_and_b32 %[t1], %[msk], %[m]\n"
                       MF(c2) "v_sub_f32 %[r0], %[r0], %[t0]\n v_sub_f32 %[r1], %[r1], %[t1]\n v_cvt_pk_bf16_f32 %[l], %[r0], %[r1]\n"
                       : [c0] "+a"(acc[i0]), [c1] "+a"(acc[i1]), [c2] "+a"(acc[i2]), [h] "=&v"(h), [m] "=&v"(m), [l] "=&v"(l), [t0] "=&v"(t0), [t1] "=&v"(t1), [r0] "=&v"(r0), [r1] "=&v"(r1)
                       : [a] "v"(a), [b] "v"(b), [x0] "v"(x0), [x1] "v"(x1), [msk] "s"(0xFFFF0000u));
        else
          asm volatile("v_cvt_pk_bf16_f32 %[h], %[x0], %[x1]\n v_lshlrev_b32 %[t0], 16, %[h]\n v_and_b32 %[t1], %[msk], %[h]\n v_sub_f32 %[r0], %[x0], %[t0]\n"
                       "v_sub_f32 %[r1], %[x1], %[t1]\n v_cvt_pk_bf16_f32 %[m], %[r0], %[r1]\n v_lshlrev_b32 %[t0], 16, %[m]\n v_and_b32 %[t1], %[msk], %[m]\n"
                       "v_sub_f32 %[r0], %[r0], %[t0]\n v_sub_f32 %[r1], %[r1], %[t1]\n v_cvt_pk_bf16_f32 %[l], %[r0], %[r1]\n" MF(c0) MF(c1) MF(c2)
                       : [c0] "+a"(acc[i0]), [c1] "+a"(acc[i1]), [c2] "+a"(acc[i2]), [h] "=&v"(h), [m] "=&v"(m), [l] "=&v"(l), [t0] "=&v"(t0), [t1] "=&v"(t1), [r0] "=&v"(r0), [r1] "=&v"(r1)
                       : [a] "v"(a), [b] "v"(b), [x0] "v"(x0), [x1] "v"(x1), [msk] "s"(0xFFFF0000u));
        sink ^= h ^ m ^ l;
      } else if (V == 2) {   // same opcodes, every instruction reads only loop-invariant registers
        asm volatile(MF(c0) "v_cvt_pk_bf16_f32 %[h], %[x0], %[x1]\n v_lshlrev_b32 %[t0], 16, %[k]\n v_and_b32 %[t1], %[msk], %[k]\n v_sub_f32 %[r0], %[x0], %[x1]\n"
                     MF(c1) "v_sub_f32 %[r1], %[x1], %[x0]\n v_cvt_pk_bf16_f32 %[m], %[x1], %[x0]\n v_lshlrev_b32 %[u0], 16, %[k]\n v_and_b32 %[u1], %[msk], %[k]\n"
                     MF(c2) "v_sub_f32 %[s0], %[x0], %[x1]\n v_sub_f32 %[s1], %[x1], %[x0]\n v_cvt_pk_bf16_f32 %[l], %[x0], %[x0]\n"
                     : [c0] "+a"(acc[i0]), [c1] "+a"(acc[i1]), [c2] "+a"(acc[i2]), [h] "=&v"(h), [m] "=&v"(m), [l] "=&v"(l), [t0] "=&v"(t0), [t1] "=&v"(t1), [r0] "=&v"(r0), [r1] "=&v"(r1),
                       [u0] "=&v"(u0), [u1] "=&v"(u1), [s0] "=&v"(s0), [s1] "=&v"(s1)
                     : [a] "v"(a), [b] "v"(b), [x0] "v"(x0), [x1] "v"(x1), [k] "v"(threadIdx.x), [msk] "s"(0xFFFF0000u));
        sink ^= h ^ m ^ l ^ t0 ^ t1 ^ u0 ^ u1 ^ __float_as_uint(r0) ^ __float_as_uint(r1) ^ __float_as_uint(s0) ^ __float_as_uint(s1);
      } else if (V == 3) {   // two chains interleaved: 6 MFMAs + 22 VALU per statement -> reported per 3 MFMAs
        asm volatile(MF(c0) "v_cvt_pk_bf16_f32 %[h], %[x0], %[x1]\n v_cvt_pk_bf16_f32 %[h2], %[y0], %[y1]\n v_lshlrev_b32 %[t0], 16, %[h]\n v_and_b32 %[t1], %[msk], %[h]\n"
                     MF(c1) "v_lshlrev_b32 %[u0], 16, %[h2]\n v_and_b32 %[u1], %[msk], %[h2]\n v_sub_f32 %[r0], %[x0], %[t0]\n v_sub_f32 %[r1], %[x1], %[t1]\n"
                     MF(c2) "v_sub_f32 %[s0], %[y0], %[u0]\n v_sub_f32 %[s1], %[y1], %[u1]\n v_cvt_pk_bf16_f32 %[m], %[r0], %[r1]\n v_cvt_pk_bf16_f32 %[m2], %[s0], %[s1]\n"
                     MF(c0) "v_lshlrev_b32 %[t0], 16, %[m]\n v_and_b32 %[t1], %[msk], %[m]\n v_lshlrev_b32 %[u0], 16, %[m2]\n v_and_b32 %[u1], %[msk], %[m2]\n"
                     MF(c1) "v_sub_f32 %[r0], %[r0], %[t0]\n v_sub_f32 %[r1], %[r1], %[t1]\n v_sub_f32 %[s0], %[s0], %[u0]\n v_sub_f32 %[s1], %[s1], %[u1]\n"
                     MF(c2) "v_cvt_pk_bf16_f32 %[l], %[r0], %[r1]\n v_cvt_pk_bf16_f32 %[l2], %[s0], %[s1]\n"
                     : [c0] "+a"(acc[i0]), [c1] "+a"(acc[i1]), [c2] "+a"(acc[i2]), [h] "=&v"(h), [m] "=&v"(m), [l] "=&v"(l), [h2] "=&v"(h2), [m2] "=&v"(m2), [l2] "=&v"(l2),
                       [t0] "=&v"(t0), [t1] "=&v"(t1), [r0] "=&v"(r0), [r1] "=&v"(r1), [u0] "=&v"(u0), [u1] "=&v"(u1), [s0] "=&v"(s0), [s1] "=&v"(s1)
                     : [a] "v"(a), [b] "v"(b), [x0] "v"(x0), [x1] "v"(x1), [y0] "v"(y0), [y1] "v"(y1), [msk] "s"(0xFFFF0000u));
        sink ^= h ^ m ^ l ^ h2 ^ m2 ^ l2;
      } else if (V == 7) {   // the chain software-pipelined across groups: no instruction depends on one of the same group's last three
        asm volatile(MF(c0) "v_cvt_pk_bf16_f32 %[h], %[x0], %[x1]\n v_cvt_pk_bf16_f32 %[m], %[pr0], %[pr1]\n v_cvt_pk_bf16_f32 %[l], %[pt0], %[pt1]\n v_lshlrev_b32 %[t0], 16, %[h]\n"
                     MF(c1) "v_and_b32 %[t1], %[msk], %[h]\n v_lshlrev_b32 %[u0], 16, %[m]\n v_and_b32 %[u1], %[msk], %[m]\n v_sub_f32 %[r0], %[x0], %[t0]\n"
                     MF(c2) "v_sub_f32 %[r1], %[x1], %[t1]\n v_sub_f32 %[pr0], %[pr0], %[u0]\n v_sub_f32 %[pr1], %[pr1], %[u1]\n"
                     : [c0] "+a"(acc[i0]), [c1] "+a"(acc[i1]), [c2] "+a"(acc[i2]), [h] "=&v"(h), [m] "=&v"(m), [l] "=&v"(l), [t0] "=&v"(t0), [t1] "=&v"(t1), [u0] "=&v"(u0), [u1] "=&v"(u1),
                       [r0] "=&v"(r0), [r1] "=&v"(r1), [pr0] "+v"(s0), [pr1] "+v"(s1)
                     : [a] "v"(a), [b] "v"(b), [x0] "v"(x0), [x1] "v"(x1), [pt0] "v"(y0), [pt1] "v"(y1), [msk] "s"(0xFFFF0000u));
        sink ^= h ^ m ^ l;
        y0 = s0; y1 = s1; s0 = r0; s1 = r1;      // pair g's r becomes "previous r", the finished t becomes "previous t" (renaming)
      } else if (V == 4) {   // no v_cvt_pk: 8 simple VALU in the same positions
        asm volatile(MF(c0) "v_lshlrev_b32 %[t0], 16, %[k]\n v_and_b32 %[t1], %[msk], %[k]\n v_sub_f32 %[r0], %[x0], %[t0]\n"
                     MF(c1) "v_sub_f32 %[r1], %[x1], %[t1]\n v_lshlrev_b32 %[t0], 16, %[r0]\n v_and_b32 %[t1], %[msk], %[r1]\n"
                     MF(c2) "v_sub_f32 %[r0], %[r0], %[t0]\n v_sub_f32 %[r1], %[r1], %[t1]\n"
                     : [c0] "+a"(acc[i0]), [c1] "+a"(acc[i1]), [c2] "+a"(acc[i2]), [t0] "=&v"(t0), [t1] "=&v"(t1), [r0] "=&v"(r0), [r1] "=&v"(r1)
                     : [a] "v"(a), [b] "v"(b), [x0] "v"(x0), [x1] "v"(x1), [k] "v"(threadIdx.x), [msk] "s"(0xFFFF0000u));
        sink ^= __float_as_uint(r0) ^ __float_as_uint(r1);
      } else if (V == 5) {   // 4 independent v_and per MFMA
        asm volatile(MF(c0) "v_and_b32 %[t0], %[msk], %[k]\n v_and_b32 %[t1], %[msk], %[k]\n v_and_b32 %[u0], %[msk], %[k]\n v_and_b32 %[u1], %[msk], %[k]\n"
                     MF(c1) "v_and_b32 %[t0], %[msk], %[k]\n v_and_b32 %[t1], %[msk], %[k]\n v_and_b32 %[u0], %[msk], %[k]\n v_and_b32 %[u1], %[msk], %[k]\n"
                     MF(c2) "v_and_b32 %[t0], %[msk], %[k]\n v_and_b32 %[t1], %[msk], %[k]\n v_and_b32 %[u0], %[msk], %[k]\n v_and_b32 %[u1], %[msk], %[k]\n"
                     : [c0] "+a"(acc[i0]), [c1] "+a"(acc[i1]), [c2] "+a"(acc[i2]), [t0] "=&v"(t0), [t1] "=&v"(t1), [u0] "=&v"(u0), [u1] "=&v"(u1)
                     : [a] "v"(a), [b] "v"(b), [k] "v"(threadIdx.x), [msk] "s"(0xFFFF0000u));
        sink ^= t0 ^ t1 ^ u0 ^ u1;
      }
    }
  }
  unsigned long long T1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sink;
  if (threadIdx.x == 0) cyc[blockIdx.x] = T1 - T0;
}

template <int V>
void run(const char* what, float* out, unsigned long long* cyc, int nblk) {
  const int iters = 2000;
  hipLaunchKernelGGL(weave<V>, dim3(nblk), dim3(256), 0, 0, out, cyc, 10);
  hipLaunchKernelGGL(weave<V>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long* h = (unsigned long long*)malloc(nblk * 8);
  hipMemcpy(h, cyc, nblk * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < nblk; ++i) sum += (double)h[i];
  const double groups = (double)iters * 3 * (V == 3 ? 2 : 1);   // groups of 3 MFMAs per iteration (g = 0, 3, 6)
  printf("%-70s %7.1f cycles per 3 MFMAs (96 = the MFMAs alone)\n", what, sum / nblk / groups);
  free(h);
}

int main() {
  const int nblk = 256;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, nblk * 256 * 4); hipMalloc(&cyc, nblk * 8);
  run<0>("0: three MFMAs", out, cyc, nblk);
  run<1>("1: the product's group (M vvvv M vvvv M vvv, one dependent chain)", out, cyc, nblk);
  run<2>("2: same opcodes, no dependences", out, cyc, nblk);
  run<3>("3: two chains interleaved (6 MFMAs + 22 VALU)", out, cyc, nblk);
  run<4>("4: without the three v_cvt_pk_bf16_f32 (8 VALU)", out, cyc, nblk);
  run<5>("5: 4 independent v_and_b32 per MFMA (12 VALU)", out, cyc, nblk);
  run<6>("6: the 11 VALU first, then the 3 MFMAs", out, cyc, nblk);
  run<7>("7: the chain pipelined over three groups (same 11 VALU, none dependent)", out, cyc, nblk);
  return 0;
}
