// Lab: what does a bare v_mfma_f32_32x32x2_f32 / 16x16x4 loop sustain on this chip?  (not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// like the fused kernel's inner loop: 8 A registers x 4 n-tiles x 8 B registers, all distinct, loop-invariant
__global__ __launch_bounds__(256, 2) void kregs(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[4][8];
  for (int j = 0; j < 8; ++j) { a[j] = out[threadIdx.x + 64 * j]; for (int i = 0; i < 4; ++i) b[i][j] = out[threadIdx.x + 1000 + 64 * (i * 8 + j)]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters / 2; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[i][j], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x + 100000] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// same, A streamed from LDS (2 x ds_read_b128 per 32 MFMAs) like the fused kernel
__global__ __launch_bounds__(256, 2) void klds(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float S[32 * 516];
  for (int i = threadIdx.x; i < 32 * 516; i += 256) S[i] = out[i] * 1e-3f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float b[4][8];
  for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) b[i][j] = out[threadIdx.x + 1000 + 64 * (i * 8 + j)];
  const int lane = threadIdx.x & 63;
  const float* ap = S + (lane & 31) * 516 + 8 * (lane >> 5);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters / 2; ++it) {
    const int u = it & 31;
    const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * u);
    const float4 a1 = *reinterpret_cast<const float4*>(ap + 16 * u + 4);
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[i][j], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x + 100000] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the fused kernel's k-loop in isolation: A from LDS, B streamed from a 1-MB L2-resident fragment buffer with
// one-unit-ahead prefetch into two named register sets; `layers` x 32 units, no barriers, no epilogue
__global__ __launch_bounds__(256, 2) void kstream(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float S[32 * 516];
  for (int i = threadIdx.x; i < 32 * 516; i += 256) S[i] = out[i] * 1e-3f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* ap = S + (lane & 31) * 516 + 8 * (lane >> 5);
  const float* wf = out + 65536;
  const float* bp[4];
  for (int ni = 0; ni < 4; ++ni) bp[ni] = wf + (size_t)(w + 4 * ni) * 32 * 512 + lane * 4;
  float4 b0[4][2], b1[4][2];
  auto loadB = [&](float4 (&b)[4][2], int u) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) { b[ni][0] = *reinterpret_cast<const float4*>(bp[ni] + (size_t)u * 512); b[ni][1] = *reinterpret_cast<const float4*>(bp[ni] + (size_t)u * 512 + 256); }
  };
  auto compute = [&](const float4 (&b)[4][2], int u) {
    const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * u);
    const float4 a1 = *reinterpret_cast<const float4*>(ap + 16 * u + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const float4 bq = b[ni][j >> 2];
        const float bv = (j & 3) == 0 ? bq.x : ((j & 3) == 1 ? bq.y : ((j & 3) == 2 ? bq.z : bq.w));
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv, acc[ni], 0, 0, 0);
      }
  };
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int layers = iters / 64;   // 32 units x 32 MFMAs = 1024 MFMAs per layer = 64 'iters' of 16
  for (int l = 0; l < layers; ++l) {
    loadB(b0, 0);
    for (int u = 0; u < 32; u += 2) {
      loadB(b1, u + 1);
      compute(b0, u);
      if (u + 2 < 32) loadB(b0, u + 2);
      compute(b1, u + 1);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x + 400000] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Candidate (i): 1 wave per SIMD, 64-row slab in LDS (2 m-tiles), 4 n-tiles per wave (acc 2x4x16 = 128 VGPRs), B streamed
// from L2 fragments with DEPTH units of prefetch; 512-thread-free: 4 waves per workgroup, 1 workgroup per CU.
template <int DEPTH>
__global__ __launch_bounds__(256, 1) void kslab64(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float S[64 * 516];
  for (int i = threadIdx.x; i < 64 * 516; i += 256) S[i] = out[i] * 1e-3f;
  __syncthreads();
  f32x16 acc[2][4];
  for (int m = 0; m < 2; ++m) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* ap = S + (lane & 31) * 516 + 8 * (lane >> 5);
  const float* wf = out + 65536;
  const float* bp[4];
  for (int ni = 0; ni < 4; ++ni) bp[ni] = wf + (size_t)(w + 4 * ni) * 32 * 512 + lane * 4;
  float4 b[DEPTH + 1][4][2];
  auto loadB = [&](int slot, int u) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) { b[slot][ni][0] = *reinterpret_cast<const float4*>(bp[ni] + (size_t)u * 512); b[slot][ni][1] = *reinterpret_cast<const float4*>(bp[ni] + (size_t)u * 512 + 256); }
  };
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int layers = iters / 128;   // 32 units x 64 MFMAs = 2048 MFMAs per layer
  for (int l = 0; l < layers; ++l) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) loadB(d, d);
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (u + DEPTH < 32) loadB((u + DEPTH) % (DEPTH + 1), u + DEPTH);
      const int slot = u % (DEPTH + 1);
      float4 a[2][2];
#pragma unroll
      for (int m = 0; m < 2; ++m) { a[m][0] = *reinterpret_cast<const float4*>(ap + m * 32 * 516 + 16 * u); a[m][1] = *reinterpret_cast<const float4*>(ap + m * 32 * 516 + 16 * u + 4); }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const float4 bq = b[slot][ni][j >> 2], aq = a[m][j >> 2];
            const float bv = (j & 3) == 0 ? bq.x : ((j & 3) == 1 ? bq.y : ((j & 3) == 2 ? bq.z : bq.w));
            const float av = (j & 3) == 0 ? aq.x : ((j & 3) == 1 ? aq.y : ((j & 3) == 2 ? aq.z : aq.w));
            acc[m][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][ni], 0, 0, 0);
          }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int m = 0; m < 2; ++m) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[m][i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x + 400000] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Candidate (iii): register-streaming TN (dW): each wave owns a 128x128 output tile (4x4 MFMA tiles = 256 acc VGPRs) over its
// own K range; A[k][128 m] and B[k][128 n] rows are loaded straight to registers (512 B per half-wave), no LDS.
template <int DEPTH>
__global__ __launch_bounds__(256, 1) void ktn(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[4][4];
  for (int m = 0; m < 4; ++m) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + w;                       // global wave id
  const int tile = gw & 15, split = gw >> 4;                // 16 tiles of a 512x512 dW, 64 K-splits of 256 points
  const float* A = out + 65536 + (size_t)(split * 256) * 512 + (tile >> 2) * 128;      // dP [16384][512]
  const float* B = out + 65536 + 16384 * 512 + (size_t)(split * 256) * 512 + (tile & 3) * 128;   // act [16384][512]
  const int fr = lane & 31, fh = lane >> 5;
  const float* ap = A + (size_t)fh * 512 + 4 * fr;
  const float* bq = B + (size_t)fh * 512 + 4 * fr;
  float4 ra[DEPTH + 1], rb[DEPTH + 1];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int layers = iters / 128;   // 128 k-steps x 16 MFMAs = 2048 MFMAs per layer
  for (int l = 0; l < layers; ++l) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { ra[d] = *reinterpret_cast<const float4*>(ap + (size_t)(2 * d) * 512); rb[d] = *reinterpret_cast<const float4*>(bq + (size_t)(2 * d) * 512); }
#pragma unroll 1
    for (int s0 = 0; s0 < 128; s0 += (DEPTH + 1)) {
#pragma unroll
      for (int q = 0; q < DEPTH + 1; ++q) {
        const int s = s0 + q;
        if (s < 128) {
          if (s + DEPTH < 128) {
            ra[(q + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const float4*>(ap + (size_t)(2 * (s + DEPTH)) * 512);
            rb[(q + DEPTH) % (DEPTH + 1)] = *reinterpret_cast<const float4*>(bq + (size_t)(2 * (s + DEPTH)) * 512);
          }
          const float av[4] = {ra[q].x, ra[q].y, ra[q].z, ra[q].w}, bv[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w};
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int m = 0; m < 4; ++m) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[m][i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x + 400000] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Candidate (i'): 64-row slab, 8 waves per workgroup (2 per SIMD, same workgroup), wave = 2 m-tiles x 2 n-tiles (64 acc),
// B depth-3 prefetch (4 named slots of 16 VGPRs), k-loop rolled in groups of 4 units.
__global__ __launch_bounds__(512, 2) void kslab64x8(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float S[64 * 516];
  for (int i = threadIdx.x; i < 64 * 516; i += 512) S[i] = out[i] * 1e-3f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int m = 0; m < 2; ++m) for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* ap = S + (lane & 31) * 516 + 8 * (lane >> 5);
  const float* wf = out + 65536;
  const float* bp[2];
  for (int ni = 0; ni < 2; ++ni) bp[ni] = wf + (size_t)(w + 8 * ni) * 32 * 512 + lane * 4;
  float4 b[4][2][2];
  auto loadB = [&](float4 (&bb)[2][2], int u) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { bb[ni][0] = *reinterpret_cast<const float4*>(bp[ni] + (size_t)u * 512); bb[ni][1] = *reinterpret_cast<const float4*>(bp[ni] + (size_t)u * 512 + 256); }
  };
  auto compute = [&](const float4 (&bb)[2][2], int u) {
    float4 a[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { a[m][0] = *reinterpret_cast<const float4*>(ap + m * 32 * 516 + 16 * u); a[m][1] = *reinterpret_cast<const float4*>(ap + m * 32 * 516 + 16 * u + 4); }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float4 bq = bb[ni][j >> 2], aq = a[m][j >> 2];
          const float bv = (j & 3) == 0 ? bq.x : ((j & 3) == 1 ? bq.y : ((j & 3) == 2 ? bq.z : bq.w));
          const float av = (j & 3) == 0 ? aq.x : ((j & 3) == 1 ? aq.y : ((j & 3) == 2 ? aq.z : aq.w));
          acc[m][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][ni], 0, 0, 0);
        }
  };
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int layers = iters / 64;   // 32 units x 32 MFMAs = 1024 MFMAs per wave per layer
  for (int l = 0; l < layers; ++l) {
    loadB(b[0], 0); loadB(b[1], 1); loadB(b[2], 2);
    for (int u = 0; u < 32; u += 4) {
      loadB(b[3], u + 3); compute(b[0], u);
      if (u + 4 < 32) loadB(b[0], u + 4); compute(b[1], u + 1);
      if (u + 5 < 32) loadB(b[1], u + 5); compute(b[2], u + 2);
      if (u + 6 < 32) loadB(b[2], u + 6); compute(b[3], u + 3);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int m = 0; m < 2; ++m) for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[m][i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x + 400000] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Candidate (i''): 64-row slab, 4 waves (ONE per SIMD), wave = 2 m-tiles x 4 n-tiles (128 acc VGPRs), B depth-1 (two named sets),
// capped at 256 arch VGPRs (launch_bounds 2) so the compiler does not shuffle through AGPRs.
__global__ __launch_bounds__(256, 2) void kslab64x4(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float S[64 * 516];
  for (int i = threadIdx.x; i < 64 * 516; i += 256) S[i] = out[i] * 1e-3f;
  __syncthreads();
  f32x16 acc[2][4];
  for (int m = 0; m < 2; ++m) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* ap = S + (lane & 31) * 516 + 8 * (lane >> 5);
  const float* wf = out + 65536;
  const float* bp[4];
  for (int ni = 0; ni < 4; ++ni) bp[ni] = wf + (size_t)(w + 4 * ni) * 32 * 512 + lane * 4;
  float4 b0[4][2], b1[4][2];
  auto loadB = [&](float4 (&bb)[4][2], int u) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) { bb[ni][0] = *reinterpret_cast<const float4*>(bp[ni] + (size_t)u * 512); bb[ni][1] = *reinterpret_cast<const float4*>(bp[ni] + (size_t)u * 512 + 256); }
  };
  auto compute = [&](const float4 (&bb)[4][2], int u) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * u + 4 * hf);
      const float4 a1 = *reinterpret_cast<const float4*>(ap + 32 * 516 + 16 * u + 4 * hf);
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const float4 bq = bb[ni][hf];
          const float bv = e == 0 ? bq.x : (e == 1 ? bq.y : (e == 2 ? bq.z : bq.w));
          acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv, acc[0][ni], 0, 0, 0);
          acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv, acc[1][ni], 0, 0, 0);
        }
    }
  };
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int layers = iters / 128;   // 32 units x 64 MFMAs = 2048 MFMAs per wave per layer
  for (int l = 0; l < layers; ++l) {
    loadB(b0, 0);
    for (int u = 0; u < 32; u += 2) {
      loadB(b1, u + 1); compute(b0, u);
      if (u + 2 < 32) loadB(b0, u + 2); compute(b1, u + 1);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int m = 0; m < 2; ++m) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[m][i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x + 400000] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int blocks, int iters, double flop_per_mfma) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 256 * 4 + 80000000); hipMemset(out, 0, blocks * 256 * 4 + 80000000); hipMalloc(&cyc, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  double nm = (double)iters * 16;
  printf("%-28s blocks=%4d  %8.1f us  %6.1f TFLOP/s   memtime-cycles per MFMA per wave = %.1f\n", name, blocks, ms * 1e3,
         nm * blocks * 4 * flop_per_mfma / (ms * 1e-3) / 1e12, (double)h[0] / nm);
  hipFree(out); hipFree(cyc);
}

template <typename K>
void runb(const char* name, K kern, int blocks, int threads, int iters, double flop_per_mfma) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * threads * 4 + 80000000); hipMemset(out, 0, blocks * threads * 4 + 80000000); hipMalloc(&cyc, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  double nm = (double)iters * 16;
  printf("%-28s blocks=%4d  %8.1f us  %6.1f TFLOP/s   memtime-cycles per MFMA per wave = %.1f  clock=%.2f GHz\n", name, blocks, ms * 1e3,
         nm * blocks * (threads / 64) * flop_per_mfma / (ms * 1e-3) / 1e12, (double)h[0] / nm, (double)h[0] / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int it = 20000;
  run("32x32x2 acc=4", k32<4>, 256, it, 4096.0);
  run("32x32x2 acc=4", k32<4>, 512, it, 4096.0);
  run("32x32x2 acc=2", k32<2>, 256, it, 4096.0);
  run("32x32x2 acc=1", k32<1>, 256, it, 4096.0);
  run("32x32x2 acc=1", k32<1>, 512, it, 4096.0);
  run("regs 8A x 4x8B", kregs, 256, it, 4096.0);
  run("regs 8A x 4x8B", kregs, 512, it, 4096.0);
  run("lds A + regs B", klds, 256, it, 4096.0);
  run("lds A + regs B", klds, 512, it, 4096.0);
  run("stream B(L2)+lds A", kstream, 256, 64 * 320, 4096.0);
  run("stream B(L2)+lds A", kstream, 512, 64 * 320, 4096.0);
  runb("slab64 8 waves depth3", kslab64x8, 256, 512, 64 * 320, 4096.0);
  runb("slab64 4 waves 1/SIMD", kslab64x4, 256, 256, 128 * 160, 4096.0);
  runb("stream32 1w/SIMD", kstream, 256, 256, 64 * 320, 4096.0);
  runb("stream32 2w/SIMD", kstream, 512, 256, 64 * 320, 4096.0);
  run("tn regstream depth3", ktn<3>, 256, 128 * 160, 4096.0);
  run("tn regstream depth7", ktn<7>, 256, 128 * 160, 4096.0);
  run("16x16x4 acc=4", k16<4>, 256, it, 2048.0);
  run("16x16x4 acc=8", k16<8>, 256, it, 2048.0);
  run("16x16x4 acc=8", k16<8>, 512, it, 2048.0);
  run("16x16x4 acc=16", k16<16>, 512, it, 2048.0);
  return 0;
}
