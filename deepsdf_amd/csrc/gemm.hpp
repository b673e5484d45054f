// gemm.hpp -- fp32 MFMA GEMMs of the decoder (gfx950).
//
//   gemm_nt_kernel : C[M,N] = A[M,K] * B[N,K]^T   forward layers (B = W) and dX (B = W^T), fused epilogues
//   gemm_tn_kernel : C[s][M,N] = sum_{k in chunk s} A[k,M]^T B[k,N]   dW = dP^T * X, split-K slabs
//
// Both: 128x128 block tile, 4 waves as 2x2, each wave a 64x64 sub-tile = 2x2 v_mfma_f32_32x32x2_f32
// accumulators (64 acc VGPRs), BK = 32, register-staged double-buffered LDS, 2 workgroups per CU.
// fp32 MFMA runs at 64 FLOP/clk/SIMD (= 1/16 of bf16), so one k-tile (64 MFMAs x 64 cycles per wave)
// hides its own staging with room to spare: the kernels are MFMA-issue bound by construction.
//
// MFMA operand maps (cdna_hip_programming.md section 3): A: lane l holds A[i=l&31][k=l>>5], B: lane l holds
// B[k=l>>5][j=l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#pragma once
#include "common.hpp"

namespace dsdf {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BK + 4;  // NT tile row stride (floats): 144-B rows -> ds_read_b128 conflict-free

enum { EPI_PLAIN = 0, EPI_FWD = 1, EPI_BWD = 2 };

struct NtArgs {
  const float* A; const float* B; float* C;
  int M, N, K;
  int lda, ldb, ldc;
  const float* bias;       // [N] or nullptr
  // EPI_FWD: v = relu(v + bias); dropout (hash) if drop_thr != 0
  uint32_t drop_key, drop_thr; float drop_scale; uint32_t row_offset; int relu;
  // EPI_BWD: cols < mask_cols: v = act > 0 ? v * mask_scale : 0 -> C, column partial sums -> colsum;
  //          cols >= mask_cols (skip part, no activation): v -> C2[:, col - mask_cols] if col - mask_cols < c2_cols
  const float* act; int ldact; float mask_scale; int mask_cols;
  float* C2; int ldc2; int c2_cols;
  float* colsum; int ldcs;  // [gridM][ldcs]
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const NtArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BM * LDT];  // 73,728 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntn = (p.N + BN - 1) / BN;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = id / ntn;
  const int bm = tm * BM, bn = (id % ntn) * BN;

  // global -> register staging map: 8 lanes cover one row's 128-B k-slab, 32 rows per pass, 4 passes
  const int sr = tid >> 3;
  const int sc = (tid & 7) * 4;
  const float* Ap = p.A + (size_t)(bm + sr) * p.lda + sc;
  const float* Bp = p.B + (size_t)(bn + sr) * p.ldb + sc;
  float4 ra[4], rb[4];

  auto gload = [&](int k0) {
    const int rem = p.K - (k0 + sc);  // valid floats in this lane's chunk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (rem > 0) {
        if (bm + sr + 32 * i < p.M) va = *reinterpret_cast<const float4*>(Ap + (size_t)(32 * i) * p.lda + k0);
        if (bn + sr + 32 * i < p.N) vb = *reinterpret_cast<const float4*>(Bp + (size_t)(32 * i) * p.ldb + k0);
        if (rem < 4) { zero_tail4(va, rem); zero_tail4(vb, rem); }
      }
      ra[i] = va; rb[i] = vb;
    }
  };
  auto sstore = [&](int buf) {
    float* As = smem + buf * (2 * BM * LDT);
    float* Bs = As + BM * LDT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(As + (sr + 32 * i) * LDT + sc) = ra[i];
      *reinterpret_cast<float4*>(Bs + (sr + 32 * i) * LDT + sc) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  sstore(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    // lane (fr, fh) reads 4 consecutive k at offset 8s + 4fh: MFMA e of group s contracts k = {8s+e, 8s+4+e}
    // (a permutation of k shared by A and B -- only the summation order changes)
    const float* As = smem + buf * (2 * BM * LDT) + (wm * 64 + fr) * LDT + 4 * fh;
    const float* Bs = smem + buf * (2 * BM * LDT) + BM * LDT + (wn * 64 + fr) * LDT + 4 * fh;
    const int ngrp = min(4, (p.K - kt * BK + 7) >> 3);  // skip all-zero k-groups of the K tail
    if (ngrp == 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 a0 = *reinterpret_cast<const float4*>(As + 8 * s);
        const float4 a1 = *reinterpret_cast<const float4*>(As + 32 * LDT + 8 * s);
        const float4 b0 = *reinterpret_cast<const float4*>(Bs + 8 * s);
        const float4 b1 = *reinterpret_cast<const float4*>(Bs + 32 * LDT + 8 * s);
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
        const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
        }
      }
    } else {
      for (int s = 0; s < ngrp; ++s) {
        const float4 a0 = *reinterpret_cast<const float4*>(As + 8 * s);
        const float4 a1 = *reinterpret_cast<const float4*>(As + 32 * LDT + 8 * s);
        const float4 b0 = *reinterpret_cast<const float4*>(Bs + 8 * s);
        const float4 b1 = *reinterpret_cast<const float4*>(Bs + 32 * LDT + 8 * s);
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
        const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const int row_w = bm + wm * 64 + 4 * fh;  // + i*32 + (reg&3) + 8*(reg>>2)
  const int col_w = bn + wn * 64 + fr;      // + j*32

  if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col_w + 32 * j;
      const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_w + 32 * i + (r & 3) + 8 * (r >> 2);
          if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = acc[i][j][r] + bv;
        }
    }
  } else if constexpr (EPI == EPI_FWD) {
    const bool drop = p.drop_thr != 0u;
    const bool even = (p.row_offset & 1u) == 0u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col_w + 32 * j;
      const bool cok = col < p.N;
      const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
      const uint32_t ck = drop_col_key((uint32_t)col, p.drop_key);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {  // register pair (2rp, 2rp+1) = two consecutive rows
          const int r0 = 2 * rp;
          const int row = row_w + 32 * i + (r0 & 3) + 8 * (r0 >> 2);
          float v0 = acc[i][j][r0] + bv, v1 = acc[i][j][r0 + 1] + bv;
          if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          if (drop) {
            const uint32_t g0 = p.row_offset + (uint32_t)row;
            const uint32_t h0 = drop_pair_hash(ck, g0);
            const uint32_t h1 = even ? h0 : drop_pair_hash(ck, g0 + 1u);
            v0 = drop_keep(h0, g0, p.drop_thr) ? v0 * p.drop_scale : 0.f;
            v1 = drop_keep(h1, g0 + 1u, p.drop_thr) ? v1 * p.drop_scale : 0.f;
          }
          if (cok) {
            if (row < p.M) p.C[(size_t)row * p.ldc + col] = v0;
            if (row + 1 < p.M) p.C[(size_t)(row + 1) * p.ldc + col] = v1;
          }
        }
    }
  } else {  // EPI_BWD
    float csum[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col_w + 32 * j;
      if (col < p.mask_cols) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row_w + 32 * i + (r & 3) + 8 * (r >> 2);
            if (row < p.M) {
              const float a = p.act[(size_t)row * p.ldact + col];
              const float v = a > 0.f ? acc[i][j][r] * p.mask_scale : 0.f;
              p.C[(size_t)row * p.ldc + col] = v;
              csum[j] += v;
            }
          }
      } else if (p.C2 != nullptr && col < p.N && col - p.mask_cols < p.c2_cols) {
        const int c2 = col - p.mask_cols;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row_w + 32 * i + (r & 3) + 8 * (r >> 2);
            if (row < p.M) p.C2[(size_t)row * p.ldc2 + c2] = acc[i][j][r];
          }
      }
    }
    if (p.colsum != nullptr) {  // deterministic per-row-tile column sums (db partials)
      __syncthreads();
      float* red = smem;  // [2 (wm)][128]
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float s = csum[j] + __shfl_xor(csum[j], 32, 64);
        if (fh == 0) red[wm * 128 + wn * 64 + 32 * j + fr] = s;
      }
      __syncthreads();
      if (tid < 128) {
        const int col = bn + tid;
        if (col < p.mask_cols) p.colsum[(size_t)tm * p.ldcs + col] = red[tid] + red[128 + tid];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
struct TnArgs {
  const float* A; const float* B; float* C;  // A [K][lda] (M cols used), B [K][ldb] (N cols), C slabs
  int M, N, K;
  int lda, ldb, ldc;
  int kchunk;             // multiple of BK
  long long slab;         // floats between consecutive split slabs
};

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * 128];  // 65,536 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int T = ntm * ntn;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int split = id / T, t = id % T;
  const int bm = (t / ntn) * BM, bn = (t % ntn) * BN;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  const int kr = tid >> 5;          // 0..7 (+8i)
  const int c4 = (tid & 31) * 4;    // column offset inside the 128-wide tile
  const int rema = p.M - (bm + c4), remb = p.N - (bn + c4);
  const float* Ap = p.A + (size_t)(kbeg + kr) * p.lda + bm + c4;
  const float* Bp = p.B + (size_t)(kbeg + kr) * p.ldb + bn + c4;
  float4 ra[4], rb[4];

  auto gload = [&](int k0) {  // k0 relative to kbeg
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (kbeg + k0 + kr + 8 * i < kend) {
        if (rema > 0) { va = *reinterpret_cast<const float4*>(Ap + (size_t)(k0 + 8 * i) * p.lda); if (rema < 4) zero_tail4(va, rema); }
        if (remb > 0) { vb = *reinterpret_cast<const float4*>(Bp + (size_t)(k0 + 8 * i) * p.ldb); if (remb < 4) zero_tail4(vb, remb); }
      }
      ra[i] = va; rb[i] = vb;
    }
  };
  auto sstore = [&](int buf) {
    float* As = smem + buf * (2 * BK * 128);
    float* Bs = As + BK * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(As + (kr + 8 * i) * 128 + c4) = ra[i];
      *reinterpret_cast<float4*>(Bs + (kr + 8 * i) * 128 + c4) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    // tiles are [k][m]: lane (fr, fh) reads row 2kk+fh, column fr -> 32 consecutive floats per half-wave
    const float* As = smem + buf * (2 * BK * 128) + fh * 128 + wm * 64 + fr;
    const float* Bs = smem + buf * (2 * BK * 128) + BK * 128 + fh * 128 + wn * 64 + fr;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a0 = As[kk * 256], a1 = As[kk * 256 + 32];
      const float b0 = Bs[kk * 256], b1 = Bs[kk * 256 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  float* Cs = p.C + (size_t)split * p.slab;
  const int row_w = bm + wm * 64 + 4 * fh;
  const int col_w = bn + wn * 64 + fr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col_w + 32 * j;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_w + 32 * i + (r & 3) + 8 * (r >> 2);
        if (row < p.M && col < p.N) Cs[(size_t)row * p.ldc + col] = acc[i][j][r];
      }
  }
}

}  // namespace dsdf
