// dwstream.hpp -- weight gradients of ALL layers in one launch (gfx950):  dW_l = dP_l^T a_l  (contraction over points).
//
// Register-streaming TN GEMM, no LDS: both operands are k-major ([point][column]), which is exactly the operand
// order of v_mfma_f32_32x32x2_f32 -- lane (r, h) of a wave needs A[k + h][m] and B[k + h][n].  Each lane loads ONE
// float4 of A (4 consecutive m) and ONE float4 of B (4 consecutive n) per k-step of 2 points; element i of the A
// vector feeds MFMA row-tile i (rows m0 + 4r + i), element j of B feeds column-tile j: 2 loads -> 16 MFMAs.  A wave
// owns a whole 128x128 output tile (4x4 accumulators = 256 AGPRs) over its own range of points; ONE wave per SIMD,
// a ring of 8 prefetched k-steps (64 VGPRs) hides the memory latency (tools/lab/mfma_peak.hip: 134 TFLOP/s).
// Work items (layer, K-split, tile) are laid out so that the tiles of one split run on one XCD and share the streamed
// rows through its L2.  Output: split-K slabs, summed in fixed order by finalize_layer_kernel (deterministic).
#pragma once
#include "common.hpp"
#include "fused.hpp"   // crow()

namespace dsdf {

struct DwLayer {
  const float* dp; int ld_dp;     // dP_l  [N][ld_dp], M = out_l columns used
  const float* act; int ld_act;   // a_l   [N][ld_act], Nc = in_l columns used
  float* slabs; long long slab;   // [nsplit][M][ldc]
  int M, Nc, ldc;
  int tiles_n, tiles;             // column tiles of 128, total tiles
  int nsplit, kchunk;             // K-splits and points per split (even)
  int item0;                      // first work item of this layer
};
struct DwArgs { int n_layers, n_items, N; DwLayer ly[DSDF_MAX_LAYERS]; };

constexpr int DW_RING = 8;

__global__ __launch_bounds__(256, 1) void dw_stream_kernel(const DwArgs p) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int nwaves = gridDim.x * 4;
  for (int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + w; item < p.n_items; item += nwaves) {
    int l = 0;
    while (l + 1 < p.n_layers && item >= p.ly[l + 1].item0) ++l;
    const DwLayer& L = p.ly[l];
    const int local = item - L.item0;
    const int split = local / L.tiles, tile = local - split * L.tiles;
    const int m0 = (tile / L.tiles_n) * 128, n0 = (tile % L.tiles_n) * 128;
    const int kbeg = split * L.kchunk;
    const int kend = min(p.N, kbeg + L.kchunk);
    const int nsteps = (kend - kbeg) >> 1;                    // full k-steps of 2 points
    const float* ap = L.dp + (size_t)(kbeg + fh) * L.ld_dp + m0 + 4 * fr;
    const float* bq = L.act + (size_t)(kbeg + fh) * L.ld_act + n0 + 4 * fr;
    const size_t astep = (size_t)2 * L.ld_dp, bstep = (size_t)2 * L.ld_act;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[DW_RING], rb[DW_RING];
    auto mma = [&](const float4& a, const float4& b) {
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    };
    // prologue: fill the ring (steps beyond nsteps are simply not loaded)
#pragma unroll
    for (int q = 0; q < DW_RING - 1; ++q) {
      if (q < nsteps) {
        ra[q] = *reinterpret_cast<const float4*>(ap + q * astep);
        rb[q] = *reinterpret_cast<const float4*>(bq + q * bstep);
      }
    }
    int s = 0;
    for (; s + DW_RING <= nsteps; s += DW_RING) {   // steady state: static ring slots, one new step in flight per MMA group
#pragma unroll
      for (int q = 0; q < DW_RING; ++q) {
        const int nxt = s + q + DW_RING - 1;
        if (nxt < nsteps) {
          ra[(q + DW_RING - 1) % DW_RING] = *reinterpret_cast<const float4*>(ap + (size_t)nxt * astep);
          rb[(q + DW_RING - 1) % DW_RING] = *reinterpret_cast<const float4*>(bq + (size_t)nxt * bstep);
        }
        mma(ra[q], rb[q]);
      }
    }
    // tail: the remaining (< DW_RING) full steps are already in ring slots 0..rem-1
#pragma unroll
    for (int q = 0; q < DW_RING - 1; ++q)
      if (s + q < nsteps) mma(ra[q], rb[q]);
    if ((kend - kbeg) & 1) {                                   // odd last point: lanes of the second half contribute zero
      const int k = kend - 1;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (fh == 0) {
        a = *reinterpret_cast<const float4*>(L.dp + (size_t)k * L.ld_dp + m0 + 4 * fr);
        b = *reinterpret_cast<const float4*>(L.act + (size_t)k * L.ld_act + n0 + 4 * fr);
      }
      mma(a, b);
    }

    // epilogue: acc[i][j][reg] = dW[m0 + 4 (crow(reg) + 4 fh) + i][n0 + 4 fr + j]  ->  one 16-byte store per (i, reg)
    float* slab = L.slabs + (size_t)split * L.slab;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, L.M * L.ldc * 4, 0x00020000);
    const int n = n0 + 4 * fr;
    const bool nok = n < L.ldc;                                // ldc % 4 == 0: the whole float4 is inside the row or not
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 4 * (crow(r) + 4 * fh) + i;
        const uint32_t voff = nok ? (uint32_t)((m * L.ldc + n) * 4) : 0x7FFFFFFFu;   // rows >= M fall outside the descriptor
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v = {__float_as_uint(acc[i][0][r]), __float_as_uint(acc[i][1][r]), __float_as_uint(acc[i][2][r]),
                   __float_as_uint(acc[i][3][r])};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0);
      }
  }
}

}  // namespace dsdf
