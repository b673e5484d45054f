// Config 5 forward, second form: 8 waves per 64-point workgroup in two STAGGERED sets, transposed accumulators.
//
// Why: with 64 points per CU a 512x512 bf16 layer is an L2 weight stream (17 k cycles, ~14 TB/s chip-wide) followed by a
// VALU/LDS epilogue (8-13 k cycles) -- the 256 MFMAs per SIMD are only 8.2 k.  Weaving the epilogue into the k-loop of ONE wave
// (fused.hpp BF_PIPELINED) lost to the compiler's scheduling/spills.  Here the overlap comes from two wave sets per SIMD instead:
//   set A (waves 0-3) owns the EVEN 32-column n-tiles, set B (waves 4-7) the ODD ones (a wave: up to 2 n-tiles x 64 rows = 64
//   accumulator registers).  B runs half a layer behind A:
//        A:  K_l(UA) | K_l(UB)  E_l     | K_l+1(UA) | K_l+1(UB)  E_l+1 | ...
//        B:  E_l-1   | K_l(UB)  K_l(UA) | E_l       | K_l+1(UB)  K_l+1(UA) | ...
//   (| = the workgroup barrier; K_l(UX) = the k-units of layer l whose input columns were produced by set X, E = epilogue).
//   A's epilogue runs while B streams weights and the other way round, so the L2 stream never stops; every dependence is covered
//   by the two barriers per layer: K_l(UA) needs E_A(l-1) (two barriers back), K_l(UB) needs E_B(l-1) (one barrier back), and
//   E_X(l) writes the OTHER slab than the one K_l reads (two bf16 slabs, as in fused_forward_bf16_body).
// Transposed accumulators: the WEIGHT fragment is the MFMA's A operand and the activation fragment its B operand (the fragment
// registers are the same either way), so a lane holds ONE point and 4 consecutive output features per register group: the
// epilogue converts 4 values and writes them with one 8-byte LDS store (the untransposed form needs 128 two-byte stores per lane
// and layer), and the 512 -> 1 output layer becomes a per-lane dot product folded into the last hidden layer's epilogue (no fp32
// slab).  Inference form (no activation copies, no dropout, no mask bits): dsdf_decode / dsdf_decode_latent.
// Specification: oracle decoder_forward(bf16=True); the k-units are contracted in another order than in the 4-wave kernel
// (own set's units first, rotated per workgroup), which only permutes the fp32 accumulation.
#pragma once
#include "fused.hpp"

namespace dsdf {

constexpr int F8_THREADS = 512;
#ifndef BF8_RING_UNITS
#define BF8_RING_UNITS 4
#endif
constexpr int BF8_RING = BF8_RING_UNITS;   // even; k-units of weights in flight per wave (2 KiB each)

// k-units of a layer input split by producer set: unit u (input columns 16u .. 16u+15) lies in n-tile u/2 of the previous layer
__device__ __forceinline__ int bf8_count(int nu, int ph) {
  const int r = nu & 3;
  return 2 * (nu >> 2) + (ph == 0 ? min(r, 2) : max(r - 2, 0));
}
__device__ __forceinline__ int bf8_unit(int j, int ph) { return 4 * (j >> 1) + 2 * ph + (j & 1); }

struct Bf8View { __amdgpu_buffer_rsrc_t rsrc; int tb[2]; int voff; };
__device__ __forceinline__ Bf8View bf8_view(const __bf16* wfb, int U, int ntiles, int t0, int lane) {
  Bf8View v;
  v.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wfb, 0, (ntiles * U) << 10, 0x00020000);   // exact size: nothing past the layer's copy is ever read
  v.tb[0] = t0 * U; v.tb[1] = (t0 + 8) * U;
  v.voff = lane * 16;
  return v;
}

// acc[m][j] += W[n-tile j of this wave][units of phase ph] * X^T[.][rows 32m ..]: D = W X^T, lane (fh, fr) register 4g+i =
// Y[row 32m + fr][feature 32 t_j + 8g + 4fh + i].  Loads past the last unit are dropped by the buffer bounds check (the vector
// offset is pushed out of range): a wrapped-around prefetch would cost real L2 bandwidth, which is the bound here.
template <int NT>
__device__ __forceinline__ void bf8_kloop(f32x16 (&acc)[2][2], const __bf16* ap, const Bf8View& B, int nu, int ph) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int cnt = bf8_count(nu, ph);
  if (cnt <= 0) return;
  const int rot = (int)((((unsigned)blockIdx.x >> 3) * (unsigned)cnt) >> 5) % cnt;   // blocks b, b+8, ... share an XCD
  bf16x8 ring[BF8_RING][2];
  bf16x8 a0[2], a1[2];
  int jb = rot, ib = 0, ja = rot;   // position (mod cnt) of the next unit to request / to read rows for; ib = units requested
  auto loadB = [&](bf16x8 (&dst)[2]) {
    const int u = bf8_unit(jb, ph);
    const int vo = ib < cnt ? B.voff : (int)0xFFFFFF00u;   // past num_records by any reading of the range check
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(B.rsrc, vo, (B.tb[j] + u) << 10, 0);
      dst[j] = __builtin_bit_cast(bf16x8, r);
    }
    jb = jb + 1 == cnt ? 0 : jb + 1; ++ib;
  };
  auto readA = [&](bf16x8 (&a)[2]) {
    const int u = bf8_unit(ja, ph);
    a[0] = *reinterpret_cast<const bf16x8*>(ap + 16 * u);
    a[1] = *reinterpret_cast<const bf16x8*>(ap + 32 * FLDH + 16 * u);
    ja = ja + 1 == cnt ? 0 : ja + 1;
  };
  auto mma = [&](const bf16x8 (&a)[2], const bf16x8 (&b)[2]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[0], acc[0][j], 0, 0, 0);
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[1], acc[1][j], 0, 0, 0);
    }
  };
#pragma unroll
  for (int q = 0; q < BF8_RING - 1; ++q) loadB(ring[q]);
  readA(a0);
  int s = 0;
  for (; s + BF8_RING <= cnt; s += BF8_RING) {
#pragma unroll
    for (int q = 0; q < BF8_RING; q += 2) {
      loadB(ring[(q + BF8_RING - 1) % BF8_RING]);
      readA(a1);
      mma(a0, ring[q]);
      loadB(ring[q % BF8_RING]);
      readA(a0);
      mma(a1, ring[q + 1]);
    }
  }
#pragma unroll
  for (int q = 0; q < BF8_RING - 1; ++q) {
    if (s + q < cnt) {
      if (q & 1) { readA(a0); mma(a1, ring[q]); }
      else { readA(a1); mma(a0, ring[q]); }
    }
  }
}
__device__ __forceinline__ void bf8_kloop_dispatch(f32x16 (&acc)[2][2], const __bf16* ap, const Bf8View& B, int nu, int ph, int nt) {
  if (nt == 2) bf8_kloop<2>(acc, ap, B, nu, ph);
  else if (nt == 1) bf8_kloop<1>(acc, ap, B, nu, ph);
}

// per-lane constants of a layer's epilogue: bias (and, for the last hidden layer, the output layer's weights) of the lane's
// features, requested before the k-loop
__device__ __forceinline__ void bf8_load_vec(float (&dst)[2][16], const float* src, int od, int t0, int fh) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = t0 + 8 * j;
    if (32 * t + 32 <= od) {          // whole tile inside: no per-element checks (wave-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[j][r] = src[32 * t + 8 * (r >> 2) + 4 * fh + (r & 3)];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = 32 * t + 8 * (r >> 2) + 4 * fh + (r & 3);
        dst[j][r] = f < od ? src[f] : 0.f;
      }
    }
  }
}

__device__ __forceinline__ void bf8_zero(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
}
// hoisted layer (segment mode): U_s[f] + <xyz[row], W[f, xyz]>, same expression and order as fused_hoist_init
__device__ __forceinline__ void bf8_hoist_init(f32x16 (&acc)[2][2], const float* hu, const float4* hwx, const float4* xs, int od,
                                               int t0, int fr, int fh) {
  const float4 x0 = xs[fr], x1 = xs[32 + fr];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = t0 + 8 * j;
    const bool full = 32 * t + 32 <= od;   // wave-uniform
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * t + 8 * (r >> 2) + 4 * fh + (r & 3);
      float ub = 0.f;
      float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
      if (full || f < od) { ub = hu[f]; wq = hwx[f]; }
      acc[0][j][r] = fmaf(x0.w, wq.w, fmaf(x0.z, wq.z, fmaf(x0.y, wq.y, fmaf(x0.x, wq.x, ub))));
      acc[1][j][r] = fmaf(x1.w, wq.w, fmaf(x1.z, wq.z, fmaf(x1.y, wq.y, fmaf(x1.x, wq.x, ub))));
    }
  }
}

// bias + ReLU -> bf16 -> OUT slab (row stride FLDH), 4 features per 8-byte store.  odp: columns this layer owns in the slab
// (out_dim, or out_dim rounded up to the k-unit when no x0 columns follow: the pad is written as zeros -- features past out_dim
// come out as exactly 0 by themselves: zero weight rows, zero bias, zero hoist).  A tile that ends inside the slab's owned
// columns needs no checks at all; the one ragged tile of a layer takes the element-wise path.
template <bool FULL>
__device__ __forceinline__ void bf8_epilogue_tile(const f32x16& a0, const f32x16& a1, const float (&bias)[16], __bf16* OUT, int odp,
                                                  int t, int fr, int fh) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int f0 = 32 * t + 8 * g + 4 * fh;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x16& a = m == 0 ? a0 : a1;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(a[4 * g + i] + bias[4 * g + i], 0.f);
      __bf16* dst = OUT + (32 * m + fr) * FLDH + f0;
      if (FULL || f0 + 3 < odp) {
        const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<bf16x4*>(dst) = h;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (f0 + i < odp) dst[i] = (__bf16)v[i];
      }
    }
  }
}
template <int NT>
__device__ __forceinline__ void bf8_epilogue(const f32x16 (&acc)[2][2], const float (&bias)[2][16], __bf16* OUT, int odp, int t0,
                                             int fr, int fh) {
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + 8 * j;
    if (32 * t + 32 <= odp) bf8_epilogue_tile<true>(acc[0][j], acc[1][j], bias[j], OUT, odp, t, fr, fh);
    else bf8_epilogue_tile<false>(acc[0][j], acc[1][j], bias[j], OUT, odp, t, fr, fh);
  }
}
// last hidden layer: its activation stays in registers and meets the output layer's weights there
template <int NT>
__device__ __forceinline__ void bf8_epilogue_last(const f32x16 (&acc)[2][2], const float (&bias)[2][16], const float (&wl)[2][16],
                                                  float (&part)[2]) {
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int m = 0; m < 2; ++m) part[m] = fmaf(fmaxf(acc[m][j][r] + bias[j][r], 0.f), wl[j][r], part[m]);   // (bias / wl are 0 past out_dim)
}

template <int NTHR>
__device__ __forceinline__ void bf8_load_x0(__bf16* S, const float* x0, int ldx0, int W0, int row0, int N, int col0, int tid) {
  constexpr int XCH = 12;
  const int zc = (((col0 + W0) + 15) & ~15) - col0;      // columns written incl. the zero pad
  const int total = FROWS * zc;
  for (int base = 0; base < total; base += NTHR * XCH) {
    float v[XCH];
#pragma unroll
    for (int k = 0; k < XCH; ++k) {
      const int i = base + tid + NTHR * k;
      v[k] = 0.f;
      if (i < total) {
        const int r = i / zc, c = i - r * zc;
        if (c < W0 && row0 + r < N) v[k] = x0[(size_t)(row0 + r) * ldx0 + c];
      }
    }
#pragma unroll
    for (int k = 0; k < XCH; ++k) {
      const int i = base + tid + NTHR * k;
      if (i < total) {
        const int r = i / zc, c = i - r * zc;
        S[r * FLDH + col0 + c] = (__bf16)v[k];
      }
    }
  }
}

__global__ __launch_bounds__(F8_THREADS, 1) void fused_forward_bf16x8_kernel(const FusedFwdArgs p) {
  __shared__ __attribute__((aligned(16))) __bf16 SLAB[2 * FROWS * FLDH];   // layer l reads slab l & 1 and writes the other
  __shared__ float4 xs[FROWS];
  __shared__ float hu[FHOIST][FMAXW];
  __shared__ float4 hwx[FHOIST][FMAXW];
  __shared__ float red[16][FROWS];     // output-layer partials: [wave][fh][row]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = w >> 2, t0 = 2 * (w & 3) + set;     // this wave's n-tiles: t0 and t0 + 8
  const int fr = lane & 31, fh = lane >> 5;
  const int row0 = blockIdx.x * FROWS;
  const bool segm = p.seg.wg_per_seg > 0;
  const int nh = p.n_hidden;

  if (segm) {
    if (tid < FROWS) {
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + tid < p.N) {
        const float* q = p.seg.xyz + (size_t)(row0 + tid) * p.seg.G;
        x.x = bf16_round(q[0]);
        if (p.seg.G > 1) x.y = bf16_round(q[1]);
        if (p.seg.G > 2) x.z = bf16_round(q[2]);
        if (p.seg.G > 3) x.w = bf16_round(q[3]);
      }
      xs[tid] = x;
    }
    const int sidx = blockIdx.x / p.seg.wg_per_seg;
#pragma unroll
    for (int t = 0; t < FHOIST; ++t) {
      const FusedHoist& H = p.seg.h[t];
      if (H.layer < 0) continue;
      const int od = p.ly[H.layer].out_dim;
      for (int c = tid; c < od; c += F8_THREADS) {
        hu[t][c] = p.seg.U[((size_t)sidx * FHOIST + t) * p.seg.ldu + c];
        const float* q = H.wx + (size_t)c * H.ldw;
        float4 x = make_float4(bf16_round(q[0]), 0.f, 0.f, 0.f);
        if (p.seg.G > 1) x.y = bf16_round(q[1]);
        if (p.seg.G > 2) x.z = bf16_round(q[2]);
        if (p.seg.G > 3) x.w = bf16_round(q[3]);
        hwx[t][c] = x;
      }
    }
  } else {
    bf8_load_x0<F8_THREADS>(SLAB, p.x0, p.ldx0, p.W0, row0, p.N, 0, tid);
  }
  __syncthreads();

  f32x16 acc[2][2];
  float bias[2][16], wl[2][16];
  float part[2] = {0.f, 0.f};
  int nt_prev = 0;

  // everything a wave needs before the k-loop of layer l: accumulators, epilogue constants
  auto begin = [&](int l) -> int {
    const FusedLayer& L = p.ly[l];
    const int nt = (32 * t0 < L.out_dim ? 1 : 0) + (32 * (t0 + 8) < L.out_dim ? 1 : 0);
    int hidx = -1;
    if (segm) hidx = l == p.seg.h[0].layer ? 0 : (l == p.seg.h[1].layer ? 1 : -1);
    if (hidx >= 0) bf8_hoist_init(acc, hu[hidx], hwx[hidx], xs, L.out_dim, t0, fr, fh);
    else bf8_zero(acc);
    bf8_load_vec(bias, L.bias, L.out_dim, t0, fh);
    if (l + 1 == nh) bf8_load_vec(wl, p.w_last, min(L.out_dim, p.in_last), t0, fh);
    return nt;
  };
  auto kloop = [&](int l, int ph, int nt) {
    const FusedLayer& L = p.ly[l];
    const int nu = (L.in + 15) >> 4;
    if (nu <= 0 || nt == 0) return;
    const Bf8View B = bf8_view(reinterpret_cast<const __bf16*>(L.wf), L.U, (L.out_dim + 31) >> 5, t0, lane);
    bf8_kloop_dispatch(acc, SLAB + (l & 1) * (FROWS * FLDH) + fr * FLDH + 8 * fh, B, nu, ph, nt);
  };
  auto epilogue = [&](int l, int nt) {
    const FusedLayer& L = p.ly[l];
    if (l + 1 == nh) {
      if (nt == 2) bf8_epilogue_last<2>(acc, bias, wl, part);
      else if (nt == 1) bf8_epilogue_last<1>(acc, bias, wl, part);
      return;
    }
    __bf16* OUT = SLAB + ((l + 1) & 1) * (FROWS * FLDH);
    const int odp = L.x0_col >= 0 ? L.out_dim : ((L.out_dim + 15) & ~15);
    if (nt == 2) bf8_epilogue<2>(acc, bias, OUT, odp, t0, fr, fh);
    else if (nt == 1) bf8_epilogue<1>(acc, bias, OUT, odp, t0, fr, fh);
  };

  if (set == 0) {
    for (int l = 0; l < nh; ++l) {
      const int nt = begin(l);
      kloop(l, 0, nt);
      __syncthreads();                       // E_B(l-1) is complete
      kloop(l, 1, nt);
      epilogue(l, nt);
      if (p.ly[l].x0_col >= 0 && l + 1 < nh)   // general mode, skip layer: x0 joins the next layer's input (columns no epilogue writes)
        bf8_load_x0<256>(SLAB + ((l + 1) & 1) * (FROWS * FLDH), p.x0, p.ldx0, p.W0, row0, p.N, p.ly[l].x0_col, tid);
      __syncthreads();                       // E_A(l) (+ x0) is complete
    }
    __syncthreads();
  } else {
    for (int l = 0; l < nh; ++l) {
      if (l > 0) epilogue(l - 1, nt_prev);
      __syncthreads();
      const int nt = begin(l);
      kloop(l, 1, nt);
      kloop(l, 0, nt);
      nt_prev = nt;
      __syncthreads();
    }
    epilogue(nh - 1, nt_prev);
    __syncthreads();
  }
  // (the barrier above orders nothing for `red`: each slot below has exactly one writer)
  red[2 * w + fh][fr] = part[0];
  red[2 * w + fh][32 + fr] = part[1];
  __syncthreads();
  if (tid < FROWS && row0 + tid < p.N) {
    float u = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) u += red[k][tid];     // fixed order: run-to-run bit-identical
    u += p.b_last[0];
    const float t1 = p.use_tanh ? tanhf(u) : u;
    if (p.y_out) p.y_out[row0 + tid] = tanhf(t1);
    if (p.u_out) p.u_out[row0 + tid] = u;
  }
}

}  // namespace dsdf
