// Config 5 forward, inference form: 8 waves per 64-point workgroup in two STAGGERED sets, transposed accumulators.
//
// Why: with 64 points per CU a 512x512 bf16 layer is an L2 weight stream (17 k cycles, ~14 TB/s chip-wide) followed by a
// VALU/LDS epilogue (8 k cycles in the 4-wave kernel of fused.hpp) -- the 256 MFMAs per SIMD are only 8.2 k.  Weaving the epilogue
// into the k-loop of ONE wave lost to the compiler's scheduling/spills.  Here the overlap comes from two wave sets per SIMD instead:
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
// slab).  No activation copies, no dropout, no mask bits: dsdf_decode / dsdf_decode_latent (the training form stays with the
// 4-wave kernels, see the kernel's comment below).
// Specification: oracle decoder_forward(bf16=True); the k-units are contracted in another order than in the 4-wave kernel
// (own set's units first), which only permutes the fp32 accumulation.
#pragma once
#include "fused.hpp"

namespace dsdf {

constexpr int F8_THREADS = 512;
#ifndef BF8_RING_UNITS
#define BF8_RING_UNITS 6
#endif
constexpr int BF8_RING = BF8_RING_UNITS;   // multiple of 3; k-units of weights in flight per wave + 1 (2 KiB each)

// k-units of a layer input split by producer set: unit u (input columns 16u .. 16u+15) lies in n-tile u/2 of the previous layer;
// phase 0 = the units of the even tiles (set A's), phase 1 = those of the odd tiles (set B's).
__device__ __forceinline__ int bf8_count(int nu, int ph) {
  const int r = nu & 3;
  return 2 * (nu >> 2) + (ph == 0 ? min(r, 2) : max(r - 2, 0));
}
// PHASE-MAJOR order of the contraction index, used by BOTH operands so that the k-loop's cursors are plain counters: the slab keeps
// feature f (32-column tile t = f >> 5) in column bf8_col(f): even tiles in columns 0..255, odd tiles in 256..511; Wfb keeps k-unit u
// of an n-tile in slot ((u >> 1) & 1) * 16 + 2 * (u >> 2) + (u & 1) of 32 (wn_tiles_kernel).  Slot s of phase p is s = 16 p + j.
constexpr int BF8_HALF = 16;       // slots per phase
__device__ __forceinline__ int bf8_tile_slot(int t) { return (t & 1) * 8 + (t >> 1); }          // slab tile (32 columns) of n-tile t
__device__ __forceinline__ int bf8_col(int f) { return 32 * bf8_tile_slot(f >> 5) + (f & 31); }

template <int... I, class F>
__device__ __forceinline__ void bf8_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

struct Bf8View { __amdgpu_buffer_rsrc_t rsrc; int tb[2]; int voff; };
__device__ __forceinline__ Bf8View bf8_view(const __bf16* wfb, int ntiles, int t0, int lane) {
  Bf8View v;
  v.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wfb, 0, (ntiles * 32) << 10, 0x00020000);   // exact size: nothing past the layer's copy is ever read
  v.tb[0] = t0 * 32; v.tb[1] = (t0 + 8) * 32;
  v.voff = lane * 16;
  return v;
}

// acc[m][j] += W[n-tile j of this wave][units of phase ph] * X^T[.][rows 32m ..]: D = W X^T, lane (fh, fr) register 4g+i =
// Y[row 32m + fr][feature 32 t_j + 8g + 4fh + i].  Loads past the last unit are dropped by the buffer bounds check (the vector
// offset is pushed out of range): a wrapped-around prefetch would cost real L2 bandwidth, which is the bound here.
//
// REGISTER REUSE DISTANCE (measured, tools/lab_bf16x8_err.py): with TWO waves per SIMD an MFMA may still be reading its A/B source
// registers when a later-issued ds_read (or an out-of-range buffer load, which returns at once) writes them: the first version let
// the allocator give a load the registers the MFMA just in front of it had read -- fine for a lone wave per SIMD (the 4-wave kernel
// runs the same pattern bit-reproducibly), but here 3 % of the rows came out different from run to run, always rows 48-63 of a
// workgroup (the late columns of the LAST MFMA of a group).  So a step's loads are issued AFTER its MFMAs, they only ever replace
// what the step BEFORE consumed, and the fragments the step itself consumed are kept live (empty asm) until those loads are out:
// the allocator cannot hand their registers to a load before a full step (2 NT MFMAs) has passed.
// Code-object evidence (tools/lab/mfma_reuse_audit.py = deepsdf_amd/asmcheck.py mfma_src_reuse_distances, on the rebuilt first version
// 5f67ff3 and on this one): v1 has 24 ds_read_b128 and 22 buffer_load_dwordx4 whose destination is srcA / srcB of the MFMA issued
// straight in front of them (no other MFMA, no branch in between); this version has none.  What it still has are the GUARDED steps
// (bubbles at the phase boundary, items past the end): their MFMAs are skipped but their loadB is not, so a dropped (out-of-range,
// i.e. immediately returning) load lands in the ring slot the LAST real step's MFMAs read with no MFMA in between.  Those steps
// do not issue loads that would be dropped anyway, and first read the last accumulator on the VALU (bf8_drain: the read completes
// only when the MFMA that writes it has finished, and an MFMA that has finished has read its sources) -- what is left in them are
// real weight loads of the next phase (a full L2 round trip behind an MFMA that is already done).  asmcheck.check_mfma_src_reuse
// enforces the first property on every build: on no straight-line path may a load overwrite srcA / srcB of an MFMA with fewer than
// one other MFMA in between.
template <int NT>
__device__ __forceinline__ void bf8_keep(const bf16x8 (&a)[2], const bf16x8 (&b)[2]) {
  asm volatile("" ::"v"(a[0]), "v"(a[1]));
#pragma unroll
  for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(b[j]));
}
// A compiler-emitted VALU read (v_readfirstlane) of the accumulator the step's LAST MFMA writes: the compiler pads the MFMA-result ->
// VALU-read hazard itself (an inline-asm v_mov would not be padded), so the read completes only when that MFMA -- and, the matrix pipe
// being in-order, every MFMA before it -- is done.
template <int NT>
__device__ __forceinline__ void bf8_drain(const f32x16 (&acc)[2][2]) {
  const int done = __builtin_amdgcn_readfirstlane(__float_as_int(acc[1][NT - 1][15]));
  asm volatile("" ::"s"(done));
}
// One pass over ALL k-units of a layer: first those of phase `first`, then the others.  (No per-workgroup rotation of the unit order
// as in the fp32 kernel: the two sets are half a layer apart anyway, and the cursor arithmetic it needs costs more scalar
// instructions per step than the loop has room for.)
// The ring is indexed by the ITEM number n (slot n % R, row-fragment buffer n % 3, both compile-time inside a chunk of R steps).
// BAR (set A): the workgroup barrier sits between the two phases -- the second phase's input columns are being written by set B's
// epilogue until then, so row fragments are not read ahead across it; the WEIGHT ring does not care and keeps streaming: phase 1 is
// padded with bubbles (items without a unit: nothing loaded, nothing multiplied) up to a multiple of R, so that phase 2 starts on
// ring slot 0 again with its first R-1 units already requested during the last steps of phase 1.
template <int NT, bool BAR>
__device__ __forceinline__ void bf8_kloop(f32x16 (&acc)[2][2], const __bf16* ap, const Bf8View& B, int nu, int first) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  constexpr int R = BF8_RING;
  static_assert(R % 3 == 0, "the A fragments rotate through 3 buffers");
  const int c0 = bf8_count(nu, first), c1 = bf8_count(nu, 1 - first), T = c0 + c1;
  const int P = BAR ? ((c0 + R - 1) / R) * R : c0;      // first item of phase 2
  const int N = P + c1;                                 // items in all
  bf16x8 ring[R][2];
  bf16x8 a[3][2];
  // item n of the load side / unit k of the row side -> slot (16 phase + rotated position): a handful of scalar selects, no branches
  int bn = 0, ak = 0;
  const int base0 = first * BF8_HALF, base1 = (1 - first) * BF8_HALF - c0;
  auto slot_of = [&](int k) __attribute__((always_inline)) { return k + (k >= c0 ? base1 : base0); };   // k-th unit of the sequence (0 .. T-1)
  // item bn: a unit, a bubble, or past the end.  The last two load nothing: in the steady-state steps as a DROPPED load (vector offset
  // out of range -- no branch in the loop body; it returns at once, which is harmless there because the step's own MFMAs sit between
  // the MFMAs that read the slot and this write); in the guarded steps (skipc), whose own MFMAs may be missing, not issued at all.
  auto loadB = [&](bf16x8 (&dst)[2], auto skipc) __attribute__((always_inline)) {
    const bool real = (bn < c0 || bn >= P) && bn < N;
    const int sl = slot_of(bn < c0 ? bn : bn - P + c0);
    const int vo = real ? B.voff : (int)0xFFFFFF00u;    // past num_records by any reading of the range check
    if (!decltype(skipc)::value || real) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(B.rsrc, vo, (B.tb[j] + sl) << 10, 0);
        dst[j] = __builtin_bit_cast(bf16x8, r);
      }
    }
    ++bn;
  };
  auto readA = [&](bf16x8 (&x)[2]) __attribute__((always_inline)) {   // (past the end: the last unit again -- valid data, never used)
    const int sl = slot_of(ak);
    x[0] = *reinterpret_cast<const bf16x8*>(ap + 16 * sl);
    x[1] = *reinterpret_cast<const bf16x8*>(ap + 32 * FLDH + 16 * sl);
    ak = ak + 1 < T ? ak + 1 : ak;
  };
  auto mma = [&](const bf16x8 (&x)[2], const bf16x8 (&b)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], x[0], acc[0][j], 0, 0, 0);   // D = W X^T
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], x[1], acc[1][j], 0, 0, 0);
    }
  };
  // one step with compile-time buffer indices; GUARD: phase-1 boundary chunk / tail (items may be bubbles or past the end)
  auto step = [&](int n, auto qc, auto guardc, int lim_mma, int lim_pre) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    constexpr bool GUARD = decltype(guardc)::value;
    if (!GUARD || n < lim_mma) mma(a[q % 3], ring[q]);
    else bf8_drain<NT>(acc);                                   // no MFMA in this step: the last real step's MFMAs have read their
    __builtin_amdgcn_sched_barrier(0);                         // sources before this step's loads overwrite them
    if (!GUARD || n + 2 < lim_pre) readA(a[(q + 2) % 3]);      // rows of item n+2 -> the buffer step n-1 consumed
    loadB(ring[(q + R - 1) % R], guardc);                      // weights of item n+R-1 -> the slot step n-1 consumed
    __builtin_amdgcn_sched_barrier(0);
    bf8_keep<NT>(a[q % 3], ring[q]);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto chunk = [&](int s, auto guardc, int lim_mma, int lim_pre) __attribute__((always_inline)) {
    bf8_static_for(std::make_integer_sequence<int, R>{},
                   [&](auto qc) __attribute__((always_inline)) { step(s + decltype(qc)::value, qc, guardc, lim_mma, lim_pre); });
  };
#pragma unroll
  for (int q = 0; q < R - 1; ++q) loadB(ring[q], std::false_type{});
  int s = 0;
  if (BAR) {
    if (0 < c0) readA(a[0]);
    if (1 < c0) readA(a[1]);
    for (; s + R + 2 <= c0; s += R) chunk(s, std::false_type{}, 0, 0);     // every step multiplies and reads two steps ahead
    for (; s < P; s += R) chunk(s, std::true_type{}, c0, c0);               // the boundary: last units, bubbles, no reading ahead
    __syncthreads();                 // set B's epilogue is complete: its columns (phase 2) may be read
    readA(a[0]);
    readA(a[1]);
  } else {
    readA(a[0]);
    readA(a[1]);
  }
  for (; s + R <= N; s += R) chunk(s, std::false_type{}, 0, 0);
  if (s < N) chunk(s, std::true_type{}, N, N + 2);   // tail (reading ahead past the end is harmless)
  // leaving: whatever comes next (the epilogue's table reads) loads into fresh registers at once -- wait until the LAST MFMA has
  // written back (a VALU read of its result), only then let go of the fragment registers
  const int done = __builtin_amdgcn_readfirstlane(__float_as_int(acc[1][NT - 1][15]));   // (compiler-emitted: it pads the MFMA -> VALU hazard)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 3; ++q) asm volatile("" ::"v"(a[q][0]), "v"(a[q][1]), "s"(done));
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(ring[q][j]));
  __builtin_amdgcn_sched_barrier(0);
}
// every wave of set A takes the mid-layer barrier exactly once per layer, with or without work
template <bool BAR>
__device__ __forceinline__ void bf8_kloop_dispatch(f32x16 (&acc)[2][2], const __bf16* ap, const Bf8View& B, int nu, int first, int nt) {
  if (nt == 2 && nu > 0) bf8_kloop<2, BAR>(acc, ap, B, nu, first);
  else if (nt == 1 && nu > 0) bf8_kloop<1, BAR>(acc, ap, B, nu, first);
  else if (BAR) __syncthreads();
}

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

// Everything the epilogue of one layer needs besides the accumulators (wave-uniform).
struct Bf8Epi {
  __bf16* OUT;            // next layer's input slab
  int od, odp;            // out_dim; columns this layer owns in the slab (out_dim, or rounded up to the k-unit when no x0 columns
                          // follow: the pad is written as zeros -- features past out_dim come out as exactly 0 by themselves)
  const float* btab;      // LDS: bias per column (0 past out_dim)
};

// One n-tile: bias + ReLU on the lane's 2 x 16 values = rows 32m + fr, features 32t + 8g + 4fh + i.
//   !LASTL: -> bf16 -> OUT slab, 4 features per 8-byte store;   LASTL: the output layer's dot product instead (part[m])
// FULL: the tile ends inside the layer's columns -- no checks at all; the one ragged tile of a layer takes the guarded stores.
template <bool FULL, bool LASTL>
__device__ __forceinline__ void bf8_epilogue_tile(const f32x16& a0, const f32x16& a1, const Bf8Epi& E, const float (&wl)[16],
                                                  float (&part)[2], int t, int lane) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int f0 = 32 * t + 8 * g + 4 * fh;
    const float4 b4 = *reinterpret_cast<const float4*>(E.btab + f0);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x16& a = m == 0 ? a0 : a1;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(a[4 * g + i] + bb[i], 0.f);
      if constexpr (LASTL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) part[m] = fmaf(v[i], wl[4 * g + i], part[m]);   // (wl is 0 past out_dim)
      } else {
        __bf16* dst = E.OUT + (32 * m + fr) * FLDH + 32 * bf8_tile_slot(t) + 8 * g + 4 * fh;   // (phase-major column of feature f0)
        if (FULL || f0 + 3 < E.odp) {
          const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *reinterpret_cast<bf16x4*>(dst) = h;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (f0 + i < E.odp) dst[i] = (__bf16)v[i];
        }
      }
    }
  }
}
template <int NT, bool LASTL>
__device__ __forceinline__ void bf8_epilogue(const f32x16 (&acc)[2][2], const Bf8Epi& E, const float (&wl)[2][16], float (&part)[2],
                                             int t0, int lane) {
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + 8 * j;
    if (32 * t + 32 <= E.od) bf8_epilogue_tile<true, LASTL>(acc[0][j], acc[1][j], E, wl[j], part, t, lane);
    else bf8_epilogue_tile<false, LASTL>(acc[0][j], acc[1][j], E, wl[j], part, t, lane);
  }
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
        S[r * FLDH + bf8_col(col0 + c)] = (__bf16)v[k];
      }
    }
  }
}

// Inference only (dsdf_decode / dsdf_decode_latent / the module's eval forward without copies): no activation copies, no dropout, no
// mask bits.  The training form keeps the 4-wave kernels of fused.hpp: with 268 MB of fp32 activation copies per forward the layer
// is bound by the store path, not by the weight stream, and neither form of this kernel's epilogue (16-byte stores of one row per
// lane; 4-byte or quad-transposed 16-byte stores of the untransposed layout) got its stores out faster than the 4-wave kernel's
// (measured: 139 / 175 us against 118 us).
__global__ __launch_bounds__(F8_THREADS, 1) void fused_forward_bf16x8_kernel(const FusedFwdArgs p) {
  __shared__ __attribute__((aligned(16))) __bf16 SLAB[2 * FROWS * FLDH];   // layer l reads slab l & 1 and writes the other
  __shared__ float4 xs[FROWS];
  __shared__ float hu[FHOIST][FMAXW];
  __shared__ float4 hwx[FHOIST][FMAXW];
  __shared__ float red[16][FROWS];     // output-layer partials: [wave][fh][row]
  __shared__ __attribute__((aligned(16))) float btab[FMAXW];      // bias of the layer a set is working on: each set writes and reads
                                                                  // only the columns of ITS n-tiles (see the schedule)
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
  float wl[2][16];
  float part[2] = {0.f, 0.f};
  // the column of the per-set table this thread fills: the k-th column of the set's 8 n-tiles
  const int tcol = 32 * (2 * ((tid & 255) >> 5) + set) + (tid & 31);

  // everything a wave needs before the k-loop of layer l: accumulators, its set's bias table; returns the wave's n-tile count
  auto begin = [&](int l) __attribute__((always_inline)) -> int {
    const FusedLayer& L = p.ly[l];
    const int nt = (32 * t0 < L.out_dim ? 1 : 0) + (32 * (t0 + 8) < L.out_dim ? 1 : 0);
    int hidx = -1;
    if (segm) hidx = l == p.seg.h[0].layer ? 0 : (l == p.seg.h[1].layer ? 1 : -1);
    if (hidx >= 0) bf8_hoist_init(acc, hu[hidx], hwx[hidx], xs, L.out_dim, t0, fr, fh);
    else bf8_zero(acc);
    btab[tcol] = tcol < L.out_dim ? L.bias[tcol] : 0.f;
    if (l + 1 == nh) bf8_load_vec(wl, p.w_last, min(L.out_dim, p.in_last), t0, fh);
    return nt;
  };
  auto kloop = [&](int l, int nt, auto barc) __attribute__((always_inline)) {   // barc: set A (mid-layer barrier inside) or set B
    constexpr bool BAR = decltype(barc)::value;
    const FusedLayer& L = p.ly[l];
    const int nu = (L.in + 15) >> 4;
    const Bf8View B = bf8_view(reinterpret_cast<const __bf16*>(L.wf), (L.out_dim + 31) >> 5, t0, lane);
    bf8_kloop_dispatch<BAR>(acc, SLAB + (l & 1) * (FROWS * FLDH) + fr * FLDH + 8 * fh, B, nu, BAR ? 0 : 1, nt);
  };
  auto epilogue = [&](int l, int nt) __attribute__((always_inline)) {
    if (nt == 0) return;
    const FusedLayer& L = p.ly[l];
    Bf8Epi E;
    E.OUT = SLAB + ((l + 1) & 1) * (FROWS * FLDH);
    E.od = L.out_dim;
    E.odp = L.x0_col >= 0 ? L.out_dim : ((L.out_dim + 15) & ~15);
    E.btab = btab;
    if (l + 1 == nh) {
      if (nt == 2) bf8_epilogue<2, true>(acc, E, wl, part, t0, lane);
      else bf8_epilogue<1, true>(acc, E, wl, part, t0, lane);
    } else {
      if (nt == 2) bf8_epilogue<2, false>(acc, E, wl, part, t0, lane);
      else bf8_epilogue<1, false>(acc, E, wl, part, t0, lane);
    }
  };

#ifdef DSDF_LAB
  // lab: s_memtime stamps of wave 0 (set A, slots 0..31) and wave 4 (set B, slots 32..63) of every workgroup, 4 per layer
  auto stamp = [&](int slot) __attribute__((always_inline)) {
    if (p.dbg && lane == 0 && (w & 3) == 0 && slot < 32) p.dbg[blockIdx.x * 64 + 32 * set + slot] = __builtin_amdgcn_s_memtime();
  };
#else
  auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
  if (set == 0) {
    for (int l = 0; l < nh; ++l) {
      stamp(4 * l);
      const int nt = begin(l);
      kloop(l, nt, std::true_type{});        // K_l(UA) | barrier: E_B(l-1) is complete, this set's tables of layer l are visible | K_l(UB)
      stamp(4 * l + 1);
      epilogue(l, nt);
      stamp(4 * l + 2);
      if (p.ly[l].x0_col >= 0 && l + 1 < nh)   // general mode, skip layer: x0 joins the next layer's input (columns no epilogue writes)
        bf8_load_x0<256>(SLAB + ((l + 1) & 1) * (FROWS * FLDH), p.x0, p.ldx0, p.W0, row0, p.N, p.ly[l].x0_col, tid);
      __syncthreads();                       // E_A(l) (+ x0) is complete
      stamp(4 * l + 3);
    }
    __syncthreads();
  } else {
    int nt_prev = 0;
    for (int l = 0; l <= nh; ++l) {
      stamp(4 * l);
      if (l > 0) epilogue(l - 1, nt_prev);
      stamp(4 * l + 1);
      __syncthreads();
      stamp(4 * l + 2);
      if (l < nh) {
        const int nt = begin(l);
        kloop(l, nt, std::false_type{});     // K_l(UB) K_l(UA)
        nt_prev = nt;
        stamp(4 * l + 3);
        __syncthreads();
      }
    }
  }
  red[2 * w + fh][fr] = part[0];        // (each slot has exactly one writer)
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
