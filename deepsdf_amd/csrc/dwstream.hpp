// dwstream.hpp -- weight gradients of ALL layers in one launch (gfx950):  dW_l = dP_l^T a_l  (contraction over points).
//
// Register-streaming TN GEMM, no LDS: both operands are k-major ([point][column]), which is exactly the operand
// order of v_mfma_f32_32x32x2_f32 -- lane (r, h) of a wave needs A[k + h][m] and B[k + h][n].  Each lane loads ONE
// float4 of A (4 consecutive m) and ONE float4 of B (4 consecutive n) per k-step of 2 points; element i of the A
// vector feeds MFMA row-tile i (rows m0 + 4r + i), element j of B feeds column-tile j: 2 loads -> 16 MFMAs.  A wave
// owns a whole 128x128 output tile (4x4 accumulators = 256 AGPRs) over its own range of points; ONE wave per SIMD,
// a ring of 16 prefetched k-steps hides the memory latency; the streamed loads go through buffer resources with SCALAR
// k-step offsets and the wave index is made scalar, so no vector address arithmetic sits between the MFMAs (413 -> 388 us).
// Work items (layer, K-split, tile) are laid out so that the tiles of one split run on one XCD and share the streamed
// rows through its L2.  Output: split-K slabs, summed in fixed order by finalize_row (deterministic).  The workgroups the
// items leave idle run the post-backward roles of kernels.hpp meanwhile.
#pragma once
#include "common.hpp"
#include "fused.hpp"   // crow()
#include "kernels.hpp" // PostBwdArgs: the small post-backward roles ride on this launch's idle workgroups

namespace dsdf {

struct DwLayer {
  const float* dp; int ld_dp;     // dP_l  [N][ld_dp], M = out_l columns used
  const float* act; int ld_act;   // a_l   [N][ld_act], Nc = in_l columns used
  float* slabs; long long slab;   // [nsplit][M][ldc]
  int M, Nc, ldc;
  int tiles_m, tiles_n;           // row / column tiles of 128
  int last_nj;                    // 32-column sub-tiles in the LAST column tile (1..4); < 4 = a narrow (short) edge tile
  int nfull_n;                    // column tiles that are full width (tiles_n or tiles_n - 1)
  int nsplit, kchunk;             // K-splits and points per split (even)
  int full0, narrow0;             // first full-width / narrow work item of this layer
};
// Items: all full-width (128x128) items of all layers first, then the narrow edge items.  Full items go one per wave
// (round-robin beyond that); the narrow ones are dealt to the waves that got no full item in the last round.
struct DwArgs { int n_layers, n_full, n_narrow, N; DwLayer ly[DSDF_MAX_LAYERS];
                unsigned long long* dbg; };   // lab builds (-DDSDF_LAB): per-wave s_memrealtime stamps (100 MHz, chip-wide), else unused

#ifndef DW_RING_STEPS
#define DW_RING_STEPS 16
#endif
constexpr int DW_RING = DW_RING_STEPS;
#ifndef DW_BRANCHLESS
#define DW_BRANCHLESS 0      // measured: no difference (388 us both ways); the per-step overhead is not the branch
#endif

template <int NJ> struct DwVec;
template <> struct DwVec<1> { typedef float type; };
template <> struct DwVec<2> { typedef float2 type; };
template <> struct DwVec<3> { typedef float3 type; };
template <> struct DwVec<4> { typedef float4 type; };

// one work item: a 128 x (32 NJ) tile over points [kbeg, kend)
template <int NJ>
__device__ __forceinline__ void dw_item(const DwLayer& L, int split, int m0, int n0, int kbeg, int kend, int fr, int fh) {
  typedef typename DwVec<NJ>::type bvec;
  const int nsteps = (kend - kbeg) >> 1;                    // full k-steps of 2 points
  const float* bq = L.act + (size_t)(kbeg + fh) * L.ld_act + n0 + NJ * fr;   // narrow tiles (NJ < 4): plain loads
  const size_t bstep = (size_t)2 * L.ld_act;
  // the streamed loads go through buffer resources based at the item's first row: the address of k-step q is a SCALAR
  // offset (q * step bytes, < 2^31) plus a per-lane constant -- no vector address arithmetic between the MFMAs
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(L.dp + (size_t)kbeg * L.ld_dp), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(L.act + (size_t)kbeg * L.ld_act), 0, 0x7FFFFFFF, 0x00020000);
  const int voa = (fh * L.ld_dp + m0 + 4 * fr) * 4, vob = (fh * L.ld_act + n0 + NJ * fr) * 4;
  const int astepb = 8 * L.ld_dp, bstepb = 8 * L.ld_act;
  auto lda = [&](int q) -> float4 {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ra, voa, q * astepb, 0);
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
  };

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 rga[DW_RING];
  bvec rgb[DW_RING];
  auto ldb = [&](const float* q) -> bvec {
    if constexpr (NJ == 3) { bvec v; v.x = q[0]; v.y = q[1]; v.z = q[2]; return v; }   // rows are only 4-byte aligned for NJ*fr
    else return *reinterpret_cast<const bvec*>(q);
  };
  auto ldbq = [&](int q) -> bvec {            // k-step q of the B operand
    if constexpr (NJ == 4) {
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rb, vob, q * bstepb, 0);
      return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
    } else return ldb(bq + (size_t)q * bstep);
  };
  auto mma = [&](const float4& a, const bvec& b) {
    const float av[4] = {a.x, a.y, a.z, a.w};
    float bv[NJ];
    if constexpr (NJ == 1) bv[0] = b;
    else { bv[0] = b.x; bv[1] = b.y; if constexpr (NJ > 2) bv[2] = b.z; if constexpr (NJ > 3) bv[3] = b.w; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
  };
  // prologue: fill the ring (steps beyond nsteps are simply not loaded)
#pragma unroll
  for (int q = 0; q < DW_RING - 1; ++q) {
    if (q < nsteps) {
      rga[q] = lda(q);
      rgb[q] = ldbq(q);
    }
  }
  int s = 0;
  for (; s + DW_RING <= nsteps; s += DW_RING) {   // steady state: static ring slots, one new step in flight per MMA group
#pragma unroll
    for (int q = 0; q < DW_RING; ++q) {
#if DW_BRANCHLESS
      const int nxt = min(s + q + DW_RING - 1, nsteps - 1);   // past the end: re-request the last step (in bounds, never used) --
      rga[(q + DW_RING - 1) % DW_RING] = lda(nxt);            // a scalar min instead of a compare + branch per k-step
      rgb[(q + DW_RING - 1) % DW_RING] = ldbq(nxt);
#else
      const int nxt = s + q + DW_RING - 1;
      if (nxt < nsteps) {
        rga[(q + DW_RING - 1) % DW_RING] = lda(nxt);
        rgb[(q + DW_RING - 1) % DW_RING] = ldbq(nxt);
      }
#endif
      mma(rga[q], rgb[q]);
    }
  }
  // tail: the remaining (< DW_RING) full steps are already in ring slots 0..rem-1
#pragma unroll
  for (int q = 0; q < DW_RING - 1; ++q)
    if (s + q < nsteps) mma(rga[q], rgb[q]);
  if ((kend - kbeg) & 1) {                                   // odd last point: lanes of the second half contribute zero
    const int k = kend - 1;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    bvec b;
    if constexpr (NJ == 1) b = 0.f; else { b.x = 0.f; b.y = 0.f; if constexpr (NJ > 2) b.z = 0.f; if constexpr (NJ > 3) b.w = 0.f; }
    if (fh == 0) {
      a = *reinterpret_cast<const float4*>(L.dp + (size_t)k * L.ld_dp + m0 + 4 * fr);
      b = ldb(L.act + (size_t)k * L.ld_act + n0 + NJ * fr);
    }
    mma(a, b);
  }

  // epilogue: acc[i][j][reg] = dW[m0 + 4 (crow(reg) + 4 fh) + i][n0 + NJ fr + j]  ->  one NJ-float store per (i, reg)
  float* slab = L.slabs + (size_t)split * L.slab;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, L.M * L.ldc * 4, 0x00020000);
  const int n = n0 + NJ * fr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 4 * (crow(r) + 4 * fh) + i;       // rows >= M fall outside the descriptor and are dropped
      if constexpr (NJ == 4) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t voff = n < L.ldc ? (uint32_t)((m * L.ldc + n) * 4) : 0x7FFFFFFFu;   // ldc % 4 == 0
        u32x4 v = {__float_as_uint(acc[i][0][r]), __float_as_uint(acc[i][1][r]), __float_as_uint(acc[i][2][r]),
                   __float_as_uint(acc[i][3][r])};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const uint32_t voff = n + j < L.ldc ? (uint32_t)((m * L.ldc + n + j) * 4) : 0x7FFFFFFFu;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rs, voff, 0, 0);
        }
      }
    }
}

// ---- split mode (DsdfNet.gemm_split, fused.hpp fused_kloop_split): 6 bf16 MFMAs on 3-way cut fp32 operands per tile product -------
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive k (= POINTS here) of one column per lane, and the operands are k-major: a lane takes its
// 8 points x 4 consecutive columns per operand and 16-point step; component j of the 8 vectors IS the fragment of tile j.  At 2.7 x the
// MFMA rate the register ring of dw_item cannot cover the HBM latency any more (two attempts: 1315 us spilling, 447 us with two steps in
// flight), so the prefetch ring lives in LDS: the 4 waves of a workgroup own a 2 x 2 block of tiles of one split, i.e.
// TWO 128-column panels of each operand, and stream them with direct-to-LDS loads (buffer_load_dwordx4 ... lds: no registers in flight)
// into a ring of DWS_RING steps of 32 KB; one barrier per step publishes a step and frees the slot of the previous one.  Rows past the
// item's last point read as zero through the buffer bounds check (row offset in the VECTOR offset).  Accumulators as in dw_item<4>.
#ifndef DW_SPLIT_ONE_WAIT
#define DW_SPLIT_ONE_WAIT 0     // 1: the fragment reads end in ONE wait (no loads in flight across C++ code); measured in DESIGN.md 4.3
#endif
constexpr int DWS_RING = 4;
constexpr int DWS_SLOT = 2 * 16 * 256;      // floats per ring slot: A panel [16 points][256 columns], then the B panel

__device__ __forceinline__ Split3 split8v(const float (&x)[8]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  uint32_t hw[4], mw[4], lw[4];
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) cut_pair(x[2 * pr], x[2 * pr + 1], hw[pr], mw[pr], lw[pr]);      // (fused.hpp: round-to-nearest terms)
  Split3 s;
  s.h = __builtin_bit_cast(bf16x8, (u32x4){hw[0], hw[1], hw[2], hw[3]});
  s.m = __builtin_bit_cast(bf16x8, (u32x4){mw[0], mw[1], mw[2], mw[3]});
  s.l = __builtin_bit_cast(bf16x8, (u32x4){lw[0], lw[1], lw[2], lw[3]});
  return s;
}

// the workgroup's block: row tiles tmb, tmb + 1 and column tiles tnb, tnb + 1 of `split`; wave w takes (tmb + (w >> 1), tnb + (w & 1))
__device__ __forceinline__ void dw_block_split_v1(const DwLayer& L, int split, int tmb, int tnb, int kbeg, int kend, int w, int lane, float* ring) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int fr = lane & 31, fh = lane >> 5;
  const int npts = kend - kbeg, nsteps = (npts + 15) >> 4;      // steps of 16 points (uniform over the workgroup)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(L.dp + (size_t)kbeg * L.ld_dp), 0, npts * L.ld_dp * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(L.act + (size_t)kbeg * L.ld_act), 0, npts * L.ld_act * 4, 0x00020000);
  // producer side: this wave brings rows 4w .. 4w+3 of every step, one load instruction = one 256-column row of a panel (lane l:
  // columns 4l .. 4l+3 -> the row lands contiguously at the LDS pointer)
  const int pva = (tmb * 128 + 4 * lane) * 4, pvb = (tnb * 128 + 4 * lane) * 4;
  auto issue = [&](int st) __attribute__((always_inline)) {
    float* slot = ring + (st % DWS_RING) * DWS_SLOT;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * w + rr, pt = 16 * st + row;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(slot + row * 256), 16, pt * L.ld_dp * 4 + pva, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(slot + 4096 + row * 256), 16, pt * L.ld_act * 4 + pvb, 0, 0, 0);
    }
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int st = 0; st < DWS_RING - 1; ++st) issue(st);      // (steps past the end: every row out of range, zeros land in LDS)
  const int ca = (w >> 1) * 128 + 4 * fr, cb = 4096 + (w & 1) * 128 + 4 * fr;
  for (int st = 0; st < nsteps; ++st) {
    // this wave's loads of step st are the oldest 8 of the 8 (DWS_RING - 1) in flight
    __builtin_amdgcn_s_waitcnt(0x0F70 | (((8 * (DWS_RING - 2)) & 15)) | ((((8 * (DWS_RING - 2)) >> 4) & 3) << 14));   // vmcnt(16), nothing else
    __builtin_amdgcn_s_barrier();          // step st is complete in LDS; everybody is done reading step st - 1  (the bare barrier:
                                           // __syncthreads()' fence makes the compiler wait for ALL loads in flight, vmcnt(0))
    issue(st + DWS_RING - 1);              // ... whose slot takes step st + DWS_RING - 1
    // The fragment reads are inline asm ON PURPOSE: the compiler's wait-count pass cannot tell the ring slots apart and puts
    // s_waitcnt vmcnt(0) in front of any ds_read it sees -- i.e. it waits for the loads just issued, and the ring is worth nothing.
    // (Inline asm hides hazards from it as well, cf. fused_bf16x8's history: the lgkmcnt wait is part of the asm.)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 a[8], b[8];
    {
      typedef __attribute__((address_space(3))) const float* lds_cf;
      const float* slot = ring + (st % DWS_RING) * DWS_SLOT;
      const uint32_t aa = (uint32_t)(uintptr_t)(lds_cf)(slot + 8 * fh * 256 + ca), ab = (uint32_t)(uintptr_t)(lds_cf)(slot + 8 * fh * 256 + cb);
      asm volatile(
          "ds_read_b128 %0, %16\n ds_read_b128 %1, %16 offset:1024\n ds_read_b128 %2, %16 offset:2048\n ds_read_b128 %3, %16 offset:3072\n"
          "ds_read_b128 %4, %16 offset:4096\n ds_read_b128 %5, %16 offset:5120\n ds_read_b128 %6, %16 offset:6144\n ds_read_b128 %7, %16 offset:7168\n"
          "ds_read_b128 %8, %17\n ds_read_b128 %9, %17 offset:1024\n ds_read_b128 %10, %17 offset:2048\n ds_read_b128 %11, %17 offset:3072\n"
          "ds_read_b128 %12, %17 offset:4096\n ds_read_b128 %13, %17 offset:5120\n ds_read_b128 %14, %17 offset:6144\n ds_read_b128 %15, %17 offset:7168\n"
#if DW_SPLIT_ONE_WAIT
          "s_waitcnt lgkmcnt(0)"     // A/B switch (tools/lab_split.sh): everything has landed when the statement ends
#else
          "s_waitcnt lgkmcnt(8)"     // the A fragments are there; the B reads stay in flight while A is cut (second wait below)
#endif
          : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]),
            "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7])
          : "v"(aa), "v"(ab)
          : "memory");
    }
    Split3 sa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = i == 0 ? a[k].x : (i == 1 ? a[k].y : (i == 2 ? a[k].z : a[k].w));
      sa[i] = split8v(x);
    }
#if !DW_SPLIT_ONE_WAIT
    // (the operands tie the B registers to this wait: nothing that READS them can be scheduled in front of it.  What the operands
    // cannot express is that b[] is not yet valid between the two statements: a copy / AGPR move / spill of a B register placed
    // there by the register allocator would read stale data.  deepsdf_amd/asmcheck.py checks every build's code object for exactly
    // that -- no instruction in the window names a B destination, no scratch traffic -- and deepsdf_amd/build.py refuses the
    // library otherwise; -DDW_SPLIT_ONE_WAIT=1 is the variant without a window.)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = j == 0 ? b[k].x : (j == 1 ? b[k].y : (j == 2 ? b[k].z : b[k].w));
      const Split3 sb = split8v(x);
#define DW_PASS(AX, BX)                                                                                                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                          \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].AX, sb.BX, acc[i][j], 0, 0, 0);
      DW_PASS(l, h) DW_PASS(h, l) DW_PASS(m, m) DW_PASS(m, h) DW_PASS(h, m) DW_PASS(h, h)
#undef DW_PASS
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the loads issued past the last step have landed ...
  __syncthreads();                         // ... before anybody reuses the ring
  // epilogue: acc[i][j][reg] = dW[m0 + 4 (crow(reg) + 4 fh) + i][n0 + 4 fr + j]  ->  one 16-byte store per (i, reg)
  const int m0 = (tmb + (w >> 1)) * 128, n0 = (tnb + (w & 1)) * 128;
  float* slab = L.slabs + (size_t)split * L.slab;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, L.M * L.ldc * 4, 0x00020000);
  const int n = n0 + 4 * fr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int m = m0 + 4 * (crow(rg) + 4 * fh) + i;       // rows >= M fall outside the descriptor and are dropped
      const uint32_t voff = n < L.ldc ? (uint32_t)((m * L.ldc + n) * 4) : 0x7FFFFFFFu;   // ldc % 4 == 0
      u32x4 v = {__float_as_uint(acc[i][0][rg]), __float_as_uint(acc[i][1][rg]), __float_as_uint(acc[i][2][rg]),
                 __float_as_uint(acc[i][3][rg])};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0);
    }
}

// ---- the same block with the cut woven between the MFMAs (DW_SPLIT_WEAVE, default) ----------------------------------------------------
// dw_block_split_v1 above runs a step as  [16 LDS reads] [cut A: 176 VALU] [per column: cut B, 24 MFMAs]: a lone in-order wave overlaps
// none of the cut with its MFMAs (tools/lab/split_weave.hip: 11 VALU in front of 3 MFMAs = 140 cycles, woven between them 101-112), the
// kernel sat at 60 % MFMA-busy.  Here the step is software-pipelined by one step and written as 32 asm groups (fused.hpp SPLIT_GROUP_P:
// MFMA, 4 VALU, MFMA, 4 VALU, MFMA, 3 VALU), the MFMAs of step st woven with the cut of what comes next:
//   column block j (24 MFMAs = 8 groups):  groups 0-3 cut column (j + 1) of the B operand (column 0 of step st + 1 in block 3),
//                                          groups 4-7 cut row tile j of the A operand of step st + 1
// so every term is cut one block / one step before its first MFMA.  Registers: two sets of A terms (this step's / the next's, 2 x 48),
// two of one B column (2 x 12), the raw A rows of the next step (32) and the raw B rows of this and the next step (2 x 32).
// The raw rows of step st + 1 are read from the LDS ring at the top of step st (16 ds_read_b128 in one asm statement); their first use
// is group 4, which opens with the s_waitcnt lgkmcnt(0) -- groups 0-3 run in that window and name none of the 64 destination
// registers (deepsdf_amd/asmcheck.py checks exactly that on every build's code object).
#ifndef DW_SPLIT_WEAVE
#define DW_SPLIT_WEAVE 1
#endif
typedef float dw_f32x4 __attribute__((ext_vector_type(4)));
template <int J> __device__ __forceinline__ float dw_comp(const dw_f32x4& v) { return J == 0 ? v.x : (J == 1 ? v.y : (J == 2 ? v.z : v.w)); }
struct DwRaw { dw_f32x4 v[8]; };                 // 8 points x 4 consecutive columns of one operand panel

__device__ __forceinline__ void dw_block_split(const DwLayer& L, int split, int tmb, int tnb, int kbeg, int kend, int w, int lane, float* ring) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef __attribute__((address_space(3))) const float* lds_cf;
  const int fr = lane & 31, fh = lane >> 5;
  const int npts = kend - kbeg, nsteps = (npts + 15) >> 4;      // steps of 16 points (uniform over the workgroup)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(L.dp + (size_t)kbeg * L.ld_dp), 0, npts * L.ld_dp * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(L.act + (size_t)kbeg * L.ld_act), 0, npts * L.ld_act * 4, 0x00020000);
  const int pva = (tmb * 128 + 4 * lane) * 4, pvb = (tnb * 128 + 4 * lane) * 4;
  auto issue = [&](int st) __attribute__((always_inline)) {     // this wave brings rows 4w .. 4w+3 of step st's two panels (see v1)
    float* slot = ring + (st % DWS_RING) * DWS_SLOT;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * w + rr, pt = 16 * st + row;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(slot + row * 256), 16, pt * L.ld_dp * 4 + pva, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(slot + 4096 + row * 256), 16, pt * L.ld_act * 4 + pvb, 0, 0, 0);
    }
  };
  // the same, one row pair at a time and pinned behind the asm group in front of it (the row number passes through an empty
  // volatile asm): inside the loop the 8 DMA instructions of a step go out between the MFMA groups instead of as a cluster at the
  // top of the step, where each of them costs the idle MFMA pipe ~100 cycles (MI355X_MICROARCH.md: LDS-DMA issue cost)
  // (the two row strides are made opaque once: left as kernel-argument loads the compiler re-fetches them -- s_load + s_waitcnt
  // lgkmcnt(0), a full stall of the lone wave -- at every one of the four issue points of a step)
  int lda4 = L.ld_dp * 4, ldb4 = L.ld_act * 4;
  asm volatile("" : "+s"(lda4), "+s"(ldb4));
  auto issue_row = [&](int st, int rr) __attribute__((always_inline)) {
    float* slot = ring + (st % DWS_RING) * DWS_SLOT;
    int row = 4 * w + rr;
    asm volatile("" : "+s"(row));
    const int pt = 16 * st + row;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(slot + row * 256), 16, pt * lda4 + pva, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(slot + 4096 + row * 256), 16, pt * ldb4 + pvb, 0, 0, 0);
  };
  const int ca = (w >> 1) * 128 + 4 * fr, cb = 4096 + (w & 1) * 128 + 4 * fr;
  // 16 fragment reads of one step's slot -> raw A, raw B.  WAIT = 1: the statement ends in lgkmcnt(0); WAIT = 0: the reads stay in
  // flight, the caller's next DW_WAIT_LGKM closes the window (nothing in between may name a.v[] / b.v[])
#define DW_READ_RAW(ST, A, B, WAITSTR)                                                                                              \
  {                                                                                                                                 \
    const float* slot_ = ring + ((ST) % DWS_RING) * DWS_SLOT;                                                                       \
    const uint32_t aa_ = (uint32_t)(uintptr_t)(lds_cf)(slot_ + 8 * fh * 256 + ca), ab_ = (uint32_t)(uintptr_t)(lds_cf)(slot_ + 8 * fh * 256 + cb); \
    asm volatile(                                                                                                                   \
        "ds_read_b128 %0, %16\n ds_read_b128 %1, %16 offset:1024\n ds_read_b128 %2, %16 offset:2048\n ds_read_b128 %3, %16 offset:3072\n" \
        "ds_read_b128 %4, %16 offset:4096\n ds_read_b128 %5, %16 offset:5120\n ds_read_b128 %6, %16 offset:6144\n ds_read_b128 %7, %16 offset:7168\n" \
        "ds_read_b128 %8, %17\n ds_read_b128 %9, %17 offset:1024\n ds_read_b128 %10, %17 offset:2048\n ds_read_b128 %11, %17 offset:3072\n" \
        "ds_read_b128 %12, %17 offset:4096\n ds_read_b128 %13, %17 offset:5120\n ds_read_b128 %14, %17 offset:6144\n ds_read_b128 %15, %17 offset:7168\n" \
        WAITSTR                                                                                                                     \
        : "=&v"((A).v[0]), "=&v"((A).v[1]), "=&v"((A).v[2]), "=&v"((A).v[3]), "=&v"((A).v[4]), "=&v"((A).v[5]), "=&v"((A).v[6]), "=&v"((A).v[7]), \
          "=&v"((B).v[0]), "=&v"((B).v[1]), "=&v"((B).v[2]), "=&v"((B).v[3]), "=&v"((B).v[4]), "=&v"((B).v[5]), "=&v"((B).v[6]), "=&v"((B).v[7])  \
        : "v"(aa_), "v"(ab_)                                                                                                        \
        : "memory");                                                                                                                \
  }
#define DW_WAIT_LGKM(A, B)                                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                               \
               : "+v"((A).v[0]), "+v"((A).v[1]), "+v"((A).v[2]), "+v"((A).v[3]), "+v"((A).v[4]), "+v"((A).v[5]), "+v"((A).v[6]), "+v"((A).v[7]), \
                 "+v"((B).v[0]), "+v"((B).v[1]), "+v"((B).v[2]), "+v"((B).v[3]), "+v"((B).v[4]), "+v"((B).v[5]), "+v"((B).v[6]), "+v"((B).v[7]));
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int st = 0; st < DWS_RING - 1; ++st) issue(st);      // (steps past the end: every row out of range, zeros land in LDS)

  DwRaw an, b0, b1;                    // raw A of the next step; raw B of the even / odd steps
  Split3 sa0[4], sa1[4], sb0, sb1;     // A terms of the even / odd steps; one B column's terms, alternating
  // prologue = "step -1": the raw rows of step 0, their A terms and the terms of B column 0, cut by compiler-scheduled code
  __builtin_amdgcn_s_waitcnt(0x0F70 | (((8 * (DWS_RING - 2)) & 15)) | ((((8 * (DWS_RING - 2)) >> 4) & 3) << 14));   // vmcnt(16): step 0 has landed
  __builtin_amdgcn_s_barrier();
  issue(DWS_RING - 1);
  DW_READ_RAW(0, an, b0, "s_waitcnt lgkmcnt(0)")
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = i == 0 ? an.v[k].x : (i == 1 ? an.v[k].y : (i == 2 ? an.v[k].z : an.v[k].w));
    sa0[i] = split8v(x);
  }
  {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = b0.v[k].x;
    sb0 = split8v(x);
  }
  // one group: MFMAs 3g .. 3g+2 of the step (MFMA q: column q / 24, pass (q % 24) / 4 in the order h h, h m, m h, m m, h l, l h
  // (A term, B term: big terms first), row tile q % 4) + one link of the cut chain, software-pipelined over three groups (fused.hpp
  // SPLIT_GROUP_P): stage 1 of pair g, stage 2 of pair g - 1, stage 3 of pair g - 2.  Pair g: block j = g / 8; g % 8 < 4: pair g % 8 of
  // B column (j + 1) % 4 (of BN in block 3, else of BC); else pair g % 8 - 4 of row tile j of AN.  The last three words of a step's chain
  // (m of pair 31, l of pairs 30, 31 = words of the NEXT step's A terms of row tile 3) come out of the next step's groups 0 and 1,
  // early enough because a column's m terms are operands from its 9th MFMA on and its l terms from its 17th.
#define DW_SA_OF(SAC, q_) (((q_) % 24) / 4 == 5 ? SAC[(q_) % 4].l : (((q_) % 24) / 4 == 2 || ((q_) % 24) / 4 == 3) ? SAC[(q_) % 4].m : SAC[(q_) % 4].h)
#define DW_SBSEL(q_) ((((q_) / 24) & 1) ? sb1 : sb0)
#define DW_SB_OF(q_) (((q_) % 24) / 4 == 4 ? DW_SBSEL(q_).l : (((q_) % 24) / 4 == 1 || ((q_) % 24) / 4 == 3) ? DW_SBSEL(q_).m : DW_SBSEL(q_).h)
#define DW_X(g, e, AN, BC, BN)                                                                                                       \
  (((g) % 8) < 4 ? dw_comp<(((g) / 8) + 1) % 4>(((g) / 8) == 3 ? (BN).v[2 * ((g) % 8) + (e)] : (BC).v[2 * ((g) % 8) + (e)])            \
                 : dw_comp<(g) / 8>((AN).v[2 * (((g) % 8) - 4) + (e)]))
#define DW_MFMAS(g, SAC)                                                                                                             \
  acc[(3 * (g)) % 4][(3 * (g)) / 24], DW_SA_OF(SAC, 3 * (g)), DW_SB_OF(3 * (g)),                                                        \
  acc[(3 * (g) + 1) % 4][(3 * (g) + 1) / 24], DW_SA_OF(SAC, 3 * (g) + 1), DW_SB_OF(3 * (g) + 1),                                        \
  acc[(3 * (g) + 2) % 4][(3 * (g) + 2) / 24], DW_SA_OF(SAC, 3 * (g) + 2), DW_SB_OF(3 * (g) + 2)
  // even g: stage 1 -> (ra0, ra1), stage 2 on (rb0, rb1), stage 3 on (ta0, ta1); odd g the other way round (see fused.hpp)
#define DW_GE(g, SAC, AN, BC, BN, M_, L_)                                                                                            \
  SPLIT_GROUP_P(DW_MFMAS(g, SAC), DW_X(g, 0, AN, BC, BN), DW_X(g, 1, AN, BC, BN), hw_[(g) / 4][(g) % 4], ra0, ra1, rb0, rb1, M_, ta0, ta1, L_) \
  tb0 = rb0; tb1 = rb1;
#define DW_GO(g, SAC, AN, BC, BN, M_, L_)                                                                                            \
  SPLIT_GROUP_P(DW_MFMAS(g, SAC), DW_X(g, 0, AN, BC, BN), DW_X(g, 1, AN, BC, BN), hw_[(g) / 4][(g) % 4], rb0, rb1, ra0, ra1, M_, tb0, tb1, L_) \
  ta0 = ra0; ta1 = ra1;
#define DW_MW(g) mw_[((g) - 1) / 4][((g) - 1) % 4]
#define DW_LW(g) lw_[((g) - 2) / 4][((g) - 2) % 4]
#define DW_G2(g, SAC, AN, BC, BN) DW_GE(g, SAC, AN, BC, BN, DW_MW(g), DW_LW(g)) DW_GO((g) + 1, SAC, AN, BC, BN, DW_MW((g) + 1), DW_LW((g) + 1))
  // assemble the terms of one cut operand from its words (register renaming, no instructions)
#define DW_PACK(DST, q)                                                                                                              \
  DST.h = __builtin_bit_cast(bf16x8, (u32x4){hw_[q][0], hw_[q][1], hw_[q][2], hw_[q][3]});                                           \
  DST.m = __builtin_bit_cast(bf16x8, (u32x4){mw_[q][0], mw_[q][1], mw_[q][2], mw_[q][3]});                                           \
  DST.l = __builtin_bit_cast(bf16x8, (u32x4){lw_[q][0], lw_[q][1], lw_[q][2], lw_[q][3]});
  // one step: SAC = this step's A terms (its row tile 3 still owed three words), SAN = the next step's (written here), BC = this
  // step's raw B, BN = the next step's (read here).  The 8 DMA rows of step ST + 4 go out behind groups 9, 15, 21, 27.
#define DW_STEP(ST, SAC, SAN, BC, BN)                                                                                                \
  {                                                                                                                                 \
    uint32_t hw_[8][4], mw_[8][4], lw_[8][4], mlate_, llate0_, llate1_;                                                             \
    float ra0, ra1, rb0 = carry.r0, rb1 = carry.r1, ta0 = carry.t0, ta1 = carry.t1, tb0, tb1;                                       \
    __builtin_amdgcn_s_waitcnt(0x0F70 | (((8 * (DWS_RING - 2)) & 15)) | ((((8 * (DWS_RING - 2)) >> 4) & 3) << 14)); /* vmcnt(16): step ST + 1 has landed */ \
    __builtin_amdgcn_s_barrier();          /* ... for every wave; everybody has finished reading the slot of step ST, which takes step ST + 4 */ \
    DW_READ_RAW((ST) + 1, an, BN, "")      /* window: open until DW_WAIT_LGKM below */                                                \
    DW_GE(0, SAC, an, BC, BN, mlate_, llate0_)                                                                                       \
    DW_GO(1, SAC, an, BC, BN, DW_MW(1), llate1_)                                                                                     \
    SAC[3].m = bf16x8_set_word<3>(SAC[3].m, mlate_);                                                                                 \
    SAC[3].l = bf16x8_set_word<3>(bf16x8_set_word<2>(SAC[3].l, llate0_), llate1_);                                                   \
    DW_G2(2, SAC, an, BC, BN)                                                                                                        \
    DW_WAIT_LGKM(an, BN)                   /* groups 0-3 cut pairs of BC only */                                                      \
    DW_G2(4, SAC, an, BC, BN) DW_G2(6, SAC, an, BC, BN)                                                                              \
    DW_PACK(sb1, 0)                        /* column 1: its last word came out of group 5 */                                          \
    DW_GE(8, SAC, an, BC, BN, DW_MW(8), DW_LW(8)) DW_GO(9, SAC, an, BC, BN, DW_MW(9), DW_LW(9)) issue_row((ST) + DWS_RING, 0);        \
    DW_G2(10, SAC, an, BC, BN) DW_G2(12, SAC, an, BC, BN)                                                                            \
    DW_GE(14, SAC, an, BC, BN, DW_MW(14), DW_LW(14)) DW_GO(15, SAC, an, BC, BN, DW_MW(15), DW_LW(15)) issue_row((ST) + DWS_RING, 1); \
    DW_PACK(sb0, 2)                                                                                                                  \
    DW_G2(16, SAC, an, BC, BN) DW_G2(18, SAC, an, BC, BN)                                                                            \
    DW_GE(20, SAC, an, BC, BN, DW_MW(20), DW_LW(20)) DW_GO(21, SAC, an, BC, BN, DW_MW(21), DW_LW(21)) issue_row((ST) + DWS_RING, 2); \
    DW_G2(22, SAC, an, BC, BN)                                                                                                       \
    DW_PACK(sb1, 4)                                                                                                                  \
    DW_G2(24, SAC, an, BC, BN)                                                                                                       \
    DW_GE(26, SAC, an, BC, BN, DW_MW(26), DW_LW(26)) DW_GO(27, SAC, an, BC, BN, DW_MW(27), DW_LW(27)) issue_row((ST) + DWS_RING, 3); \
    DW_G2(28, SAC, an, BC, BN) DW_G2(30, SAC, an, BC, BN)                                                                            \
    carry.r0 = rb0; carry.r1 = rb1; carry.t0 = ta0; carry.t1 = ta1;    /* owed: r of pair 31 (odd: rb), t of pair 30 */               \
    mw_[7][3] = 0; lw_[7][2] = 0; lw_[7][3] = 0;                        /* (the next step's groups 0, 1 deliver these words) */        \
    DW_PACK(sb0, 6) DW_PACK(SAN[0], 1) DW_PACK(SAN[1], 3) DW_PACK(SAN[2], 5) DW_PACK(SAN[3], 7)                                      \
  }
  // the chain's carry as if step 0's A terms had come out of the groups: r of its last pair, t of the one before (row tile 3, points 4-7)
  SplitCarry carry;
  {
    uint32_t h_, m_, l_;
    float d0, d1;
    cut_pair_rt(an.v[4].w, an.v[5].w, h_, m_, l_, d0, d1, carry.t0, carry.t1);
    cut_pair_rt(an.v[6].w, an.v[7].w, h_, m_, l_, carry.r0, carry.r1, d0, d1);
  }
  // (compiler-written terms -> the first groups' MFMAs: held and padded as in fused_kloop_split_asm4)
  asm volatile("s_nop 1"
               : "+v"(sa0[0].h), "+v"(sa0[0].m), "+v"(sa0[0].l), "+v"(sa0[1].h), "+v"(sa0[1].m), "+v"(sa0[1].l), "+v"(sa0[2].h), "+v"(sa0[2].m),
                 "+v"(sa0[2].l), "+v"(sa0[3].h), "+v"(sa0[3].m), "+v"(sa0[3].l), "+v"(sb0.h), "+v"(sb0.m), "+v"(sb0.l));
  for (int st = 0; st < nsteps; st += 2) {
    DW_STEP(st, sa0, sa1, b0, b1)
    if (st + 1 >= nsteps) break;
    DW_STEP(st + 1, sa1, sa0, b1, b0)
  }
#undef DW_STEP
#undef DW_PACK
#undef DW_G2
#undef DW_LW
#undef DW_MW
#undef DW_GO
#undef DW_GE
#undef DW_MFMAS
#undef DW_X
#undef DW_SB_OF
#undef DW_SBSEL
#undef DW_SA_OF
#undef DW_WAIT_LGKM
#undef DW_READ_RAW
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the loads issued past the last step have landed ...
  __syncthreads();                         // ... before anybody reuses the ring
  // epilogue: acc[i][j][reg] = dW[m0 + 4 (crow(reg) + 4 fh) + i][n0 + 4 fr + j]  ->  one 16-byte store per (i, reg)
  const int m0 = (tmb + (w >> 1)) * 128, n0 = (tnb + (w & 1)) * 128;
  float* slab = L.slabs + (size_t)split * L.slab;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, L.M * L.ldc * 4, 0x00020000);
  const int n = n0 + 4 * fr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int m = m0 + 4 * (crow(rg) + 4 * fh) + i;       // rows >= M fall outside the descriptor and are dropped
      const uint32_t voff = n < L.ldc ? (uint32_t)((m * L.ldc + n) * 4) : 0x7FFFFFFFu;   // ldc % 4 == 0
      u32x4 v = {__float_as_uint(acc[i][0][rg]), __float_as_uint(acc[i][1][rg]), __float_as_uint(acc[i][2][rg]),
                 __float_as_uint(acc[i][3][rg])};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0);
    }
}

// `busy_wg` workgroups (= ceil(items / 4)) stream the dW tiles; the launch covers the whole chip, and the workgroups beyond
// them -- the items never fill it exactly (96 tiles x 10 splits = 960 of 1024 waves for the 8x512 net) -- work through
// the post-backward roles of kernels.hpp (head partials, x0 columns of dW, per-segment latent gradient) meanwhile, so
// those cost no launch of their own on the critical path.  post.rr_n + post.dw_n + post_lat_n == 0: nothing to do.
template <bool SPLIT>
__device__ __forceinline__ void dw_stream_body(const DwArgs& p, const PostBwdArgs& post, const int post_lat_n, const int busy_wg,
                                               float* ring) {   // ring: the split kernel's LDS ring; both kernels: the roles' scratch
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: item, K range and the
  const int fr = lane & 31, fh = lane >> 5;                                                  // ring's bounds checks stay scalar
  const int lwg = xcd_remap(blockIdx.x, gridDim.x);
#ifdef DSDF_LAB
  // lab: [0] start, [1] end, [2] 1 = role workgroup, [3..5] time inside the three role kinds (wave 0 of a role workgroup)
  unsigned long long* const dbgw = p.dbg ? p.dbg + (size_t)(lwg * 4 + w) * 8 : nullptr;
  if (dbgw && lane == 0) { dbgw[0] = __builtin_amdgcn_s_memrealtime(); dbgw[2] = lwg >= busy_wg; }
#endif
  if (lwg >= busy_wg) {
    if constexpr (SPLIT) return;      // (the roles do not ride in gemm_split mode: dsdf_api.hip run_backward_fused)
    const int total = post.rr_n + post.dw_n + post_lat_n;
#ifdef DSDF_LAB
    unsigned long long tk[3] = {0, 0, 0};
#endif
    for (int i = lwg - busy_wg; i < total; i += (int)gridDim.x - busy_wg) {
      __syncthreads();   // the roles reuse their static LDS arrays
#ifdef DSDF_LAB
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
      post_bwd_role_ride(post, i, post_lat_n, ring);
#ifdef DSDF_LAB
      tk[i < post.rr_n ? 0 : (i < post.rr_n + post.dw_n ? 1 : 2)] += __builtin_amdgcn_s_memrealtime() - t0;
#endif
    }
#ifdef DSDF_LAB
    if (dbgw && lane == 0) { dbgw[1] = __builtin_amdgcn_s_memrealtime(); dbgw[3] = tk[0]; dbgw[4] = tk[1]; dbgw[5] = tk[2]; }
#endif
    return;
  }
  const int nwaves = busy_wg * 4;
  const int wave = lwg * 4 + w;
  for (int item = wave; item - w < p.n_full; item += nwaves) {     // (item - w: the workgroup's first item -- uniform trip count)
    if constexpr (SPLIT) {
      // the workgroup's 4 items as ONE 2 x 2 block of tiles of one split: then the operand panels go through the LDS ring
      const int item0 = item - w;
      int l0 = 0;
      while (l0 + 1 < p.n_layers && item0 >= p.ly[l0 + 1].full0) ++l0;
      const DwLayer& L0 = p.ly[l0];
      const int tf0 = L0.tiles_m * L0.nfull_n, local0 = item0 - L0.full0;
      const bool block = item0 + 3 < p.n_full && (l0 + 1 >= p.n_layers || item0 + 3 < p.ly[l0 + 1].full0) && !(L0.tiles_m & 1) &&
                         !(L0.nfull_n & 1) && !(local0 & 3) && !(tf0 & 3);
      if (block) {
        const int split = local0 / tf0, b = (local0 - split * tf0) >> 2, nbn = L0.nfull_n >> 1;
        const int kbeg = split * L0.kchunk;
#if DW_SPLIT_WEAVE
        dw_block_split(L0, split, 2 * (b / nbn), 2 * (b % nbn), kbeg, min(p.N, kbeg + L0.kchunk), w, lane, ring);
#else
        dw_block_split_v1(L0, split, 2 * (b / nbn), 2 * (b % nbn), kbeg, min(p.N, kbeg + L0.kchunk), w, lane, ring);
#endif
        continue;
      }
    }
    if (item >= p.n_full) continue;
    int l = 0;
    while (l + 1 < p.n_layers && item >= p.ly[l + 1].full0) ++l;
    const DwLayer& L = p.ly[l];
    const int tf = L.tiles_m * L.nfull_n;
    const int local = item - L.full0;
    const int split = local / tf, tile = local - split * tf;
    const int kbeg = split * L.kchunk;
    // the 4 waves of a workgroup take 4 consecutive tiles of one split: as a 2 x 2 block they share each operand panel
    // pairwise through the CU's L1 (4 KB per k-step instead of 5 KB for a 1 x 4 strip)
    int tm = tile / L.nfull_n, tn = tile - tm * L.nfull_n;
    if (!(L.tiles_m & 1) && !(L.nfull_n & 1)) {
      const int nbn = L.nfull_n >> 1, b = tile >> 2, i = tile & 3;
      tm = 2 * (b / nbn) + (i >> 1);
      tn = 2 * (b % nbn) + (i & 1);
    }
    dw_item<4>(L, split, tm * 128, tn * 128, kbeg, min(p.N, kbeg + L.kchunk), fr, fh);
  }
  if (p.n_narrow > 0) {
    const int used = p.n_full % nwaves;                 // waves busy in the last round of full items
    const int spare = used == 0 ? nwaves : nwaves - used;
    const int first = used == 0 ? 0 : used;
    if (wave >= first) {
      for (int item = wave - first; item < p.n_narrow; item += spare) {
        int l = 0;
        while (l + 1 < p.n_layers && item >= p.ly[l + 1].narrow0) ++l;
        const DwLayer& L = p.ly[l];
        const int local = item - L.narrow0;
        const int split = local / L.tiles_m, tm = local - split * L.tiles_m;
        const int kbeg = split * L.kchunk, kend = min(p.N, kbeg + L.kchunk);
        const int m0 = tm * 128, n0 = (L.tiles_n - 1) * 128;
        switch (L.last_nj) {
          case 1: dw_item<1>(L, split, m0, n0, kbeg, kend, fr, fh); break;
          case 2: dw_item<2>(L, split, m0, n0, kbeg, kend, fr, fh); break;
          default: dw_item<3>(L, split, m0, n0, kbeg, kend, fr, fh); break;
        }
      }
    }
  }
#ifdef DSDF_LAB
  if (dbgw && lane == 0) dbgw[1] = __builtin_amdgcn_s_memrealtime();
#endif
}

__global__ __launch_bounds__(256, 1) void dw_stream_kernel(const DwArgs p, const PostBwdArgs post, const int post_lat_n,
                                                           const int busy_wg) {
  __shared__ __attribute__((aligned(16))) float role_lds[ROLE_LDS_FLOATS];     // only the riding roles use LDS here (37 KB)
  dw_stream_body<false>(p, post, post_lat_n, busy_wg, role_lds);
}
// DsdfNet.gemm_split: blocks of full-width items on the bf16 pipe (dw_block_split); everything else (items that do not form a 2 x 2
// block, narrow edge items, the riding roles) as above.  A kernel of its own so that the fp32 kernel keeps its register allocation.
__global__ __launch_bounds__(256, 1) void dw_stream_split_kernel(const DwArgs p, const PostBwdArgs post, const int post_lat_n,
                                                                 const int busy_wg) {
  __shared__ __attribute__((aligned(16))) float ring[DWS_RING * DWS_SLOT];     // 128 KB (role workgroups use it as their scratch)
  static_assert(DWS_RING * DWS_SLOT >= ROLE_LDS_FLOATS, "the roles' scratch must fit the ring");
  dw_stream_body<true>(p, post, post_lat_n, busy_wg, ring);
}

}  // namespace dsdf
