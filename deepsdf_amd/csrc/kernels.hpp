// kernels.hpp -- the HBM-bound kernels around the GEMMs (gfx950): latent renorm/gather/concat, weight-norm
// materialisation, last-layer GEMV + tanh + clamped-L1 + its backward, split-K/weight-norm finalisation,
// segmented latent-gradient reduction, fused Adam; segment mode: per-scene hoisting (seg_hoist), the x0 columns of the weight
// gradients and the per-segment latent gradient from per-workgroup sums (post_bwd roles).  All reductions are deterministic.
#pragma once
#include "common.hpp"

namespace dsdf {

constexpr int FSEG_MAXW = 512;   // widest layer of the segment-sum latent-gradient path

// ---------------------------------------------------------------------------------------------------
// K0a: max-norm renorm of every looked-up latent row, in place (torch embedding_renorm_,
// train_deep_sdf.py:385,509).  One wave per segment; a segment whose scene already appears in an earlier
// segment is skipped, so every distinct scene is scaled exactly once.
// The same launch zeroes the dense latent-gradient table (nzero floats, grid-stride; nullptr: leave it).
__global__ void latent_renorm_kernel(float* __restrict__ table, int L, const int64_t* __restrict__ seg_scene,
                                     int R, float max_norm, float* __restrict__ dlat, long long nzero) {
  if (dlat != nullptr)
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < nzero; i += (long long)gridDim.x * blockDim.x * 4) {
      if (i + 3 < nzero) *reinterpret_cast<float4*>(dlat + i) = make_float4(0.f, 0.f, 0.f, 0.f);   // dlat is 16-byte aligned
      else for (long long q = i; q < nzero; ++q) dlat[q] = 0.f;
    }
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R || max_norm <= 0.f) return;
  const int64_t j = seg_scene[r];
  int dup = 0;
  for (int q = lane; q < r; q += 64) dup |= (seg_scene[q] == j);
  if (__any(dup)) return;
  float* row = table + (size_t)j * L;
  float ss = 0.f;
  for (int c = lane; c < L; c += 64) { const float v = row[c]; ss += v * v; }
  ss = wave_sum(ss);
  const float nu = sqrtf(ss);
  if (nu > max_norm) {
    const float s = max_norm / (nu + 1e-7f);
    for (int c = lane; c < L; c += 64) row[c] *= s;
  }
}

// K0b: x0[n] = [E[scene(n)] || xyz[n]] written to up to 1 + popcount(skip_mask) destinations
// (train_deep_sdf.py:509-511; the skip destinations realise deep_sdf_decoder.py:88-89 without a cat).
// A destination takes columns [src0, src0 + ncols) of x0 at its column col0: the whole x0 for layer 0 and the skip layer, only
// the xyz columns for the layers of an xyz_in_all net (deep_sdf_decoder.py:90-91).  drop != 0 (layer 0 of a latent_dropout net
// in training, :79-82): the latent columns pass through the dropout hash (key drop_key, p = 0.2) on their way.
struct GatherDst { float* ptr; int ld; int col0; int src0; int ncols; int drop; };
struct GatherArgs {
  const float* table; int L; const float* xyz; int G;
  const int64_t* seg_scene; const int64_t* seg_offset; int R;
  const float* input; long long ld_in;  // module path: rows come from an explicit [n, L+G] input instead
  uint32_t drop_key, drop_thr; float drop_scale; uint32_t row_offset;
  int n; int ndst; GatherDst dst[DSDF_MAX_LAYERS + 1];
};
__global__ void gather_concat_kernel(const GatherArgs p) {
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= p.n) return;
  const int W = p.L + p.G;
  const float* src_lat;
  const float* src_xyz;
  if (p.input != nullptr) {
    src_lat = p.input + (size_t)n * p.ld_in;
    src_xyz = src_lat + p.L;
  } else {
    int lo = 0, hi = p.R;  // last segment with offset <= n
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.seg_offset[mid] <= n) lo = mid; else hi = mid; }
    src_lat = p.table + (size_t)p.seg_scene[lo] * p.L;
    src_xyz = p.xyz + (size_t)n * p.G;
  }
  const uint32_t g = p.row_offset + (uint32_t)n;
  for (int c = lane; c < W; c += 64) {
    const float v = c < p.L ? src_lat[c] : src_xyz[c - p.L];
    for (int d = 0; d < p.ndst; ++d) {
      const GatherDst& D = p.dst[d];
      if (c < D.src0 || c >= D.src0 + D.ncols) continue;
      float o = v;
      if (D.drop && c < p.L) o = drop_keep(drop_pair_hash(drop_col_key((uint32_t)c, p.drop_key), g), g, p.drop_thr) ? v * p.drop_scale : 0.f;
      D.ptr[(size_t)n * D.ld + D.col0 + (c - D.src0)] = o;
    }
  }
}

// latent_dropout backward: d/d(latent) that came through layer 0 passes the same mask (in place, columns < L of dz [n][ld])
__global__ void latent_drop_bwd_kernel(float* dz, int ld, int n, int L, uint32_t key, uint32_t thr, float scale, uint32_t row_offset) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * L) return;
  const int r = (int)(i / L), c = (int)(i % L);
  const uint32_t g = row_offset + (uint32_t)r;
  float* q = dz + (size_t)r * ld + c;
  *q = drop_keep(drop_pair_hash(drop_col_key((uint32_t)c, key), g), g, thr) ? *q * scale : 0.f;
}

// dst[r][c0 + j] (+)= src[r][j], j < w   (xyz_in_all: d/d(xyz) of one layer added to the running sum)
__global__ void acc_cols_kernel(const float* src, int lds, float* dst, int ldd, int c0, int n, int w, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * w) return;
  const int r = (int)(i / w), j = (int)(i % w);
  float* q = dst + (size_t)r * ldd + c0 + j;
  const float v = src[(size_t)r * lds + j];
  *q = accumulate ? *q + v : v;
}

// xyz_in_all, module path: the LAST layer's own d/d(xyz) = du * w_last[out_prev + j] (its input is [a || xyz]); du is rebuilt from
// the saved pre-activation u and the incoming gradient exactly as last_layer_kernel<LAST_BWD_EXT> does
__global__ void last_xyz_grad_kernel(const float* d_sdf, const float* u_in, const float* w_xyz, int G, int use_tanh, float* dst,
                                     int ldd, int c0, int n, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * G) return;
  const int r = (int)(i / G), j = (int)(i % G);
  const float u = u_in[r];
  const float t1 = use_tanh ? tanhf(u) : u;
  const float y = tanhf(t1);
  float du = d_sdf[r] * (1.f - y * y);
  if (use_tanh) du *= (1.f - t1 * t1);
  float* q = dst + (size_t)r * ldd + c0 + j;
  const float v = du * w_xyz[j];
  *q = accumulate ? *q + v : v;
}

// ---------------------------------------------------------------------------------------------------
// K0c (segment mode): the per-scene part of the layers that see x0 = [latent | xyz] (layer 0 and the skip layer),
//   U[s][t][n] = sum_c W_t[n][c0_t + c] * latent[scene_s][c]      (deep_sdf_decoder.py:86-92 restricted to the latent columns)
// computed once per segment instead of once per point; the fused forward starts its accumulators from it (fused.hpp).
// One wave per weight row (its latent columns stay in registers), looping over the segments.
constexpr int HOIST_MAXL = 512;
constexpr int HOIST_SC = 16;               // segments per wave: all their loads are in flight together
struct HoistArgs {
  int nh;                                  // hoisted layers (1 or 2)
  const float* W[2]; int ldw[2]; int c0[2]; int out[2];
  int L; const int64_t* seg_scene; const float* table; int R;
  float* U; int ldu;                       // [R][2][ldu]
  int bf16;                                // config 5: both operands rounded to bf16 (RNE), fp32 accumulation
  // training steps: the max-norm renorm of the looked-up rows (K0a, embedding_renorm_, train_deep_sdf.py:385,509) is folded in.
  // Every wave computes the scale of its segments' rows from the RAW table (same arithmetic in every wave: the lane-strided sum
  // of squares + wave_sum of latent_renorm_kernel) and contracts the scaled values; nobody writes the table here -- other blocks
  // are reading it.  The scaled rows go to zr [R][L] (written by the first block row, read by everything downstream of this
  // launch instead of table[seg_scene[r]]); seg_scatter_body writes them back into the table.
  float max_norm;                          // <= 0: no renorm (zr, if given, is a plain copy)
  float* zr;                               // or nullptr (inference: dsdf_decode_latent)
  float* dlat; long long nzero;            // the dense latent gradient, zeroed here (grid-stride), or nullptr
};
// KU = ceil(L / 64) as a compile-time constant: every load of the wave (the weight row's and the 16 latent rows' columns lane + 64 k)
// is unconditional -- lanes past L read column L - 1 and drop it -- and goes out before the first use: ONE wait.  (Loads under a
// per-lane or per-k condition each became a branch with a wait of its own: 36 us for this kernel instead of 10.)
template <int KU>
__device__ __forceinline__ void seg_hoist_body(const HoistArgs& p, const int n, const int t) {
  const int lane = threadIdx.x & 63;
  const int s0 = blockIdx.y * HOIST_SC;
  const long long myscene = (long long)p.seg_scene[min(s0 + (lane & (HOIST_SC - 1)), p.R - 1)];   // one load; broadcast below
  const float* wrow = p.W[t] + (size_t)n * p.ldw[t] + p.c0[t];
  float wv[KU], v[HOIST_SC][KU];
#pragma unroll
  for (int k = 0; k < KU; ++k) wv[k] = wrow[min(lane + 64 * k, p.L - 1)];
#pragma unroll
  for (int q = 0; q < HOIST_SC; ++q) {
    const float* row = p.table + (size_t)__shfl(myscene, q, 64) * p.L;
#pragma unroll
    for (int k = 0; k < KU; ++k) v[q][k] = row[min(lane + 64 * k, p.L - 1)];
  }
#pragma unroll
  for (int k = 0; k < KU; ++k) {
    const bool ok = lane + 64 * k < p.L;
    wv[k] = ok ? wv[k] : 0.f;
#pragma unroll
    for (int q = 0; q < HOIST_SC; ++q) v[q][k] = ok ? v[q][k] : 0.f;
  }
  static_assert(HOIST_SC == 16, "wave_sum16");
  if (p.max_norm > 0.f) {     // (the 16 + 16 reductions of this kernel as 16 x 6 shuffles each were two thirds of its time)
    float ss[HOIST_SC];
#pragma unroll
    for (int q = 0; q < HOIST_SC; ++q) {
      ss[q] = 0.f;
#pragma unroll
      for (int k = 0; k < KU; ++k) ss[q] += v[q][k] * v[q][k];    // (columns >= L hold 0)
    }
    const float mine = wave_sum16(ss, lane);
#pragma unroll
    for (int q = 0; q < HOIST_SC; ++q) {
      const float nu = sqrtf(__shfl(mine, wave_sum16_lane(q), 64));
      if (nu > p.max_norm) {
        const float sc = p.max_norm / (nu + 1e-7f);
#pragma unroll
        for (int k = 0; k < KU; ++k) v[q][k] *= sc;
      }
    }
  }
  if (p.zr != nullptr && blockIdx.x == 0 && threadIdx.x < 64) {
#pragma unroll
    for (int q = 0; q < HOIST_SC; ++q)
      if (s0 + q < p.R)
#pragma unroll
        for (int k = 0; k < KU; ++k)
          if (lane + 64 * k < p.L) p.zr[(size_t)(s0 + q) * p.L + lane + 64 * k] = v[q][k];
  }
  float a[HOIST_SC];
#pragma unroll
  for (int q = 0; q < HOIST_SC; ++q) a[q] = 0.f;
#pragma unroll
  for (int k = 0; k < KU; ++k) {
    const float w = p.bf16 ? (float)(__bf16)wv[k] : wv[k];
#pragma unroll
    for (int q = 0; q < HOIST_SC; ++q) {
      const float x = p.bf16 ? (float)(__bf16)v[q][k] : v[q][k];
      if (lane + 64 * k < p.L) a[q] = fmaf(w, x, a[q]);
    }
  }
  const float r = wave_sum16(a, lane);
  const int qi = wave_sum16_index(lane);
  if ((lane & 3) == 0 && s0 + qi < p.R) p.U[((size_t)(s0 + qi) * 2 + t) * p.ldu + n] = r;
}
template <int KU>    // one kernel per KU: the 8-unit body's 206 VGPRs would otherwise set the occupancy of every latent size
__global__ __launch_bounds__(256) void seg_hoist_kernel(const HoistArgs p) {   // grid (rows / 4, ceil(R / HOIST_SC)); needs L <= 64 KU
  if (p.dlat != nullptr) {
    const long long nb = (long long)gridDim.x * gridDim.y, b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    for (long long i = (b * 256 + threadIdx.x) * 4; i < p.nzero; i += nb * 256 * 4) {
      if (i + 3 < p.nzero) *reinterpret_cast<float4*>(p.dlat + i) = make_float4(0.f, 0.f, 0.f, 0.f);   // dlat is 16-byte aligned
      else for (long long q = i; q < p.nzero; ++q) p.dlat[q] = 0.f;
    }
  }
  int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  int t = 0;
  if (n >= p.out[0]) { n -= p.out[0]; t = 1; }
  if (t >= p.nh || n >= p.out[t]) return;
  seg_hoist_body<KU>(p, n, t);
}

// K5c (segment mode): the x0 columns of the hoisted layers' weight gradients, from per-workgroup sums of the fused
// backward instead of a GEMM over the points (x0 = [latent_s | xyz] is constant / 3 wide):
//   HS[t][i][c]     = sum_s (sum of segment s's workgroup column sums of dP_t[:, i]) * latent[scene_s][c]     c < L
//   HS[t][i][L + j] = sum_wg xsum_t[wg][j][i]                                                                 j < G
// finalize_row adds HS to the row's columns [lat0, lat0 + L + G).  Block = 8 output rows x all columns; fixed order.
// Round 4: 32 output rows per block instead of 8 (a quarter of the blocks, each with the same number of memory round trips) and every
// run of dependent loads batched -- as riding roles on the dW launch's 16 spare workgroups these blocks were 115 us of a 391 us role
// chain that outlasted the launch's MFMA items (377 us: profiles/r04_dw_stamps_before.log).
// ... which is the RIDING form; a launch of its own has the whole chip and keeps 8 rows per block (four times the blocks).
constexpr int SDW_ROWS_RIDE = 32, SDW_ROWS_WIDE = 8;
constexpr int SDW_LDS_FLOATS = 64 * SDW_ROWS_RIDE + 256 * 4 + 2 * 64;     // css [64][rows] + xred [256 / rows][rows][4] + srow [64]
struct SegDwArgs {
  int nh; const float* cs[2]; const float* xsum[2]; int out[2];   // per-workgroup sums [nwg][ldcs] / [nwg][4][ldcs]
  int ldcs, nwg, wg_per_seg, R, L, G;
  const int64_t* seg_scene; const float* table;
  const float* zr;                                                // non-null: segment r's latent row is zr + r L (seg_hoist_kernel)
  float* HS; int ldh; long long hstride;                          // HS[t] = HS + t * hstride, [out_t][ldh]
};
template <int SDW_ROWS>
__device__ __forceinline__ void seg_dw_body(const SegDwArgs& p, int bidx, float* lds) {
  constexpr int SDW_SLOTS = 256 / SDW_ROWS;   // loader mapping: SDW_ROWS consecutive rows x SDW_SLOTS segments / workgroup slices
  float (*css)[SDW_ROWS] = reinterpret_cast<float (*)[SDW_ROWS]>(lds);                               // [64 segments][rows]
  float (*xred)[SDW_ROWS][4] = reinterpret_cast<float (*)[SDW_ROWS][4]>(lds + 64 * SDW_ROWS);          // [slots][rows][4]
  long long* srow = reinterpret_cast<long long*>(lds + 64 * SDW_ROWS + SDW_SLOTS * SDW_ROWS * 4);      // [64]
  const int tid = threadIdx.x;
  int blk = bidx, t = 0;
  const int b0 = (p.out[0] + SDW_ROWS - 1) / SDW_ROWS;
  if (blk >= b0) { blk -= b0; t = 1; }
  const int i0 = blk * SDW_ROWS;
  const float* cs = p.cs[t];
  const int r_ld = tid & (SDW_ROWS - 1), s_ld = tid / SDW_ROWS;
  const bool rok = i0 + r_ld < p.out[t];
  const int ird = min(i0 + r_ld, p.out[t] - 1);    // (loads are unconditional on clamped, valid addresses; the values are dropped by selects)
  // xyz columns first (independent loads, in flight under everything else): SDW_SLOTS slices of the workgroups, 8 loads per batch
  float xa[4] = {0.f, 0.f, 0.f, 0.f};
  for (int wg0 = s_ld; wg0 < p.nwg; wg0 += 8 * SDW_SLOTS) {
    float tx[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int wg = wg0 + u * SDW_SLOTS;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        tx[u][j] = p.xsum[t][((size_t)min(wg, p.nwg - 1) * 4 + j) * p.ldcs + ird];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) xa[j] += (rok && wg0 + u * SDW_SLOTS < p.nwg && j < p.G) ? tx[u][j] : 0.f;
  }
  constexpr int NC = HOIST_MAXL / 256;       // column passes of 256
  float acc[NC][SDW_ROWS];
#pragma unroll
  for (int k = 0; k < NC; ++k)
#pragma unroll
    for (int r = 0; r < SDW_ROWS; ++r) acc[k][r] = 0.f;
  for (int s0 = 0; s0 < p.R; s0 += 64) {
    __syncthreads();
    if (p.wg_per_seg <= 16) {      // many short segments: one thread per (row, segment), its workgroups in order (all loads in flight)
#pragma unroll
      for (int q = 0; q < 64 / SDW_SLOTS; ++q) {
        const int sl = s_ld + SDW_SLOTS * q, sg = s0 + sl;
        float tg[16];
#pragma unroll
        for (int g = 0; g < 16; ++g)
          tg[g] = cs[(size_t)(min(sg, p.R - 1) * p.wg_per_seg + min(g, p.wg_per_seg - 1)) * p.ldcs + ird];
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) a += (sg < p.R && rok && g < p.wg_per_seg) ? tg[g] : 0.f;
        css[sl][r_ld] = a;
      }
    } else {                       // longer segments (up to 16384 samples per scene): SDW_SLOTS threads share a segment's workgroups
      for (int sl = 0; sl < 64; ++sl) {
        const int sg = s0 + sl;
        float a = 0.f;
        if (sg < p.R && rok)
          for (int g0 = s_ld; g0 < p.wg_per_seg; g0 += 8 * SDW_SLOTS) {
            float tg[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int g = g0 + u * SDW_SLOTS;
              tg[u] = g < p.wg_per_seg ? cs[(size_t)(sg * p.wg_per_seg + g) * p.ldcs + i0 + r_ld] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += tg[u];
          }
        xred[s_ld][r_ld][0] = a;   // (xred is free here: its xyz partials are written after this loop)
        __syncthreads();
        if (tid < SDW_ROWS) {
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < SDW_SLOTS; ++q) v += xred[q][tid][0];
          css[sl][tid] = v;
        }
        __syncthreads();
        if (sg + 1 >= p.R) {       // the remaining slots of this pass are empty
          for (int z = sl + 1 + s_ld; z < 64; z += SDW_SLOTS) css[z][r_ld] = 0.f;
          break;
        }
      }
    }
    if (tid < 64) {
      const int sg = min(s0 + tid, p.R - 1);                       // segments beyond R: a valid row, css == 0
      srow[tid] = p.zr != nullptr ? (long long)sg * p.L : (long long)p.seg_scene[sg] * p.L;
    }
    __syncthreads();
    const float* lat = p.zr != nullptr ? p.zr : p.table;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int c = tid + 256 * k;
      if (c < p.L) {
        float v[64];                          // every latent value of this column for the 64 segments: one wait
#pragma unroll
        for (int u = 0; u < 64; ++u) v[u] = lat[srow[u] + c];
#pragma unroll
        for (int u = 0; u < 64; ++u) {
          const float4* cq = reinterpret_cast<const float4*>(css[u]);   // broadcast 16-byte reads
#pragma unroll
          for (int r4 = 0; r4 < SDW_ROWS / 4; ++r4) {
            const float4 cv = cq[r4];
            acc[k][4 * r4 + 0] = fmaf(cv.x, v[u], acc[k][4 * r4 + 0]);
            acc[k][4 * r4 + 1] = fmaf(cv.y, v[u], acc[k][4 * r4 + 1]);
            acc[k][4 * r4 + 2] = fmaf(cv.z, v[u], acc[k][4 * r4 + 2]);
            acc[k][4 * r4 + 3] = fmaf(cv.w, v[u], acc[k][4 * r4 + 3]);
          }
        }
      }
    }
  }
  float* hs = p.HS + (size_t)t * p.hstride;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int c = tid + 256 * k;
    if (c < p.L)
#pragma unroll
      for (int r = 0; r < SDW_ROWS; ++r)
        if (i0 + r < p.out[t]) hs[(size_t)(i0 + r) * p.ldh + c] = acc[k][r];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) xred[s_ld][r_ld][j] = xa[j];
  __syncthreads();
  if (tid < SDW_ROWS * 4) {                  // fixed-order sum of the slices
    const int r = tid >> 2, j = tid & 3;
    if (j < p.G && i0 + r < p.out[t]) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < SDW_SLOTS; ++q) v += xred[q][r][j];
      hs[(size_t)(i0 + r) * p.ldh + p.L + j] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K1: W = g * v / ||v||_row (torch._weight_norm(v, g, 0)) or plain copy, for ALL layers in two launches:
//   wn_scale_kernel : one wave per (layer,row): scale = g / ||v_row||  (1 for plain layers)
//   wn_tiles_kernel : one block per 32x32 tile: W tile (coalesced) and W^T tile through a 32x33 LDS transpose.
struct WnLayer {
  const float* v; const float* g; float* W; float* WT; int out, in, ldw, ldwt; int row0; int tile0; int tcols;
  float* Wf; float* WTf;   // fragment-ordered copies for the fused kernels (see fused.hpp), or nullptr
  __bf16* Wfb;             // bf16 fragment-ordered copy for the bf16 forward (fused_forward_bf16_kernel), or nullptr
  int Uf, UTf;             // k-units (of 16) allocated per n-tile in Wf / WTf
  __bf16* Ws; __bf16* WTs; // DsdfNet.gemm_split: W and W^T cut into three bf16 planes (h, m, l: fused.hpp fused_kloop_split) in the
  long long ws_plane, wts_plane;   // fragment order of Wfb with NATURAL k-unit order; elements per plane.  Or nullptr.
};
// (an optional dense Adam update -- the latent table -- rides on the same launch: blocks >= total_tiles, kernels.hpp adam_kernel math)
struct AdamRide {
  float* p; const float* g; float* m; float* v; long long n; int blocks;   // blocks == 0: none
  float omb1, b2, omb2, step_size, bc2_sqrt, eps;
};
struct WnAll { int nl; int total_rows; int total_tiles; float* scale; AdamRide adam; WnLayer ly[DSDF_MAX_LAYERS]; };

__global__ __launch_bounds__(256) void wn_scale_kernel(const WnAll p) {
  const int gr = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (gr >= p.total_rows) return;
  int l = 0;
  while (l + 1 < p.nl && gr >= p.ly[l + 1].row0) ++l;
  const WnLayer& L = p.ly[l];
  const int row = gr - L.row0;
  float s = 1.f;
  if (L.g != nullptr) {
    const float* vr = L.v + (size_t)row * L.in;
    float ss = 0.f;
    for (int c = lane; c < L.in; c += 64) { const float x = vr[c]; ss += x * x; }
    ss = wave_sum(ss);
    s = L.g[row] / sqrtf(ss);
  }
  if (lane == 0) p.scale[gr] = s;
}

// Fragment order (fused.hpp): for an operand matrix B[n][k] (n = output column of the product, k = contraction):
//   Bf[((nt * U + u) * 2 + i) * 256 + lane * 4 + e] = B[32 nt + (lane & 31)][16 u + 8 (lane >> 5) + 4 i + e]
// so that a wave's MFMA B-operand for k-unit u of n-tile nt is two perfectly coalesced 1-KiB dwordx4 loads.
__global__ __launch_bounds__(256) void wn_tiles_kernel(const WnAll p) {
  __shared__ float tile[32][33];
  const int t = blockIdx.x;
  if (t >= p.total_tiles) {   // Adam on a dense table (torch single-tensor math, same expressions as adam_kernel)
    const AdamRide& a = p.adam;
    for (long long i = (long long)(t - p.total_tiles) * 256 + threadIdx.x; i < a.n; i += (long long)a.blocks * 256) {
      const float gi = a.g[i];
      const float mi = fmaf(a.omb1, gi - a.m[i], a.m[i]);
      const float vi = a.b2 * a.v[i] + a.omb2 * gi * gi;
      a.m[i] = mi; a.v[i] = vi;
      a.p[i] = a.p[i] - a.step_size * (mi / (sqrtf(vi) / a.bc2_sqrt + a.eps));
    }
    return;
  }
  int l = 0;
  while (l + 1 < p.nl && t >= p.ly[l + 1].tile0) ++l;
  const WnLayer& L = p.ly[l];
  const int tt = t - L.tile0;
  const int rb = tt / L.tcols, cb = tt % L.tcols;
  const int r0 = rb * 32, c0 = cb * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rr = ty + 8 * k, row = r0 + rr, col = c0 + tx;
    float w = 0.f;
    if (row < L.out && col < L.in) {
      w = L.v[(size_t)row * L.in + col] * p.scale[L.row0 + row];
      L.W[(size_t)row * L.ldw + col] = w;
    }
    tile[rr][tx] = w;
  }
  if (L.WT == nullptr && L.Wf == nullptr && L.Wfb == nullptr && L.Ws == nullptr) return;
  __syncthreads();
  if (L.WT != nullptr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = ty + 8 * k, col = c0 + cc, row = r0 + tx;
      if (col < L.in && row < L.out) L.WT[(size_t)col * L.ldwt + row] = tile[tx][cc];
    }
  }
  const int uu = threadIdx.x >> 7, i = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
  const int fr = lane & 31, fh = lane >> 5, kk = 16 * uu + 8 * fh + 4 * i;
  if (L.Wf != nullptr) {   // B = W: n = out index (tile rows), k = in index (tile cols)
    float4 v4 = make_float4(tile[fr][kk], tile[fr][kk + 1], tile[fr][kk + 2], tile[fr][kk + 3]);
    *reinterpret_cast<float4*>(L.Wf + ((size_t)(rb * L.Uf + 2 * cb + uu) * 2 + i) * 256 + lane * 4) = v4;
  }
  if (L.Wfb != nullptr) {  // bf16(W), lane (r, h) holds k = 16u + 8h + j (j = 0..7) contiguously: this thread's j = 4i .. 4i+3.
    // k-units are stored PHASE-MAJOR, 32 slots per n-tile (fused_bf16x8.hpp): the units of the even 32-column input tiles in slots
    // 0.., those of the odd tiles in slots 16.. -- the order the 8-wave forward contracts them in (slot = bf8_slot(u))
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 h4 = {(__bf16)tile[fr][kk], (__bf16)tile[fr][kk + 1], (__bf16)tile[fr][kk + 2], (__bf16)tile[fr][kk + 3]};
    const int slot = (cb & 1) * 16 + 2 * (cb >> 1) + uu;      // unit u = 2 cb + uu
    *reinterpret_cast<bf16x4*>(L.Wfb + ((size_t)(rb * 32 + slot) * 64 + lane) * 8 + 4 * i) = h4;
  }
  if (L.WTf != nullptr) {  // B = W^T: n = in index (tile cols), k = out index (tile rows)
    float4 v4 = make_float4(tile[kk][fr], tile[kk + 1][fr], tile[kk + 2][fr], tile[kk + 3][fr]);
    *reinterpret_cast<float4*>(L.WTf + ((size_t)(cb * L.UTf + 2 * rb + uu) * 2 + i) * 256 + lane * 4) = v4;
  }
  if (L.Ws != nullptr) {   // x = h + m + l, every term the top 16 bits of what is left (exact: see fused_kloop_split)
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    auto cut = [](const float (&x)[4], bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float hf = __uint_as_float(__float_as_uint(x[e]) & 0xFFFF0000u), r = x[e] - hf;
        const float mf = __uint_as_float(__float_as_uint(r) & 0xFFFF0000u), t = r - mf;
        h[e] = (__bf16)hf; m[e] = (__bf16)mf; l[e] = (__bf16)__uint_as_float(__float_as_uint(t) & 0xFFFF0000u);
      }
    };
    bf16x4 h, m, l;
    const float xw[4] = {tile[fr][kk], tile[fr][kk + 1], tile[fr][kk + 2], tile[fr][kk + 3]};
    cut(xw, h, m, l);
    __bf16* q = L.Ws + ((size_t)(rb * L.Uf + 2 * cb + uu) * 64 + lane) * 8 + 4 * i;
    *reinterpret_cast<bf16x4*>(q) = h;
    *reinterpret_cast<bf16x4*>(q + L.ws_plane) = m;
    *reinterpret_cast<bf16x4*>(q + 2 * L.ws_plane) = l;
    const float xt[4] = {tile[kk][fr], tile[kk + 1][fr], tile[kk + 2][fr], tile[kk + 3][fr]};
    cut(xt, h, m, l);
    q = L.WTs + ((size_t)(cb * L.UTf + 2 * rb + uu) * 64 + lane) * 8 + 4 * i;
    *reinterpret_cast<bf16x4*>(q) = h;
    *reinterpret_cast<bf16x4*>(q + L.wts_plane) = m;
    *reinterpret_cast<bf16x4*>(q + 2 * L.wts_plane) = l;
  }
}

// ---------------------------------------------------------------------------------------------------
// K3: last layer.  One wave per point row (4 rows per block iteration).
//   u = <a, w> + b ; t1 = use_tanh ? tanh(u) : u ; y = tanh(t1)               (deep_sdf_decoder.py:92-95,108-109)
//   train: yh = clamp(y), th = clamp(gt); loss += |yh - th| / n_norm            (train_deep_sdf.py:493,517-521)
//          dy = sign(yh - th) * [|y| <= delta] / n_norm ; du = dy (1-y^2) (1-t1^2 if use_tanh)
//          dW_last partial += du * a ; db_last partial += du ; dp_prev = du * w * [a > 0] * mask_scale
//          colsum_prev partial += dp_prev (bias gradient of the previous layer)
//   module backward: dy comes from d_sdf instead of the loss.
enum { LAST_FWD = 0, LAST_TRAIN = 1, LAST_BWD_EXT = 2 };
struct LastArgs {
  const float* a; int lda; int in; const float* w; const float* b; int n;
  int use_tanh;
  float* y_out;            // [n] (may be null)
  float* u_save;           // [n] saved pre-tanh output (module path), may be null
  const float* gt; float delta; float inv_n;
  const float* d_sdf;      // LAST_BWD_EXT
  const float* u_in;       // LAST_BWD_EXT: saved u
  float* dp_prev; int lddp; float mask_scale;
  float* part_dw;          // [nblk][ld_part]
  int ld_part;
  float* part_db;          // [nblk]
  float* part_loss;        // [nblk]
  float* part_colsum;      // [nblk][ld_part]
  // latent_in names the OUTPUT layer (deep_sdf_decoder.py:88-89 with layer = num_layers - 2): its input is [a | x0]; columns >= n_act are
  // x0, their gradient du w[c] passes no ReLU / dropout mask and goes to dz [n][ldz] (first dz_cols of them), not to dp_prev.
  int n_act;               // columns of the input that are the previous layer's activations (= in without such a skip)
  float* dz; int ldz; int dz_cols;
};
template <int MODE, int NCH>  // NCH float4 chunks per lane: in <= 256 * NCH
__global__ __launch_bounds__(256) void last_layer_kernel(const LastArgs p) {
  __shared__ float red[4][4];
  __shared__ float redv[4][256 * NCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4 w4[NCH];
  bool cok[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = 4 * lane + 256 * c;
    cok[c] = col < p.in;  // in is a multiple of 4 for every supported width (checked on the host)
    w4[c] = cok[c] ? *reinterpret_cast<const float4*>(p.w + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float bias = p.b[0];
  float4 dw[NCH], cs[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { dw[c] = make_float4(0.f, 0.f, 0.f, 0.f); cs[c] = dw[c]; }
  float loss = 0.f, db = 0.f;
  const int stride = gridDim.x * 4;
  for (int row = blockIdx.x * 4 + wave; row < p.n; row += stride) {
    float4 a4[NCH];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      a4[c] = cok[c] ? *reinterpret_cast<const float4*>(p.a + (size_t)row * p.lda + 4 * lane + 256 * c)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
      dot += a4[c].x * w4[c].x + a4[c].y * w4[c].y + a4[c].z * w4[c].z + a4[c].w * w4[c].w;
    }
    float u;
    if constexpr (MODE == LAST_BWD_EXT) u = p.u_in[row];
    else u = wave_sum(dot) + bias;
    const float t1 = p.use_tanh ? tanhf(u) : u;
    const float y = tanhf(t1);
    if constexpr (MODE != LAST_BWD_EXT) {
      if (lane == 0) {
        if (p.y_out) p.y_out[row] = y;
        if (p.u_save) p.u_save[row] = u;
      }
    }
    if constexpr (MODE != LAST_FWD) {
      float dy;
      if constexpr (MODE == LAST_TRAIN) {
        const float yh = fminf(fmaxf(y, -p.delta), p.delta);
        const float th = fminf(fmaxf(p.gt[row], -p.delta), p.delta);
        const float diff = yh - th;
        loss += fabsf(diff);
        const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        dy = (y >= -p.delta && y <= p.delta) ? sg * p.inv_n : 0.f;
      } else {
        dy = p.d_sdf[row];
      }
      float du = dy * (1.f - y * y);
      if (p.use_tanh) du *= (1.f - t1 * t1);
      db += du;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (cok[c]) {
          dw[c].x += du * a4[c].x; dw[c].y += du * a4[c].y; dw[c].z += du * a4[c].z; dw[c].w += du * a4[c].w;
          float4 d;
          d.x = a4[c].x > 0.f ? du * w4[c].x * p.mask_scale : 0.f;
          d.y = a4[c].y > 0.f ? du * w4[c].y * p.mask_scale : 0.f;
          d.z = a4[c].z > 0.f ? du * w4[c].z * p.mask_scale : 0.f;
          d.w = a4[c].w > 0.f ? du * w4[c].w * p.mask_scale : 0.f;
          const int col0 = 4 * lane + 256 * c;
          if (col0 + 3 >= p.n_act) {       // (part of) the chunk lies in the x0 columns of an output-layer skip
            float dv[4] = {d.x, d.y, d.z, d.w};
            const float wv[4] = {w4[c].x, w4[c].y, w4[c].z, w4[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int cx = col0 + e - p.n_act;
              if (cx >= 0) {
                if (p.dz != nullptr && cx < p.dz_cols) p.dz[(size_t)row * p.ldz + cx] = du * wv[e];
                dv[e] = 0.f;
              }
            }
            d = make_float4(dv[0], dv[1], dv[2], dv[3]);
          }
          if (p.dp_prev) *reinterpret_cast<float4*>(p.dp_prev + (size_t)row * p.lddp + 4 * lane + 256 * c) = d;
          cs[c].x += d.x; cs[c].y += d.y; cs[c].z += d.z; cs[c].w += d.w;
        }
      }
    }
  }
  if constexpr (MODE != LAST_FWD) {
    // per-block partials, fixed order (wave 0..3)
    if (lane == 0) { red[wave][0] = loss; red[wave][1] = db; }
#pragma unroll
    for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(&redv[wave][4 * lane + 256 * c]) = dw[c];
    __syncthreads();
    if (tid == 0) {
      p.part_loss[blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
      p.part_db[blockIdx.x] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
    for (int c = tid; c < p.in; c += 256)
      p.part_dw[(size_t)blockIdx.x * p.ld_part + c] = (redv[0][c] + redv[1][c]) + (redv[2][c] + redv[3][c]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(&redv[wave][4 * lane + 256 * c]) = cs[c];
    __syncthreads();
    if (p.part_colsum)
      for (int c = tid; c < p.in; c += 256)
        p.part_colsum[(size_t)blockIdx.x * p.ld_part + c] = (redv[0][c] + redv[1][c]) + (redv[2][c] + redv[3][c]);
  }
}

// ---------------------------------------------------------------------------------------------------
// F1: per layer, one block per output row i: dW[i,:] = sum over split slabs; weight-norm backward
// (Appendix A.3: dg = <dW,v>/||v||, dv = g/||v|| dW - g dg/||v||^2 v); db[i] = sum of column partials.
struct FinArgs {
  const float* slabs; int nsplit; long long slab; int ldc;
  const float* colsum; int npart; int ldcs;
  const float* g; const float* v;   // parameters (g null for plain layers)
  float* dg; float* dv; float* db;  // gradient arena slices
  int out, in; int accumulate;
  // fused optimiser (single-GPU fast path, dsdf_train_step): when adam != 0 the gradients are consumed on the spot --
  // Adam on this row's bias, g and v (torch math, same expressions as adam_kernel), then the new weight-norm scale
  // g/||v|| of the row -- and are NOT written to the gradient arena.
  int adam;
  float* pb; float* pg; float* pv;          // parameters (mutable aliases of bias / g / v-or-weight)
  float* mb; float* mg; float* mv;          // exp_avg
  float* sb; float* sg; float* sv;          // exp_avg_sq
  float* scale_out;                         // [out] weight-norm scale of the updated row (1 for plain layers)
  float omb1, b2, omb2, step_size, bc2_sqrt, eps;
  // segment mode, hoisted layers (fused.hpp): columns [lat0, lat0 + hW) of the row multiply x0 = [latent_s | xyz]; their
  // gradient comes from seg_dw_kernel's hs[out][ldh] instead of the split-K slabs (columns < lat0 still do)
  int hoist, lat0, hW, ldh; const float* hs;
};

__device__ __forceinline__ float adam_elem(float& p, float g, float& m, float& v, const FinArgs& a) {
  const float mi = fmaf(a.omb1, g - m, m);
  const float vi = a.b2 * v + a.omb2 * g * g;
  m = mi; v = vi;
  p = p - a.step_size * (mi / (sqrtf(vi) / a.bc2_sqrt + a.eps));
  return p;
}
__device__ __forceinline__ void finalize_row(const FinArgs& p, const int i, float* red) {
  const int tid = threadIdx.x;
  float dot = 0.f, ss = 0.f;
  constexpr int MAXC = 8;  // in <= 2048
  float dwr[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = tid + 256 * k;
    float s = 0.f;
    if (c < p.in && (!p.hoist || c < p.lat0)) {
      const float* q = p.slabs + (size_t)i * p.ldc + c;
      int sp = 0;
      for (; sp + 8 <= p.nsplit; sp += 8) {  // 8 independent loads in flight, summed in fixed order
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = q[(size_t)(sp + u) * p.slab];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
      }
      for (; sp < p.nsplit; ++sp) s += q[(size_t)sp * p.slab];
    }
    dwr[k] = s;
  }
  if (p.hoist) {   // x0 columns of a hoisted layer: computed by seg_dw_kernel
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = tid + 256 * k - p.lat0;
      if (c >= 0 && c < p.hW) dwr[k] += p.hs[(size_t)i * p.ldh + c];
    }
  }
  if (p.g) {
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = tid + 256 * k;
      if (c < p.in) { const float vv = p.v[(size_t)i * p.in + c]; dot += dwr[k] * vv; ss += vv * vv; }
    }
  }
  if (p.g) {
    dot = block_sum_256(dot, red);
    ss = block_sum_256(ss, red);
    const float nrm = sqrtf(ss);
    const float gi = p.g[i];
    const float dgi = dot / nrm;
    const float a = gi / nrm, b = gi * dgi / (nrm * nrm);
    float ssn = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = tid + 256 * k;
      if (c < p.in) {
        const size_t o = (size_t)i * p.in + c;
        const float d = a * dwr[k] - b * p.v[o];
        if (p.adam) { const float vn = adam_elem(p.pv[o], d, p.mv[o], p.sv[o], p); ssn += vn * vn; }
        else p.dv[o] = p.accumulate ? p.dv[o] + d : d;
      }
    }
    if (p.adam) {
      ssn = block_sum_256(ssn, red);
      if (tid == 0) {
        const float gn = adam_elem(p.pg[i], dgi, p.mg[i], p.sg[i], p);
        p.scale_out[i] = gn / sqrtf(ssn);
      }
    } else if (tid == 0) p.dg[i] = p.accumulate ? p.dg[i] + dgi : dgi;
  } else {
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = tid + 256 * k;
      if (c < p.in) {
        const size_t o = (size_t)i * p.in + c;
        if (p.adam) adam_elem(p.pv[o], dwr[k], p.mv[o], p.sv[o], p);
        else p.dv[o] = p.accumulate ? p.dv[o] + dwr[k] : dwr[k];
      }
    }
    if (p.adam && tid == 0) p.scale_out[i] = 1.f;
  }
  // bias gradient: fixed-order sum of the column partials
  float s = 0.f;
  for (int q = tid; q < p.npart; q += 256) s += p.colsum[(size_t)q * p.ldcs + i];
  s = block_sum_256(s, red);
  if (tid == 0) {
    if (p.adam) adam_elem(p.pb[i], s, p.mb[i], p.sb[i], p);
    else p.db[i] = p.accumulate ? p.db[i] + s : s;
  }
}

__global__ __launch_bounds__(256) void finalize_layer_kernel(const FinArgs p) {
  __shared__ float red[4];
  finalize_row(p, blockIdx.x, red);
}

// The same row by ONE WAVE (rows of at most 64 * FW_MAXK = 512 inputs: every fused-path layer), four rows per block.  The block form
// above walks a chain of dependent memory round trips with three block barriers in it (slab sums -> v -> Adam state -> stores); here a
// lane requests EVERYTHING its columns need before the first use -- the split-K partials, v, both Adam moments, the bias partials -- and
// the three reductions are wave shuffles.  Same per-element slab order (sp = 0 .. nsplit - 1), its own (fixed) order for the row
// reductions.  Round 4: the finalize launch 34 -> see profiles/r04_bench_kernel_stats.csv.
constexpr int FW_MAXK = 8;
template <int FW_KU>      // columns per lane = ceil(in / 64) rounded up to 1, 2, 4 or 8: a compile-time count keeps every load unconditional
__device__ __forceinline__ void finalize_row_wave(const FinArgs& p, const int i) {
  const int lane = threadIdx.x & 63;
  float vv[FW_KU], mo[FW_KU], so[FW_KU], dwr[FW_KU];
  const size_t ro = (size_t)i * p.in;
  // (every load is UNCONDITIONAL -- out-of-range lanes read a clamped, valid address and drop the value with a select: a load under a
  // per-lane condition becomes a branch of its own with its own wait)
  const float* mp = p.adam ? p.mv : p.v;
  const float* sp_ = p.adam ? p.sv : p.v;
#pragma unroll
  for (int k = 0; k < FW_KU; ++k) {
    const int c = lane + 64 * k, cc = min(c, p.in - 1);
    const float a0 = p.v[ro + cc], a1 = mp[ro + cc], a2 = sp_[ro + cc];
    const bool ok = c < p.in;
    vv[k] = ok ? a0 : 0.f; mo[k] = ok ? a1 : 0.f; so[k] = ok ? a2 : 0.f;
  }
  float bs = 0.f;                                  // bias gradient: fixed-order sum of the column partials
  for (int q0 = lane; q0 < p.npart; q0 += 256) {
    float t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = p.colsum[(size_t)min(q0 + 64 * u, p.npart - 1) * p.ldcs + i];
#pragma unroll
    for (int u = 0; u < 4; ++u) bs += q0 + 64 * u < p.npart ? t[u] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < FW_KU; ++k) dwr[k] = 0.f;
  constexpr int FW_SB = 32 / FW_KU;                 // splits per batch: 32 loads in flight per lane (40 -- the headline's 10 splits x 8
                                                    // columns in two batches -- costs the fourth resident wave per SIMD: 134 VGPRs)
  for (int sp = 0; sp < p.nsplit; sp += FW_SB) {    // (per element the order sp = 0, 1, ... of finalize_row)
    float t[FW_SB][FW_KU];
#pragma unroll
    for (int u = 0; u < FW_SB; ++u)
#pragma unroll
      for (int k = 0; k < FW_KU; ++k)
        t[u][k] = p.slabs[(size_t)min(sp + u, p.nsplit - 1) * p.slab + (size_t)i * p.ldc + min(lane + 64 * k, p.in - 1)];
#pragma unroll
    for (int u = 0; u < FW_SB; ++u)
#pragma unroll
      for (int k = 0; k < FW_KU; ++k) {
        const int c = lane + 64 * k;
        dwr[k] += (sp + u < p.nsplit && c < p.in && (!p.hoist || c < p.lat0)) ? t[u][k] : 0.f;
      }
  }
  if (p.hoist) {   // x0 columns of a hoisted layer: computed by seg_dw_body
#pragma unroll
    for (int k = 0; k < FW_KU; ++k) {
      const int c = lane + 64 * k - p.lat0;
      if (c >= 0 && c < p.hW) dwr[k] += p.hs[(size_t)i * p.ldh + c];
    }
  }
  bs = wave_sum(bs);
  if (p.g) {
    float dot = 0.f, ss = 0.f;
#pragma unroll
    for (int k = 0; k < FW_KU; ++k) { dot += dwr[k] * vv[k]; ss += vv[k] * vv[k]; }     // (columns >= in hold zeros)
    dot = wave_sum(dot);
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float gi = p.g[i];
    const float dgi = dot / nrm;
    const float a = gi / nrm, b = gi * dgi / (nrm * nrm);
    float ssn = 0.f;
#pragma unroll
    for (int k = 0; k < FW_KU; ++k) {
      const int c = lane + 64 * k;
      if (c < p.in) {
        const float d = a * dwr[k] - b * vv[k];
        if (p.adam) {
          float pn = vv[k];
          const float vn = adam_elem(pn, d, mo[k], so[k], p);
          p.pv[ro + c] = vn; p.mv[ro + c] = mo[k]; p.sv[ro + c] = so[k];
          ssn += vn * vn;
        } else p.dv[ro + c] = p.accumulate ? p.dv[ro + c] + d : d;
      }
    }
    if (p.adam) {
      ssn = wave_sum(ssn);
      if (lane == 0) {
        const float gn = adam_elem(p.pg[i], dgi, p.mg[i], p.sg[i], p);
        p.scale_out[i] = gn / sqrtf(ssn);
      }
    } else if (lane == 0) p.dg[i] = p.accumulate ? p.dg[i] + dgi : dgi;
  } else {
#pragma unroll
    for (int k = 0; k < FW_KU; ++k) {
      const int c = lane + 64 * k;
      if (c < p.in) {
        if (p.adam) {
          float pn = vv[k];
          adam_elem(pn, dwr[k], mo[k], so[k], p);
          p.pv[ro + c] = pn; p.mv[ro + c] = mo[k]; p.sv[ro + c] = so[k];
        } else p.dv[ro + c] = p.accumulate ? p.dv[ro + c] + dwr[k] : dwr[k];
      }
    }
    if (p.adam && lane == 0) p.scale_out[i] = 1.f;
  }
  if (lane == 0) {
    if (p.adam) adam_elem(p.pb[i], bs, p.mb[i], p.sb[i], p);
    else p.db[i] = p.accumulate ? p.db[i] + bs : bs;
  }
}

// every layer in ONE launch: block -> (layer, output rows).  Layers of at most 512 inputs: four rows per block, one wave each
// (finalize_row_wave); wider ones (layer-by-layer path only): one row per block (finalize_row).
struct FinAll { int n; int row0[DSDF_MAX_LAYERS + 1]; FinArgs f[DSDF_MAX_LAYERS]; };   // row0: first BLOCK of each layer
__host__ __device__ inline int fin_blocks(int out, int in) { return in <= 64 * FW_MAXK ? (out + 3) / 4 : out; }
__device__ __forceinline__ void finalize_block(const FinAll& p, const int b, float* red) {
  int l = 0;
  while (l + 1 < p.n && b >= p.row0[l + 1]) ++l;
  const FinArgs& f = p.f[l];
  if (f.in <= 64 * FW_MAXK) {
    const int i = (b - p.row0[l]) * 4 + (int)(threadIdx.x >> 6);
    if (i < f.out) {
      const int ku = (f.in + 63) >> 6;
      if (ku <= 1) finalize_row_wave<1>(f, i);
      else if (ku == 2) finalize_row_wave<2>(f, i);
      else if (ku <= 4) finalize_row_wave<4>(f, i);
      else finalize_row_wave<8>(f, i);
    }
  } else finalize_row(f, b - p.row0[l], red);
}
__global__ __launch_bounds__(256) void finalize_all_kernel(const FinAll p) {
  __shared__ float red[4];
  finalize_block(p, (int)blockIdx.x, red);
}

// ---------------------------------------------------------------------------------------------------
// K5a: per segment r and 64-column chunk: segpart[r][c] = sum over the segment's rows of (dzA + dzB),
// and (chunk 0) the norm of the segment's latent row for the regulariser.
struct SegArgs {
  const float* dzA; const float* dzB; int ldz;   // dzB may be null
  const int64_t* seg_scene; const int64_t* seg_offset; int R; int L;
  const float* table;
  float* segpart;   // [R][L]  (nslice > 1: nslice partial copies slice_stride apart, one per blockIdx.z; seg_scatter_body adds them)
  float* segnorm;   // [R]
  int nslice; long long slice_stride;
};
__global__ __launch_bounds__(256) void seg_reduce_kernel(const SegArgs p) {
  __shared__ float red[4][64];
  const int r = blockIdx.x, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x, cx = tid & 63, ry = tid >> 6;
  // a long segment (config 4 off the workgroup grid: ONE shape of 8001 points was 4 blocks walking 2000 dependent loads each, 0.6 ms)
  // is cut into gridDim.z row ranges; 8 rows per thread in flight, clamped addresses
  const int64_t sbeg = p.seg_offset[r], send = p.seg_offset[r + 1];
  const int64_t per = (send - sbeg + gridDim.z - 1) / gridDim.z;
  const int64_t beg = min(sbeg + (int64_t)blockIdx.z * per, send), end = min(beg + per, send);
  const int col = c0 + cx, cc = min(col, p.L - 1);
  float s = 0.f;
  for (int64_t n = beg + ry; n < end; n += 32) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t m = min(n + 4 * u, end - 1);
      t[u] = p.dzA[(size_t)m * p.ldz + cc];
      if (p.dzB) t[u] += p.dzB[(size_t)m * p.ldz + cc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += n + 4 * u < end ? t[u] : 0.f;
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < p.L)
    p.segpart[(size_t)blockIdx.z * p.slice_stride + (size_t)r * p.L + col] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
  if (blockIdx.y == 0 && blockIdx.z == 0 && tid < 64) {
    const float* row = p.table + (size_t)p.seg_scene[r] * p.L;
    float ss = 0.f;
    for (int c = tid; c < p.L; c += 64) { const float v = row[c]; ss += v * v; }
    ss = wave_sum(ss);
    if (tid == 0) p.segnorm[r] = sqrtf(ss);
  }
}

// K5a': segment-sum path.  Because d/dx0 is linear in dP, the per-scene latent gradient is
//   sum_{n in segment} (dP_0[n] W_0 + dP_k[n] W_k[:, skip])[:L]  =  (sum_n dP_0[n]) W_0[:, :L] + (sum_n dP_k[n]) W_k[:, off:off+L]
// and the per-workgroup (64-row) column sums of dP_0 / dP_k already exist (bias-gradient partials of the fused
// backward).  Valid when every segment is a whole number of 64-row workgroups.  Replaces the per-point d/dx0 GEMM
// columns and seg_reduce_kernel.
struct SegLatArgs {
  const float* cs0; int ldcs; int out0; const float* W0; int ldw0;        // column sums of dP_0, W_0 [out0][ldw0]
  const float* csk; int outk; const float* Wk; int ldwk; int koff;        // skip layer (csk may be null)
  int wg_per_seg; int R; int L;
  const int64_t* seg_scene; const float* table;
  const float* zr;                                                         // non-null: segment r's latent row is zr + r L
  float* segpart; float* segnorm;
  // launch-of-its-own form, few long segments (config 4: ONE shape of 250 workgroups): the segment's workgroups are cut into nslice
  // ranges, block (bx, by = chunk + nchunk * slice) writes the partial products of its range to segpart + slice * slice_stride, and
  // seg_scatter_body adds the slices in order -- 8 x the blocks for a launch that had 16 of them on 256 CUs
  int nslice, nchunk; long long slice_stride;
};
constexpr int SLAT_LDS_FLOATS = 2 * FSEG_MAXW + 16 * 17;
__device__ __forceinline__ void seg_latgrad_body(const SegLatArgs& p, int bx, int by, float* lds) {
  // (bx, by) in (R, ceil(L/16)); block = 16 columns x 16 k-slices: 1024+ blocks of short dot products instead of 256 long ones
  float (*ss)[FSEG_MAXW] = reinterpret_cast<float (*)[FSEG_MAXW]>(lds);
  float (*red)[17] = reinterpret_cast<float (*)[17]>(lds + 2 * FSEG_MAXW);
  const int nsl = p.nslice > 1 ? p.nslice : 1, slice = nsl > 1 ? by / p.nchunk : 0;
  if (nsl > 1) by -= slice * p.nchunk;
  const int gper = (p.wg_per_seg + nsl - 1) / nsl, gbeg = min(slice * gper, p.wg_per_seg), gend = min(gbeg + gper, p.wg_per_seg);
  const int r = bx, c0 = by * 16, tid = threadIdx.x, cx = tid & 15, ks = tid >> 4;
  // a segment's column sums = the sum of its workgroups' column sums, in workgroup order.  The loads go out 32 at a time (the adds
  // keep their order, so the result bits do not change): one load per add left a 256-workgroup segment -- one scene x 16384
  // samples -- waiting for memory 512 times in a row per thread (237 us for the launch, 18 us for 64 scenes x 4 workgroups).
  auto seg_colsum = [&](const float* cs, int j) {
    const float* q = cs + (size_t)r * p.wg_per_seg * p.ldcs + j;
    float s = 0.f;
    int g = gbeg;
    for (; g + 32 <= gend; g += 32) {
      float t[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) t[u] = q[(size_t)(g + u) * p.ldcs];
#pragma unroll
      for (int u = 0; u < 32; ++u) s += t[u];
    }
    for (; g < gend; ++g) s += q[(size_t)g * p.ldcs];
    return s;
  };
  for (int j = tid; j < p.out0; j += 256) ss[0][j] = seg_colsum(p.cs0, j);
  if (p.csk != nullptr)
    for (int j = tid; j < p.outk; j += 256) ss[1][j] = seg_colsum(p.csk, j);
  __syncthreads();
  const int col = c0 + cx;
  float acc = 0.f;
  if (col < p.L) {
    auto dotcol = [&](const float* sv, const float* W, int ldw, int n) {   // 16 independent loads in flight per batch
      float a = 0.f;
      int j = ks;
      for (; j + 240 < n; j += 256) {
        float t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = W[(size_t)(j + 16 * q) * ldw];
#pragma unroll
        for (int q = 0; q < 16; ++q) a += sv[j + 16 * q] * t[q];
      }
      for (; j < n; j += 16) a += sv[j] * W[(size_t)j * ldw];
      return a;
    };
    acc = dotcol(ss[0], p.W0 + col, p.ldw0, p.out0);
    if (p.csk != nullptr) acc += dotcol(ss[1], p.Wk + p.koff + col, p.ldwk, p.outk);
  }
  red[ks][cx] = acc;
  __syncthreads();
  if (ks == 0 && col < p.L) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][cx];
    p.segpart[(size_t)slice * p.slice_stride + (size_t)r * p.L + col] = s;
  }
  if (by == 0 && slice == 0 && tid < 64) {
    const float* row = p.zr != nullptr ? p.zr + (size_t)r * p.L : p.table + (size_t)p.seg_scene[r] * p.L;
    float q = 0.f;
    for (int c = tid; c < p.L; c += 64) { const float v = row[c]; q += v * v; }
    q = wave_sum(q);
    if (tid == 0) p.segnorm[r] = sqrtf(q);
  }
}

// The same result computed the other way round, for batches of at most 256 workgroups (the ones whose roles ride on the dW
// launch's spare workgroups): ONE block per 16-column chunk of the latent takes ALL segments --
//   segpart[R][16] = css_0[R][out0] W_0[:, chunk] + css_k[R][outk] W_k[:, chunk],   css_t[r] = sum of segment r's workgroup column sums
// as a small tiled GEMM: per chunk of 64 rows of W the block stages the 64 x 16 weights and the 64 segments x 64 column sums in LDS
// (every load of the chunk in flight at once: one memory round trip per chunk), then thread (segment, 4 columns) adds its 64 x 4
// products.  16 blocks of ~16 round trips replace the 1024 blocks of ~6 that took 251 us on 16 workgroups (r04_dw_stamps_before.log).
// Fixed summation order (not the order of seg_latgrad_body: the two forms agree to rounding).  Needs R * wg_per_seg <= 256.
constexpr int SLA_LDS_FLOATS = 64 * 65 + 64 * 16 + 4 * 16 * 64;
__device__ __forceinline__ void seg_latgrad_all_body(const SegLatArgs& p, int by, int nby, float* lds) {
  float (*css)[65] = reinterpret_cast<float (*)[65]>(lds);                  // [64 segments][64 rows of the chunk] (+1: banks)
  float (*wch)[16] = reinterpret_cast<float (*)[16]>(lds + 64 * 65);        // [64 rows][16 columns]
  float* part = lds + 64 * 65 + 64 * 16;                                    // [4 slices][16 segments][64 rows]: long segments
  const int tid = threadIdx.x, c0 = by * 16;
  const int li = tid & 63, lq = tid >> 6;                                   // loader: row of the chunk, quarter
  const int sr = tid >> 2, cq = tid & 3;                                    // adder: segment of the group, its 4 columns
  const int wps = p.wg_per_seg;
  for (int s0 = 0; s0 < p.R; s0 += 64) {
    const int ng = min(64, p.R - s0);
    const bool sliced = ng <= 16;       // few segments (of up to 256 workgroups): the 4 quarters share every segment's workgroups
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 2; ++t) {
      const float* cs = t == 0 ? p.cs0 : p.csk;
      if (cs == nullptr) continue;
      const int n = t == 0 ? p.out0 : p.outk, ldw = t == 0 ? p.ldw0 : p.ldwk;
      const float* W = (t == 0 ? p.W0 : p.Wk + p.koff) + c0;
      for (int i0 = 0; i0 < n; i0 += 64) {
        __syncthreads();                                                    // the previous chunk has been consumed
        {   // weights of the chunk: thread (row sr, columns 4 cq ..)
          const int i = i0 + sr;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float wl = W[(size_t)min(i, n - 1) * ldw + min(4 * cq + e, p.L - 1 - c0)];
            wch[sr][4 * cq + e] = (i < n && c0 + 4 * cq + e < p.L) ? wl : 0.f;
          }
        }
        const bool iok = i0 + li < n;
        const float* qs = cs + (size_t)s0 * wps * p.ldcs + min(i0 + li, n - 1);   // workgroup row wg of this group: qs + wg ldcs
        const int lastwg = ng * wps - 1;
        if (!sliced) {      // thread (lq, li): a CONTIGUOUS quarter of the segments, i.e. consecutive workgroup rows: flat batches of 32
          const int mq = (ng + 3) >> 2, sfirst = min(lq * mq, ng), mine = min(mq, ng - sfirst), total = mine * wps;
          const int off0 = sfirst * wps * p.ldcs, offlast = lastwg * p.ldcs;      // (32-bit offsets: at most 256 workgroup rows)
          int sl = sfirst, g = 0;
          float a = 0.f;
          for (int e0 = 0; e0 < total; e0 += 32) {
            float tv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u)         // (unconditional loads: past the end a lane re-reads the group's last workgroup row)
              tv[u] = qs[min(off0 + (e0 + u) * p.ldcs, offlast)];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
              if (e0 + u < total) {
                a += iok ? tv[u] : 0.f;
                if (++g == wps) { css[sl][li] = a; a = 0.f; g = 0; ++sl; }
              }
            }
          }
        } else {            // thread (lq, li): workgroups g = lq, lq + 4, ... of EVERY segment; partial sums, combined in fixed order
          const int gq = lq < wps ? (wps - lq + 3) >> 2 : 0, total = ng * gq;
          int sl = 0, k = 0;
          float a = 0.f;
          for (int e0 = 0; e0 < total; e0 += 32) {
            float tv[32];
            {
              int sl2 = sl, k2 = k;
#pragma unroll
              for (int u = 0; u < 32; ++u) {
                tv[u] = qs[(size_t)min(sl2 * wps + lq + 4 * k2, lastwg) * p.ldcs];
                if (++k2 == gq) { k2 = 0; ++sl2; }
              }
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
              if (e0 + u < total) {
                a += iok ? tv[u] : 0.f;
                if (++k == gq) { part[(lq * 16 + sl) * 64 + li] = a; a = 0.f; k = 0; ++sl; }
              }
            }
          }
          if (gq == 0)
            for (int z = 0; z < ng; ++z) part[(lq * 16 + z) * 64 + li] = 0.f;
          __syncthreads();
          for (int z = lq; z < ng; z += 4)
            css[z][li] = (part[(0 * 16 + z) * 64 + li] + part[(1 * 16 + z) * 64 + li]) + (part[(2 * 16 + z) * 64 + li] + part[(3 * 16 + z) * 64 + li]);
        }
        __syncthreads();
        if (sr < ng) {
#pragma unroll 8
          for (int i = 0; i < 64; ++i) {
            const float a = css[sr][i];
            const float4 w4 = *reinterpret_cast<const float4*>(&wch[i][4 * cq]);
            acc[0] = fmaf(a, w4.x, acc[0]); acc[1] = fmaf(a, w4.y, acc[1]);
            acc[2] = fmaf(a, w4.z, acc[2]); acc[3] = fmaf(a, w4.w, acc[3]);
          }
        }
      }
    }
    if (sr < ng)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + 4 * cq + e < p.L) p.segpart[(size_t)(s0 + sr) * p.L + c0 + 4 * cq + e] = acc[e];
  }
  // the segments' latent norms (regulariser): block by takes segments by, by + nby, ...; one wave each
  for (int r = by + nby * lq; r < p.R; r += 4 * nby) {
    const float* row = p.zr != nullptr ? p.zr + (size_t)r * p.L : p.table + (size_t)p.seg_scene[r] * p.L;
    float q = 0.f;
    for (int c = li; c < p.L; c += 64) { const float v = row[c]; q += v * v; }
    q = wave_sum(q);
    if (li == 0) p.segnorm[r] = sqrtf(q);
  }
}

// rows [g*P/G, (g+1)*P/G) of part[P][ld] summed into out[g][ld] (fixed order): second stage of the head's partials
struct ReduceRowsArgs { const float* part; int P, ld, n; float* out; int G; };
__device__ __forceinline__ void reduce_rows_body(const ReduceRowsArgs& a, int bx, int g, float* lds) {
  float (*red)[64] = reinterpret_cast<float (*)[64]>(lds);
  const int c = bx * 64 + (threadIdx.x & 63), ry = threadIdx.x >> 6;
  const int beg = (int)((long long)g * a.P / a.G), end = (int)((long long)(g + 1) * a.P / a.G);
  float s = 0.f;
  if (c < a.n)
    for (int r = beg + ry; r < end; r += 4) s += a.part[(size_t)r * a.ld + c];
  red[ry][threadIdx.x & 63] = s;
  __syncthreads();
  if (ry == 0 && c < a.n) {
    const int x = threadIdx.x;
    a.out[(size_t)g * a.ld + c] = (red[0][x] + red[1][x]) + (red[2][x] + red[3][x]);
  }
}
// The same for ALL groups of a 64-column strip in one block (the riding form: a sixteenth of the blocks, every load of a thread --
// its column of 4 rows per group x G groups -- in flight at once; groups of at most 16 rows, i.e. P <= 16 G, else the loop form above)
__device__ __forceinline__ void reduce_rows_strip_body(const ReduceRowsArgs& a, int bx, float* lds) {
  float (*red)[64] = reinterpret_cast<float (*)[64]>(lds);
  const int x = threadIdx.x & 63, c = bx * 64 + x, ry = threadIdx.x >> 6;
  const int cc = min(c, a.n - 1);
  for (int g0 = 0; g0 < a.G; g0 += 16) {
    float t[16][4];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int g = min(g0 + u, a.G - 1);
      const int beg = (int)((long long)g * a.P / a.G), end = (int)((long long)(g + 1) * a.P / a.G);
#pragma unroll
      for (int k = 0; k < 4; ++k) t[u][k] = a.part[(size_t)min(beg + ry + 4 * k, a.P - 1) * a.ld + cc];
      (void)end;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (g0 + u < a.G) {                         // (block-uniform)
        const int g = g0 + u;
        const int beg = (int)((long long)g * a.P / a.G), end = (int)((long long)(g + 1) * a.P / a.G);
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) sm += beg + ry + 4 * k < end ? t[u][k] : 0.f;      // rows beg + ry, + 4, ...: reduce_rows_body's order
        for (int r = beg + ry + 16; r < end; r += 4) sm += a.part[(size_t)r * a.ld + cc];   // (groups longer than 16 rows)
        __syncthreads();
        red[ry][x] = sm;
        __syncthreads();
        if (ry == 0 && c < a.n) a.out[(size_t)g * a.ld + c] = (red[0][x] + red[1][x]) + (red[2][x] + red[3][x]);
      }
    }
  }
}
__global__ __launch_bounds__(256) void reduce_rows_kernel(const ReduceRowsArgs a) {
  __shared__ float red[4 * 64];
  reduce_rows_body(a, blockIdx.x, blockIdx.y, red);
}

// Segment mode: everything that consumes only the fused backward's per-workgroup partials, in ONE launch -- the head's
// second reduction stage, the hoisted layers' x0 weight-gradient columns and the per-segment latent gradient.
// The same for LONG segments in a launch of its own (hundreds of workgroups per scene: the reference's 10 x 16000 batches; round 4:
// the 8-rows-per-block form above walked a 4 x 32 net's 5000 workgroups in 8 blocks -- 56 us, the step's second-largest launch): ONE
// output row per block, out[0] + out[1] blocks.  The 256 threads walk a segment's workgroups together, eight segments per pass (all
// their loads in flight, one pair of barriers), fixed-order sums; the latent products in segment order as above.
// (host and device decide alike.  Narrow layers only: at 512 rows the 1024 one-row blocks fetch every partial line eight times over --
// the shipped 8 x 512 shapes measured +70 us per 13.4 ms step with it)
__host__ __device__ inline bool seg_dw_long_form(int wg_per_seg, int out0, int out1) { return wg_per_seg > 16 && out0 + out1 <= 256; }
constexpr int SDWR_SB = 8;
__device__ __forceinline__ void seg_dw_row_body(const SegDwArgs& p, int bidx, float* lds) {
  float (*red)[SDWR_SB] = reinterpret_cast<float (*)[SDWR_SB]>(lds);     // [4 waves][segments of a pass]
  float (*xred)[4] = reinterpret_cast<float (*)[4]>(lds + 4 * SDWR_SB);  // [4 waves][4]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int i = bidx, t = 0;
  if (i >= p.out[0]) { i -= p.out[0]; t = 1; }
  const float* cs = p.cs[t] + i;
  const float* xs = p.xsum[t] + i;
  float* hs = p.HS + (size_t)t * p.hstride + (size_t)i * p.ldh;
  // xyz columns: every workgroup's sums, four workgroups per thread and pass in flight (unconditional loads on clamped rows)
  float xa[4] = {0.f, 0.f, 0.f, 0.f};
  for (int wg0 = tid; wg0 < p.nwg; wg0 += 4 * 256) {
    float tx[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) tx[u][j] = xs[((size_t)min(wg0 + 256 * u, p.nwg - 1) * 4 + j) * p.ldcs];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) xa[j] += (wg0 + 256 * u < p.nwg && j < p.G) ? tx[u][j] : 0.f;
  }
  constexpr int NC = HOIST_MAXL / 256;
  float acc[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) acc[k] = 0.f;
  const float* lat = p.zr != nullptr ? p.zr : p.table;
  const int wps = p.wg_per_seg;
  for (int s0 = 0; s0 < p.R; s0 += SDWR_SB) {
    float a[SDWR_SB];
#pragma unroll
    for (int u = 0; u < SDWR_SB; ++u) {
      const size_t base = (size_t)min(s0 + u, p.R - 1) * wps;
      float v = 0.f;
      for (int g = tid; g < wps; g += 256) v += cs[(base + g) * p.ldcs];
      a[u] = s0 + u < p.R ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SDWR_SB; ++u) a[u] = wave_sum(a[u]);
    __syncthreads();                       // (red is free: the previous pass has read it)
    if (lane == 0)
#pragma unroll
      for (int u = 0; u < SDWR_SB; ++u) red[w][u] = a[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SDWR_SB; ++u) {
      const int sg = s0 + u;
      if (sg < p.R) {
        const float tot = (red[0][u] + red[1][u]) + (red[2][u] + red[3][u]);
        const long long srow = p.zr != nullptr ? (long long)sg * p.L : (long long)p.seg_scene[sg] * p.L;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const int c = tid + 256 * k;
          if (c < p.L) acc[k] = fmaf(tot, lat[srow + c], acc[k]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int c = tid + 256 * k;
    if (c < p.L) hs[c] = acc[k];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) xa[j] = wave_sum(xa[j]);
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) xred[w][j] = xa[j];
  __syncthreads();
  if (tid < p.G) hs[p.L + tid] = (xred[0][tid] + xred[1][tid]) + (xred[2][tid] + xred[3][tid]);
}

struct PostBwdArgs {
  ReduceRowsArgs rr; int rr_bx, rr_n;        // blocks [0, rr_n): (bx, g) = (i % rr_bx, i / rr_bx)
  SegDwArgs dw; int dw_n;                    // next dw_n blocks (0: weights frozen)
  SegLatArgs lat; int lat_bx;                // the rest: (bx, by) = (i % lat_bx, i / lat_bx); riding: one block per 16-column chunk
};                                           // of the latent takes all segments (seg_latgrad_all_body)
// LDS scratch of the roles comes from the caller (the dW kernels lend theirs): the largest role's need
constexpr int ROLE_LDS_FLOATS = SLA_LDS_FLOATS > SDW_LDS_FLOATS ? SLA_LDS_FLOATS : SDW_LDS_FLOATS;
static_assert(ROLE_LDS_FLOATS >= SLAT_LDS_FLOATS && ROLE_LDS_FLOATS >= 4 * 64, "role scratch");
// RIDE: the forms for a few workgroups beside the dW items (32 weight-gradient rows per block, all-segments latent gradient);
// otherwise the forms of a launch of its own, which has the whole chip (8 rows per block, one block per segment and 16 columns).
// The riding form is a NOINLINE call: inlined into the dW kernels its register needs leaked into their k-loops' allocation.
template <bool RIDE>
__device__ __forceinline__ void post_bwd_role(const PostBwdArgs& p, int i, int lat_n, float* lds) {
  if (i < p.rr_n) {
    if constexpr (RIDE) reduce_rows_strip_body(p.rr, i, lds);      // (riding: rr_n = rr_bx strips, every group in one block)
    else reduce_rows_body(p.rr, i % p.rr_bx, i / p.rr_bx, lds);
    return;
  }
  i -= p.rr_n;
  if (i < p.dw_n) {
    if constexpr (!RIDE) {
      if (seg_dw_long_form(p.dw.wg_per_seg, p.dw.out[0], p.dw.out[1])) { seg_dw_row_body(p.dw, i, lds); return; }
    }
    seg_dw_body<RIDE ? SDW_ROWS_RIDE : SDW_ROWS_WIDE>(p.dw, i, lds);
    return;
  }
  i -= p.dw_n;
  if constexpr (RIDE) seg_latgrad_all_body(p.lat, i, lat_n, lds);
  else seg_latgrad_body(p.lat, i % p.lat_bx, i / p.lat_bx, lds);
}
__device__ __attribute__((noinline)) void post_bwd_role_ride(const PostBwdArgs& p, int i, int lat_n, float* lds) {
  post_bwd_role<true>(p, i, lat_n, lds);
}
__global__ __launch_bounds__(256) void post_bwd_kernel(const PostBwdArgs p, const int lat_n) {
  __shared__ __attribute__((aligned(16))) float lds[ROLE_LDS_FLOATS];
  post_bwd_role<false>(p, blockIdx.x, lat_n, lds);
}

// K5b: dlat[scene] += sum over the segments of that scene (in segment order) of
//   segpart[r] + reg_coef/n_norm * count_r * E/||E||.   One block per segment; the FIRST segment of a scene owns
// the sum over all its later duplicates, so the result is deterministic and needs no atomics.
// Block 0 also emits the regulariser loss  sum_r reg_coef/n_norm * count_r * ||E_r||  (train_deep_sdf.py:523-531).
struct ScatterArgs {
  const float* segpart; const float* segnorm; const int64_t* seg_scene; const int64_t* seg_offset; int R; int L;
  float* table; float* dlat; float creg;
  int nslice; long long slice_stride;   // segpart comes in nslice partial copies (SegLatArgs), added here in order; <= 1: one
  const float* zr;             // non-null (segment-mode training steps): segment r's renormed latent row (seg_hoist_kernel); the owner
                               // block of a scene writes it back into the table here -- the in-place embedding_renorm_ of the step
  // block 0: loss_out (+)= sum(part_loss[0..n_part)) * loss_scale + regulariser loss
  const float* part_loss; int n_part; float loss_scale; float* loss_out; int accumulate;
};
__device__ __forceinline__ void seg_scatter_body(const ScatterArgs& p, int r) {
  __shared__ float red[4];
  __shared__ int dup;
  const int tid = threadIdx.x;
  if (r == 0) {
    float s = 0.f;
    if (p.creg != 0.f)
      for (int q = tid; q < p.R; q += 256) s += p.creg * (float)(p.seg_offset[q + 1] - p.seg_offset[q]) * p.segnorm[q];
    s = block_sum_256(s, red);
    float d = 0.f;
    for (int q = tid; q < p.n_part; q += 256) d += p.part_loss[q];
    d = block_sum_256(d, red);
    if (tid == 0) {
      const float v = d * p.loss_scale + s;
      *p.loss_out = p.accumulate ? *p.loss_out + v : v;
    }
  }
  const int64_t j = p.seg_scene[r];
  if (tid == 0) dup = 0;
  __syncthreads();
  int d = 0;
  for (int q = tid; q < r; q += 256) d |= (p.seg_scene[q] == j);
  if (d) dup = 1;
  __syncthreads();
  if (dup) return;
  for (int c = tid; c < p.L; c += 256) {
    const float z = p.zr != nullptr ? p.zr[(size_t)r * p.L + c] : p.table[(size_t)j * p.L + c];   // (duplicates of a scene hold the same row)
    float acc = 0.f;
    for (int q = r; q < p.R; ++q) {
      if (q != r && p.seg_scene[q] != j) continue;
      float v = p.segpart[(size_t)q * p.L + c];
      for (int sl = 1; sl < p.nslice; ++sl) v += p.segpart[(size_t)sl * p.slice_stride + (size_t)q * p.L + c];
      if (p.creg != 0.f) {
        const float nrm = p.segnorm[q];
        if (nrm > 0.f) v += p.creg * (float)(p.seg_offset[q + 1] - p.seg_offset[q]) * z / nrm;
      }
      acc += v;
    }
    p.dlat[(size_t)j * p.L + c] += acc;
    if (p.zr != nullptr) p.table[(size_t)j * p.L + c] = z;
  }
}

__global__ __launch_bounds__(256) void seg_scatter_kernel(const ScatterArgs p) { seg_scatter_body(p, blockIdx.x); }

// finalize_all_kernel with the dense latent-gradient scatter (+ loss) as extra blocks: in segment mode the scatter's inputs
// exist before the finalize launch, so the two share it (the first R blocks take one segment each)
__global__ __launch_bounds__(256) void finalize_scatter_kernel(const FinAll p, const ScatterArgs sc, const int rows) {
  // the scatter blocks come FIRST: they are a chain of dependent loads (latency, not bandwidth) and would otherwise start
  // only after every finalize block has been dispatched, as the tail of the launch
  if ((int)blockIdx.x < sc.R) { seg_scatter_body(sc, (int)blockIdx.x); return; }
  __shared__ float red[4];
  finalize_block(p, (int)blockIdx.x - sc.R, red);
}

// ---------------------------------------------------------------------------------------------------
// K7: Adam, torch/optim/adam.py single-tensor math: m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
// p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float one_minus_b1, float b2, float one_minus_b2, float step_size,
                                                   float bc2_sqrt, float eps, const float* gscale) {
  const float gs = gscale ? *gscale : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * gs;
    const float mi = fmaf(one_minus_b1, gi - m[i], m[i]);
    const float vi = b2 * v[i] + one_minus_b2 * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

// K7c: the latent-only Adam of a graph-captured loop (dsdf_adam_latent_sched): the step's scalars come from a device schedule indexed
// by a device counter, the code regulariser's gradient l2 * z is added on the way.  adam_kernel's math otherwise.
__global__ __launch_bounds__(256) void adam_sched_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long long n, const float* __restrict__ sched,
                                                         long long n_steps, const long long* __restrict__ counter, float one_minus_b1,
                                                         float b2, float one_minus_b2, float eps, float l2) {
  long long it = *counter;
  if (it >= n_steps) it = n_steps - 1;
  const float step_size = sched[2 * it], bc2_sqrt = sched[2 * it + 1];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = fmaf(l2, p[i], g[i]);
    const float mi = fmaf(one_minus_b1, gi - m[i], m[i]);
    const float vi = b2 * v[i] + one_minus_b2 * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}
__global__ void counter_inc_kernel(long long* counter) { *counter += 1; }

// K7b: Adam on the whole decoder arena, one wave per (layer, output row) -- bias, g and the v row -- which also yields
// the row's new weight-norm scale g / ||v|| in the same pass (same summation order as wn_scale_kernel).
struct AdamRowsLayer { long long v_off, g_off, b_off; int out, in, row0; };   // g_off < 0: plain layer
struct AdamRowsArgs {
  int nl, total_rows;
  float* p; const float* g; float* m; float* s;     // parameter / gradient / exp_avg / exp_avg_sq arenas
  float* scale;                                     // [total_rows]
  float omb1, b2, omb2, step_size, bc2_sqrt, eps; const float* gscale;
  AdamRowsLayer ly[DSDF_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void adam_rows_kernel(const AdamRowsArgs a) {
  const int gr = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (gr >= a.total_rows) return;
  int l = 0;
  while (l + 1 < a.nl && gr >= a.ly[l + 1].row0) ++l;
  const AdamRowsLayer& L = a.ly[l];
  const int row = gr - L.row0;
  const float gs = a.gscale ? *a.gscale : 1.f;
  auto upd = [&](long long i) -> float {
    const float gi = a.g[i] * gs;
    const float mi = fmaf(a.omb1, gi - a.m[i], a.m[i]);
    const float vi = a.b2 * a.s[i] + a.omb2 * gi * gi;
    a.m[i] = mi; a.s[i] = vi;
    const float pn = a.p[i] - a.step_size * (mi / (sqrtf(vi) / a.bc2_sqrt + a.eps));
    a.p[i] = pn;
    return pn;
  };
  float ss = 0.f;
  const long long v0 = L.v_off + (long long)row * L.in;
  for (int c = lane; c < L.in; c += 64) { const float x = upd(v0 + c); ss += x * x; }
  ss = wave_sum(ss);
  if (lane == 0) {
    upd(L.b_off + row);
    a.scale[gr] = L.g_off >= 0 ? upd(L.g_off + row) / sqrtf(ss) : 1.f;
  }
}

// K6: global L2 norm of the decoder gradient arena (clip_grad_norm_): two deterministic stages.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long long n, float* part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * g[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* part, int n, float max_norm, float* norm_out,
                                                        float* coef_out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(s);
    *norm_out = nrm;
    *coef_out = fminf(1.f, max_norm / (nrm + 1e-6f));
  }
}

// d_input[n][c] = dzA[n][c] + dzB[n][c]  (module path: d/d(input) = layer-0 dX + skip dX)
__global__ void add2_kernel(const float* a, int lda, const float* b, int ldb, float* o, long long ldo, int n, int w) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * w) return;
  const int r = (int)(i / w), c = (int)(i % w);
  float v = a[(size_t)r * lda + c];
  if (b) v += b[(size_t)r * ldb + c];
  o[(size_t)r * ldo + c] = v;
}

// tangent of the output nonlinearity (deep_sdf_decoder.py:94-95,108-109): y = tanh(t1), t1 = use_tanh ? tanh(u) : u
//   dy = (1 - y^2) (1 - t1^2 if use_tanh) du, with u the last layer's saved pre-activation
__global__ void jvp_tail_kernel(const float* du, const float* u, float* out, int n, int use_tanh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t1 = use_tanh ? tanhf(u[i]) : u[i];
  const float y = tanhf(t1);
  float d = du[i] * (1.f - y * y);
  if (use_tanh) d *= 1.f - t1 * t1;
  out[i] = d;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm variant (deep_sdf_decoder.py:60-65, 97-103: norm_layers WITHOUT weight_norm): nn.LayerNorm(out_dim) between a hidden
// Linear and its ReLU.  Layer-by-layer path only.  One wave per row; biased variance, eps 1e-5, affine (bn{l}.weight / .bias).
//   forward:  y = Linear(x) (gemm_nt, bias) -> xhat = (y - mean) rstd ; z = xhat gamma + beta ; a = dropout(relu(z))
//             xhat replaces y in place and rstd is kept when `save` (training / module path)
//   backward: dz (= d/da masked by [a > 0] * scale, from the next layer's dX epilogue) -> in place
//             dy = rstd (dz gamma - mean(dz gamma) - xhat mean(dz gamma xhat))       (gradient w.r.t. the Linear's output)
//             + per-block column partials of  sum_n dz xhat  (= d gamma)  and  sum_n dy  (= the Linear's bias gradient);
//             d beta = sum_n dz are the column sums the dX epilogue / the output layer already produced.
constexpr float LN_EPS = 1e-5f;
constexpr int LN_MAXW = 2048;
struct LnFwdArgs {
  float* y; int ldy; const float* gamma; const float* beta; float* out; int ldo; int n; int width;
  float* rstd; int save;
  uint32_t drop_key, drop_thr; float drop_scale; uint32_t row_offset;
};
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.n) return;
  float* y = p.y + (size_t)row * p.ldy;
  float s = 0.f;
  for (int c = lane; c < p.width; c += 64) s += y[c];
  const float mean = wave_sum(s) / (float)p.width;
  float q = 0.f;
  for (int c = lane; c < p.width; c += 64) { const float d = y[c] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.width + LN_EPS);
  const uint32_t g = p.row_offset + (uint32_t)row;
  for (int c = lane; c < p.width; c += 64) {
    const float xh = (y[c] - mean) * rstd;
    float a = fmaxf(fmaf(xh, p.gamma[c], p.beta[c]), 0.f);
    if (p.drop_thr != 0u) a = drop_keep(drop_pair_hash(drop_col_key((uint32_t)c, p.drop_key), g), g, p.drop_thr) ? a * p.drop_scale : 0.f;
    p.out[(size_t)row * p.ldo + c] = a;
    if (p.save) y[c] = xh;
  }
  if (p.save && lane == 0) p.rstd[row] = rstd;
}

struct LnBwdArgs {
  float* dz; int ldz; const float* xhat; int ldx; const float* rstd; const float* gamma; int n; int width;
  float* part_dgamma; float* part_db; int ldp;    // [gridDim.x][ldp]
};
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs p) {
  __shared__ float acc_g[4][LN_MAXW];
  __shared__ float acc_b[4][LN_MAXW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = lane; c < p.width; c += 64) { acc_g[wave][c] = 0.f; acc_b[wave][c] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < p.n; row += gridDim.x * 4) {   // fixed grid: fixed summation order
    float* dz = p.dz + (size_t)row * p.ldz;
    const float* xh = p.xhat + (size_t)row * p.ldx;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < p.width; c += 64) { const float d = dz[c] * p.gamma[c]; s1 += d; s2 += d * xh[c]; }
    s1 = wave_sum(s1) / (float)p.width;
    s2 = wave_sum(s2) / (float)p.width;
    const float rs = p.rstd[row];
    for (int c = lane; c < p.width; c += 64) {
      const float z = dz[c], x = xh[c];
      const float dy = rs * (z * p.gamma[c] - s1 - x * s2);
      acc_g[wave][c] += z * x;
      acc_b[wave][c] += dy;
      dz[c] = dy;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.width; c += 256) {
    p.part_dgamma[(size_t)blockIdx.x * p.ldp + c] = (acc_g[0][c] + acc_g[1][c]) + (acc_g[2][c] + acc_g[3][c]);
    p.part_db[(size_t)blockIdx.x * p.ldp + c] = (acc_b[0][c] + acc_b[1][c]) + (acc_b[2][c] + acc_b[3][c]);
  }
}

// forward-mode tangent through LayerNorm + ReLU/dropout (dsdf_module_jvp): ty (tangent of the Linear's output) -> in place
//   tz = gamma rstd (ty - mean(ty) - xhat mean(ty xhat)) ; ta = [a > 0] scale tz      (the primal's decisions, from the stored a)
struct LnJvpArgs { float* t; int ldt; const float* xhat; int ldx; const float* rstd; const float* gamma; const float* act; int ldact;
                   float mask_scale; int n; int width; };
__global__ __launch_bounds__(256) void ln_jvp_kernel(const LnJvpArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.n) return;
  float* t = p.t + (size_t)row * p.ldt;
  const float* xh = p.xhat + (size_t)row * p.ldx;
  const float* a = p.act + (size_t)row * p.ldact;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < p.width; c += 64) { s1 += t[c]; s2 += t[c] * xh[c]; }
  s1 = wave_sum(s1) / (float)p.width;
  s2 = wave_sum(s2) / (float)p.width;
  const float rs = p.rstd[row];
  for (int c = lane; c < p.width; c += 64) {
    const float tz = p.gamma[c] * rs * (t[c] - s1 - xh[c] * s2);
    t[c] = a[c] > 0.f ? tz * p.mask_scale : 0.f;
  }
}

// d gamma[c] = sum of ln_bwd's block partials; d beta[c] = sum of the column-sum partials of dz (cs, ncs rows of ldcs);
// zero != 0: a bn module forward never calls (the LAST Linear's, :60-65 creates it anyway): zero gradient
struct LnGradArgs {
  const float* part_dgamma; int nblk; int ldp; const float* cs; int ncs; int ldcs;
  float* dgamma; float* dbeta; int width; int accumulate; int zero;
};
__global__ void ln_param_grad_kernel(const LnGradArgs p) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.width) return;
  float g = 0.f, b = 0.f;
  if (!p.zero) {
    for (int k = 0; k < p.nblk; ++k) g += p.part_dgamma[(size_t)k * p.ldp + c];
    for (int k = 0; k < p.ncs; ++k) b += p.cs[(size_t)k * p.ldcs + c];
  }
  p.dgamma[c] = p.accumulate ? p.dgamma[c] + g : g;
  p.dbeta[c] = p.accumulate ? p.dbeta[c] + b : b;
}

__global__ void dropout_mask_kernel(uint32_t key, uint32_t thr, int rows, int cols, uint32_t row_offset, uint8_t* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  const uint32_t r = (uint32_t)(i / cols), c = (uint32_t)(i % cols);
  const uint32_t g = row_offset + r;
  out[i] = drop_keep(drop_pair_hash(drop_col_key(c, key), g), g, thr) ? 1 : 0;
}

}  // namespace dsdf
