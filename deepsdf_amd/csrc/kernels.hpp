// kernels.hpp -- the HBM-bound kernels around the GEMMs (gfx950): latent renorm/gather/concat, weight-norm
// materialisation, last-layer GEMV + tanh + clamped-L1 + its backward, split-K/weight-norm finalisation,
// segmented latent-gradient reduction, fused Adam.  All reductions are deterministic (fixed order).
#pragma once
#include "common.hpp"

namespace dsdf {

constexpr int FSEG_MAXW = 512;   // widest layer of the segment-sum latent-gradient path

// ---------------------------------------------------------------------------------------------------
// K0a: max-norm renorm of every looked-up latent row, in place (torch embedding_renorm_,
// train_deep_sdf.py:385,509).  One wave per segment; a segment whose scene already appears in an earlier
// segment is skipped, so every distinct scene is scaled exactly once.
__global__ void latent_renorm_kernel(float* __restrict__ table, int L, const int64_t* __restrict__ seg_scene,
                                     int R, float max_norm) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
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
struct GatherDst { float* ptr; int ld; int col0; };
struct GatherArgs {
  const float* table; int L; const float* xyz; int G;
  const int64_t* seg_scene; const int64_t* seg_offset; int R;
  const float* input; long long ld_in;  // module path: rows come from an explicit [n, L+G] input instead
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
  for (int c = lane; c < W; c += 64) {
    const float v = c < p.L ? src_lat[c] : src_xyz[c - p.L];
    for (int d = 0; d < p.ndst; ++d) p.dst[d].ptr[(size_t)n * p.dst[d].ld + p.dst[d].col0 + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// K1: W = g * v / ||v||_row (torch._weight_norm(v, g, 0)) or plain copy, for ALL layers in two launches:
//   wn_scale_kernel : one wave per (layer,row): scale = g / ||v_row||  (1 for plain layers)
//   wn_tiles_kernel : one block per 32x32 tile: W tile (coalesced) and W^T tile through a 32x33 LDS transpose.
struct WnLayer {
  const float* v; const float* g; float* W; float* WT; int out, in, ldw, ldwt; int row0; int tile0; int tcols;
  float* Wf; float* WTf;   // fragment-ordered copies for the fused kernels (see fused.hpp), or nullptr
  int Uf, UTf;             // k-units (of 16) allocated per n-tile in Wf / WTf
};
struct WnAll { int nl; int total_rows; int total_tiles; float* scale; WnLayer ly[DSDF_MAX_LAYERS]; };

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
  if (L.WT == nullptr && L.Wf == nullptr) return;
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
  if (L.WTf != nullptr) {  // B = W^T: n = in index (tile cols), k = out index (tile rows)
    float4 v4 = make_float4(tile[kk][fr], tile[kk + 1][fr], tile[kk + 2][fr], tile[kk + 3][fr]);
    *reinterpret_cast<float4*>(L.WTf + ((size_t)(cb * L.UTf + 2 * rb + uu) * 2 + i) * 256 + lane * 4) = v4;
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
    if (c < p.in) {
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
      if (p.g) { const float vv = p.v[(size_t)i * p.in + c]; dot += s * vv; ss += vv * vv; }
    }
    dwr[k] = s;
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

// every layer in ONE launch: block -> (layer, output row)
struct FinAll { int n; int row0[DSDF_MAX_LAYERS + 1]; FinArgs f[DSDF_MAX_LAYERS]; };
__global__ __launch_bounds__(256) void finalize_all_kernel(const FinAll p) {
  __shared__ float red[4];
  int l = 0;
  while (l + 1 < p.n && (int)blockIdx.x >= p.row0[l + 1]) ++l;
  finalize_row(p.f[l], blockIdx.x - p.row0[l], red);
}

// ---------------------------------------------------------------------------------------------------
// K5a: per segment r and 64-column chunk: segpart[r][c] = sum over the segment's rows of (dzA + dzB),
// and (chunk 0) the norm of the segment's latent row for the regulariser.
struct SegArgs {
  const float* dzA; const float* dzB; int ldz;   // dzB may be null
  const int64_t* seg_scene; const int64_t* seg_offset; int R; int L;
  const float* table;
  float* segpart;   // [R][L]
  float* segnorm;   // [R]
};
__global__ __launch_bounds__(256) void seg_reduce_kernel(const SegArgs p) {
  __shared__ float red[4][64];
  const int r = blockIdx.x, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x, cx = tid & 63, ry = tid >> 6;
  const int64_t beg = p.seg_offset[r], end = p.seg_offset[r + 1];
  const int col = c0 + cx;
  float s = 0.f;
  if (col < p.L) {
    for (int64_t n = beg + ry; n < end; n += 4) {
      float v = p.dzA[(size_t)n * p.ldz + col];
      if (p.dzB) v += p.dzB[(size_t)n * p.ldz + col];
      s += v;
    }
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < p.L) p.segpart[(size_t)r * p.L + col] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
  if (blockIdx.y == 0 && tid < 64) {
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
  float* segpart; float* segnorm;
};
__global__ __launch_bounds__(256) void seg_latgrad_kernel(const SegLatArgs p) {
  // grid (R, ceil(L/16)); block = 16 columns x 16 k-slices: 1024+ blocks of short dot products instead of 256 long ones
  __shared__ float ss[2][FSEG_MAXW];
  __shared__ float red[16][17];
  const int r = blockIdx.x, c0 = blockIdx.y * 16, tid = threadIdx.x, cx = tid & 15, ks = tid >> 4;
  for (int j = tid; j < p.out0; j += 256) {
    float s = 0.f;
    for (int g = 0; g < p.wg_per_seg; ++g) s += p.cs0[(size_t)(r * p.wg_per_seg + g) * p.ldcs + j];
    ss[0][j] = s;
  }
  if (p.csk != nullptr)
    for (int j = tid; j < p.outk; j += 256) {
      float s = 0.f;
      for (int g = 0; g < p.wg_per_seg; ++g) s += p.csk[(size_t)(r * p.wg_per_seg + g) * p.ldcs + j];
      ss[1][j] = s;
    }
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
    p.segpart[(size_t)r * p.L + col] = s;
  }
  if (blockIdx.y == 0 && tid < 64) {
    const float* row = p.table + (size_t)p.seg_scene[r] * p.L;
    float q = 0.f;
    for (int c = tid; c < p.L; c += 64) { const float v = row[c]; q += v * v; }
    q = wave_sum(q);
    if (tid == 0) p.segnorm[r] = sqrtf(q);
  }
}

// K5b: dlat[scene] += sum over the segments of that scene (in segment order) of
//   segpart[r] + reg_coef/n_norm * count_r * E/||E||.   One block per segment; the FIRST segment of a scene owns
// the sum over all its later duplicates, so the result is deterministic and needs no atomics.
// Block 0 also emits the regulariser loss  sum_r reg_coef/n_norm * count_r * ||E_r||  (train_deep_sdf.py:523-531).
struct ScatterArgs {
  const float* segpart; const float* segnorm; const int64_t* seg_scene; const int64_t* seg_offset; int R; int L;
  const float* table; float* dlat; float creg; float* reg_loss;
};
__global__ __launch_bounds__(256) void seg_scatter_kernel(const ScatterArgs p) {
  __shared__ float red[4];
  __shared__ int dup;
  const int r = blockIdx.x, tid = threadIdx.x;
  if (r == 0) {
    float s = 0.f;
    if (p.creg != 0.f)
      for (int q = tid; q < p.R; q += 256) s += p.creg * (float)(p.seg_offset[q + 1] - p.seg_offset[q]) * p.segnorm[q];
    s = block_sum_256(s, red);
    if (tid == 0) *p.reg_loss = s;
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
    float acc = 0.f;
    for (int q = r; q < p.R; ++q) {
      if (q != r && p.seg_scene[q] != j) continue;
      float v = p.segpart[(size_t)q * p.L + c];
      if (p.creg != 0.f) {
        const float nrm = p.segnorm[q];
        if (nrm > 0.f) v += p.creg * (float)(p.seg_offset[q + 1] - p.seg_offset[q]) * p.table[(size_t)j * p.L + c] / nrm;
      }
      acc += v;
    }
    p.dlat[(size_t)j * p.L + c] += acc;
  }
}

// loss_out (+)= sum(part_loss[0..n)) * scale + *extra
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* part, int n, float scale, const float* extra,
                                                          float* out, int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    float v = s * scale + (extra ? *extra : 0.f);
    *out = accumulate ? *out + v : v;
  }
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

__global__ void dropout_mask_kernel(uint32_t key, uint32_t thr, int rows, int cols, uint32_t row_offset, uint8_t* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  const uint32_t r = (uint32_t)(i / cols), c = (uint32_t)(i % cols);
  const uint32_t g = row_offset + r;
  out[i] = drop_keep(drop_pair_hash(drop_col_key(c, key), g), g, thr) ? 1 : 0;
}

}  // namespace dsdf
