// dsdf_api.hip -- the C ABI of libdsdf_hip.so (include/dsdf.h): workspace planning + launch sequencing.
// No device allocation, no synchronisation, no global mutable state (thread-local error string only).
#include "../../include/dsdf.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>

#include "dwstream.hpp"
#include "sample.hpp"
#include "fused.hpp"
#include "fused_bf16x8.hpp"
#include "gemm.hpp"
#include "kernels.hpp"

using namespace dsdf;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(DSDF_E_LAUNCH, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)
#define LAUNCH_OK(name)                                                                                \
  do {                                                                                                 \
    hipError_t e_ = hipGetLastError();                                                                 \
    if (e_ != hipSuccess) return fail(DSDF_E_LAUNCH, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)
#define TRY(expr)            \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

// ---- optional per-kernel-class timing with HIP events (diagnostics; off by default) ----------------
struct Prof {
  bool on = false;
  static constexpr int NCLS = DSDF_PROF_CLASSES, POOL = 8192;
  hipEvent_t ev[POOL][2];
  int cls[POOL];
  int created = 0, used = 0;
  double flops[NCLS] = {0};
};
thread_local Prof g_prof;

struct ProfScope {
  int slot = -1;
  hipStream_t st;
  ProfScope(int cls, double flops, hipStream_t s) : st(s) {
    Prof& P = g_prof;
    if (!P.on || P.used >= Prof::POOL) return;
    if (P.used >= P.created) {
      if (hipEventCreate(&P.ev[P.created][0]) != hipSuccess || hipEventCreate(&P.ev[P.created][1]) != hipSuccess) return;
      ++P.created;
    }
    slot = P.used++;
    P.cls[slot] = cls;
    P.flops[cls] += flops;
    (void)hipEventRecord(P.ev[slot][0], st);
  }
  ~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[slot][1], st);
  }
};

inline int64_t rup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int NSPLIT_MAX = 32;     // split-K factor of the dW GEMMs
constexpr int LAST_BLOCKS_MAX = 1024;
constexpr int LAT_SLICES_MAX = 8;   // partial copies of the per-segment latent gradient (SegLatArgs.nslice)
constexpr int LAST_GROUPS = 16;    // second-stage partial groups of the last-layer reduction
constexpr int LN_BLOCKS = 256;     // LayerNorm backward: fixed grid = fixed summation order of its column partials

int validate(const DsdfNet* n) {
  if (!n) return fail(DSDF_E_INVALID, "net is NULL");
  if (n->n_layers < 2 || n->n_layers > DSDF_MAX_LAYERS) return fail(DSDF_E_INVALID, "n_layers %d out of range", n->n_layers);
  const int W0 = n->latent_size + n->geom_dim;
  if (n->latent_size < 0 || n->geom_dim < 1) return fail(DSDF_E_INVALID, "bad latent_size/geom_dim");
  // the d/d(xyz) scratch of an xyz_in_all net (module_backward with d_input, the tangent pass) is laid out [N][4]
  if (n->xyz_in_all && n->geom_dim > 4) return fail(DSDF_E_INVALID, "xyz_in_all needs geom_dim <= 4 (got %d)", n->geom_dim);
  if (n->in_dim[0] != W0) return fail(DSDF_E_INVALID, "in_dim[0] %d != latent_size+geom_dim %d", n->in_dim[0], W0);
  if (n->out_dim[n->n_layers - 1] != 1) return fail(DSDF_E_INVALID, "last layer must have out_dim 1");
  if (n->skip_mask & 1u) return fail(DSDF_E_INVALID, "latent_in may not contain layer 0");
  if (n->skip_mask >> n->n_layers) return fail(DSDF_E_INVALID, "skip_mask names a layer >= n_layers");
  if (__builtin_popcount(n->skip_mask) > 1) return fail(DSDF_E_INVALID, "at most one latent_in layer is supported");
  for (int l = 0; l < n->n_layers; ++l) {
    if (n->in_dim[l] < 1 || n->out_dim[l] < 1 || n->in_dim[l] > 2048 || n->out_dim[l] > 65536)
      return fail(DSDF_E_INVALID, "layer %d: unsupported size %d -> %d", l, n->in_dim[l], n->out_dim[l]);
    if (n->fwd_bf16 && (n->in_dim[l] > 512 || (l < n->n_layers - 1 && n->out_dim[l] > 512)))
      return fail(DSDF_E_INVALID, "fwd_bf16 needs every layer width <= 512 (layer %d: %d -> %d)", l, n->in_dim[l], n->out_dim[l]);
    if (l > 0) {
      const int expect = n->out_dim[l - 1] + ((n->skip_mask >> l) & 1 ? W0 : (n->xyz_in_all ? n->geom_dim : 0));
      if (n->in_dim[l] != expect) return fail(DSDF_E_INVALID, "layer %d: in_dim %d != %d", l, n->in_dim[l], expect);
    }
  }
  if ((n->latent_dropout || n->xyz_in_all || n->ln_param_mask) && n->fwd_bf16)
    return fail(DSDF_E_INVALID, "fwd_bf16 is not available with latent_dropout / xyz_in_all / LayerNorm");
  if (n->fwd_bf16 && ((n->skip_mask >> (n->n_layers - 1)) & 1))   // the bf16 kernels' output layer is a dot product over the activations only
    return fail(DSDF_E_INVALID, "fwd_bf16 is not available when latent_in names the output layer");
  if (n->gemm_split) {
    if (n->latent_dropout || n->xyz_in_all || n->ln_param_mask)
      return fail(DSDF_E_INVALID, "gemm_split is not available with latent_dropout / xyz_in_all / LayerNorm");
    for (int l = 0; l < n->n_layers; ++l)
      if (n->in_dim[l] > 512 || (l < n->n_layers - 1 && n->out_dim[l] > 512))
        return fail(DSDF_E_INVALID, "gemm_split needs every layer width <= 512 (layer %d: %d -> %d)", l, n->in_dim[l], n->out_dim[l]);
  }
  if (n->ln_param_mask) {
    if (n->weight_norm_mask) return fail(DSDF_E_INVALID, "LayerNorm (ln_param_mask) and weight norm exclude each other");
    if (n->ln_param_mask >> n->n_layers) return fail(DSDF_E_INVALID, "ln_param_mask names a layer >= n_layers");
    for (int l = 0; l < n->n_layers; ++l)
      if (((n->ln_param_mask >> l) & 1) && n->out_dim[l] > 2048) return fail(DSDF_E_INVALID, "LayerNorm width %d > 2048", n->out_dim[l]);
  }
  if (n->in_dim[n->n_layers - 1] % 4 != 0) return fail(DSDF_E_INVALID, "last hidden width must be a multiple of 4");
  if (!(n->dropout_p >= 0.f && n->dropout_p < 1.f)) return fail(DSDF_E_INVALID, "dropout_p must be in [0,1)");
  return 0;
}

struct Packed {
  int64_t w_off[DSDF_MAX_LAYERS], wt_off[DSDF_MAX_LAYERS];
  int ldw[DSDF_MAX_LAYERS], ldwt[DSDF_MAX_LAYERS];
  int64_t scale_off;   // per-row weight-norm scales of all layers
  // fragment-ordered copies for the fused kernels: Wf (n = out, k = in), WTf (n = in, k = out)
  int64_t wf_off[DSDF_MAX_LAYERS], wtf_off[DSDF_MAX_LAYERS];
  int64_t wfb_off[DSDF_MAX_LAYERS];   // bf16 fragment copy of W for the bf16 forward (fused_bf16x8.hpp); offset in floats
  int uf[DSDF_MAX_LAYERS], utf[DSDF_MAX_LAYERS];   // k-units of 16 per n-tile
  int64_t ws_off[DSDF_MAX_LAYERS], wts_off[DSDF_MAX_LAYERS];       // gemm_split: 3 bf16 planes of W / W^T in fragment order (offsets in floats)
  int64_t ws_plane[DSDF_MAX_LAYERS], wts_plane[DSDF_MAX_LAYERS];   //   bf16 elements per plane
  int64_t total;
};
Packed packed_layout(const DsdfNet* n) {
  Packed p;
  int64_t o = 0;
  for (int l = 0; l < n->n_layers; ++l) {
    p.ldw[l] = (int)rup(n->in_dim[l], 32);
    p.ldwt[l] = (int)rup(n->out_dim[l], 32);
    p.w_off[l] = o;  o += rup((int64_t)n->out_dim[l] * p.ldw[l], 64);
    p.wt_off[l] = o; o += rup((int64_t)n->in_dim[l] * p.ldwt[l], 64);
  }
  p.scale_off = o;
  for (int l = 0; l < n->n_layers; ++l) o += n->out_dim[l];
  o = rup(o, 64);
  for (int l = 0; l < n->n_layers; ++l) {
    const int64_t ntw = rup((n->out_dim[l] + 31) / 32, 4), ntt = rup((n->in_dim[l] + 31) / 32, 4);
    p.uf[l] = 2 * ((n->in_dim[l] + 31) / 32);
    p.utf[l] = 2 * ((n->out_dim[l] + 31) / 32);
    p.wf_off[l] = o;  o += ntw * p.uf[l] * 512;
    p.wtf_off[l] = o; o += ntt * p.utf[l] * 512;
    p.wfb_off[l] = o; o += n->fwd_bf16 ? ntw * 32 * 256 : 0;   // 32 phase-major k-unit slots of 1 KiB per n-tile
    p.ws_plane[l] = ntw * p.uf[l] * 512; p.wts_plane[l] = ntt * p.utf[l] * 512;     // (1 KiB = 512 bf16 per tile and k-unit)
  }
  o = rup(o, 64);
  for (int l = 0; l < n->n_layers; ++l) {   // gemm_split: the bf16 planes come LAST, so everything in front keeps its place
    p.ws_off[l] = o;  o += n->gemm_split ? 3 * p.ws_plane[l] / 2 : 0;
    p.wts_off[l] = o; o += n->gemm_split ? 3 * p.wts_plane[l] / 2 : 0;
  }
  p.total = rup(o, 64);
  return p;
}

void param_layout(const DsdfNet* n, DsdfParamLayout* L) {
  int64_t o = 0;
  for (int l = 0; l < DSDF_MAX_LAYERS; ++l) L->bias_off[l] = L->g_off[l] = L->v_off[l] = L->ln_w_off[l] = L->ln_b_off[l] = -1;
  for (int l = 0; l < n->n_layers; ++l) {
    const int64_t out = n->out_dim[l], in = n->in_dim[l];
    if ((n->weight_norm_mask >> l) & 1) {
      L->bias_off[l] = o; o += out;
      L->g_off[l] = o;    o += out;
      L->v_off[l] = o;    o += out * in;
    } else {
      L->v_off[l] = o;    o += out * in;
      L->bias_off[l] = o; o += out;
    }
    if ((n->ln_param_mask >> l) & 1) {   // bn{l}.weight, bn{l}.bias follow lin{l}'s parameters (module registration order)
      L->ln_w_off[l] = o; o += out;
      L->ln_b_off[l] = o; o += out;
    }
  }
  L->total = o;
}

// ---- dW work schedule (dwstream.hpp): (layer, K-split, 128x128 tile) items, ~one per wave of the chip ------------
struct DwSched {
  int tiles_m[DSDF_MAX_LAYERS], tiles_n[DSDF_MAX_LAYERS], last_nj[DSDF_MAX_LAYERS], nfull_n[DSDF_MAX_LAYERS],
      nsplit[DSDF_MAX_LAYERS], kchunk[DSDF_MAX_LAYERS], full0[DSDF_MAX_LAYERS], narrow0[DSDF_MAX_LAYERS];
  long long slab[DSDF_MAX_LAYERS];
  int n_full, n_narrow;
};
// waves of the CURRENT device (4 per CU).  The CU count is an immutable property of a device, cached per device ordinal
// (relaxed atomics: every writer stores the same value), so processes / threads driving different devices each get
// their own device's schedule -- no first-device-wins global.
int chip_waves() {
  constexpr int MAXDEV = 64;
  static std::atomic<int> cache[MAXDEV];
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 4 * 256;   // MI355X
  if (dev < MAXDEV) {
    const int w = cache[dev].load(std::memory_order_relaxed);
    if (w > 0) return w;
  }
  int waves = 4 * 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) waves = 4 * cus;
  if (dev < MAXDEV) cache[dev].store(waves, std::memory_order_relaxed);
  return waves;
}
int skip_layer(const DsdfNet* n) {
  for (int l = 1; l < n->n_layers - 1; ++l)
    if ((n->skip_mask >> l) & 1) return l;
  return -1;
}
// latent_in names the OUTPUT layer (valid in the reference: deep_sdf_decoder.py:88-89 runs for every layer incl. the last Linear): its
// input is [a | x0].  No segment mode, no dsdf_decode_latent for such a net; the heads route the x0 columns' gradient (kernels.hpp LastArgs).
bool last_layer_skip(const DsdfNet* n) { return ((n->skip_mask >> (n->n_layers - 1)) & 1) != 0; }
// columns of layer l's input that the dW GEMM contracts over the points: all of them, or -- segment mode -- only the
// previous layer's activations (the x0 columns of layer 0 / the skip layer are hoisted: finalize_row, kernels.hpp)
int dw_cols(const DsdfNet* n, int l, bool segmode) {
  if (!segmode) return n->in_dim[l];
  if (l == 0) return 0;
  return l == skip_layer(n) ? n->out_dim[l - 1] : n->in_dim[l];
}
// Layers [l0, l1) only (the others get no items): the two-phase backward of a data-parallel step (DsdfLossCfg.dw_phase) schedules
// each half of the layers so that it fills the chip by itself.
// reserve: waves the K-split leaves free on purpose -- phase 1 of a phased backward in segment mode keeps 16 workgroups for the riding
// post-backward roles (without them the roles are a launch of their own: +21 us per step)
DwSched dw_schedule(const DsdfNet* n, int64_t N, const int* ld_in, bool segmode, int l0 = 0, int l1 = DSDF_MAX_LAYERS, int reserve = 0) {
  DwSched S;
  memset(&S, 0, sizeof(S));
  const int nh = n->n_layers - 1;
  int Tfull = 0, Tnarrow = 0;
  for (int l = 0; l < nh; ++l) {
    const int nc = (l >= l0 && l < l1) ? dw_cols(n, l, segmode) : 0;
    S.slab[l] = rup((int64_t)n->out_dim[l] * ld_in[l], 64);
    if (nc == 0) { S.last_nj[l] = 4; continue; }   // no items
    S.tiles_m[l] = (n->out_dim[l] + 127) / 128;
    S.tiles_n[l] = (nc + 127) / 128;
    S.last_nj[l] = ((nc - (S.tiles_n[l] - 1) * 128) + 31) / 32;
    S.nfull_n[l] = S.last_nj[l] == 4 ? S.tiles_n[l] : S.tiles_n[l] - 1;
    Tfull += S.tiles_m[l] * S.nfull_n[l];
    if (S.last_nj[l] != 4) Tnarrow += S.tiles_m[l];
  }
  // one K-split count for every layer: the largest that still gives every wave of the chip at most one full item.  A net WITHOUT
  // full-width tiles (every layer narrower than 128: the reference's shipped 4 x 64 / 4 x 32 specs) is split by its narrow items
  // instead -- round 3 left such nets at ONE split, i.e. one wave contracting all the points of a layer.
  const int Tsplit = Tfull > 0 ? Tfull : Tnarrow;
  const int waves = chip_waves() - reserve > 0 ? chip_waves() - reserve : chip_waves();
  int ns = Tsplit > 0 && waves / Tsplit > 0 ? waves / Tsplit : 1;
  const int maxsplit = N / 64 > 0 ? (int)(N / 64) : 1;
  if (ns > maxsplit) ns = maxsplit;
  int kchunk = (int)rup((N + ns - 1) / ns, 2);
  if (kchunk < 2) kchunk = 2;   // N == 0 (size queries for an empty batch)
  ns = (int)((N + kchunk - 1) / kchunk);
  if (ns < 1) ns = 1;
  int nf = 0, nn = 0;
  for (int l = 0; l < nh; ++l) {
    S.nsplit[l] = S.tiles_m[l] > 0 ? ns : 0; S.kchunk[l] = kchunk;
    S.full0[l] = nf;   nf += ns * S.tiles_m[l] * S.nfull_n[l];
    S.narrow0[l] = nn; nn += S.last_nj[l] == 4 ? 0 : ns * S.tiles_m[l];
  }
  S.n_full = nf; S.n_narrow = nn;
  return S;
}

// K-bucket gradient exchange (data parallel): the decoder's layers are cut into K contiguous groups, handed out LAST layer
// first (the order the backward finishes them in): bucket b = layers [cut[b + 1], cut[b]), cut[0] = n_layers, cut[K] = 0.
// K = 2 on the 8 x 512 net: layers [4, 8] (1.05 M parameters) then [0, 4) (0.79 M).  A net with fewer layers than buckets
// leaves the trailing buckets empty (cut repeats 0).
void dw_bucket_cuts(const DsdfNet* n, int K, int* cut) {
  for (int b = 0; b <= K; ++b) cut[b] = (int)((int64_t)n->n_layers * (K - b) / K);
}

// ---- workspace plan -----------------------------------------------------------------------------
struct Plan {
  int nl, W0, N, R;
  int ld_in[DSDF_MAX_LAYERS];
  size_t in_off[DSDF_MAX_LAYERS];
  int ld_dp, ldz, ldcs, ld_part, last_blocks, nsplit, kchunk, mt;
  long long slab;
  size_t u_off, y_off, dp_off[2], dzA_off, dzB_off, slab_off, colsum_off, part_off, part2_off, partdb_off,
      partloss_off, segpart_off, segnorm_off, gnorm_off, dxz_off[2], lnpg_off, lnpb_off, total;
  size_t lnx_off[DSDF_MAX_LAYERS], lnr_off[DSDF_MAX_LAYERS];   // LayerNorm: xhat [N][ld_in[l+1]] (the Linear's output in place), rstd [N]
  // fused backward: per hidden layer l a global dP_l buffer, the forward's mask bits and per-workgroup column sums
  size_t dpl_off[DSDF_MAX_LAYERS], mask_off[DSDF_MAX_LAYERS], cs_off[DSDF_MAX_LAYERS], dwslab_off[DSDF_MAX_LAYERS];
  int nwg, frows;    // workgroups of the fused kernels and their rows (64; 32 for batches that would leave CUs idle: pick_frows)
  DwSched dw;
  DwSched dwph[DSDF_MAX_BUCKETS];   // phased backward: the schedule of bucket b's layers [dw_cut[b + 1], dw_cut[b])
  int dw_nb, dw_cut[DSDF_MAX_BUCKETS + 1];   // dw_bucket_cuts
  // segment mode: U[R][2][ldu] of the hoisted layers, per-workgroup xyz sums [nwg][4][ldcs] for each of them
  int segmode, ldu, ldh;
  long long hstride;
  size_t hoistU_off, xsum_off[2], hs_off, zr_off;   // hs: [2][maxout][ldh] x0 columns of the hoisted layers' weight gradients
};

// nb: the bucket count the workspace is laid out for (DsdfLossCfg.dw_buckets; every call of one step passes the same one).
// 2 also serves the un-phased step, so dsdf_workspace_bytes' answer covers K <= 2.
Plan make_plan(const DsdfNet* n, int64_t N, int64_t R, bool inference, bool segmode = false, int nb = 2, int frows = FROWS) {
  Plan P;
  memset(&P, 0, sizeof(P));
  P.frows = frows;
  P.nl = n->n_layers; P.W0 = n->latent_size + n->geom_dim; P.N = (int)N; P.R = (int)R;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (size_t)rup((int64_t)bytes, 256); return r; };
  int maxw = 4;
  for (int l = 0; l < P.nl; ++l) {
    P.ld_in[l] = (int)rup(n->in_dim[l], 4);
    if (P.ld_in[l] > maxw) maxw = P.ld_in[l];
  }
  if (inference) {
    size_t pp[2] = {take((size_t)N * maxw * 4), take((size_t)N * maxw * 4)};
    for (int l = 0; l < P.nl; ++l) {
      // (xyz_in_all: every layer input carries xyz columns the gather pre-fills, so none of them can share a ping-pong buffer)
      if (l == 0 || ((n->skip_mask >> l) & 1) || n->xyz_in_all) P.in_off[l] = take((size_t)N * P.ld_in[l] * 4);
      else P.in_off[l] = pp[l & 1];
    }
    if (n->ln_param_mask) {   // one scratch row block for the Linear's output before LayerNorm (nothing is kept in inference)
      const size_t t = take((size_t)N * maxw * 4);
      for (int l = 0; l < P.nl; ++l) P.lnx_off[l] = t;
    }
    P.total = o;
    return P;
  }
  for (int l = 0; l < P.nl; ++l) P.in_off[l] = take((size_t)N * P.ld_in[l] * 4 + 4096);   // + slack: edge tiles of dw_stream over-read
  P.u_off = take((size_t)N * 4);
  P.y_off = take((size_t)N * 4);
  P.ld_dp = maxw;
  P.dp_off[0] = take((size_t)N * maxw * 4);
  P.dp_off[1] = take((size_t)N * maxw * 4);
  P.ldz = (int)rup(P.W0, 4);
  P.dzA_off = take((size_t)N * P.ldz * 4);
  P.dzB_off = take((size_t)N * P.ldz * 4);
  for (int t = 0; t < 2; ++t) P.dxz_off[t] = n->xyz_in_all ? take((size_t)N * 4 * 4) : 0;   // [N][4]: one layer's d/d(xyz), running sum
  for (int l = 0; l + 1 < P.nl; ++l)
    if ((n->ln_param_mask >> l) & 1) { P.lnx_off[l] = take((size_t)N * P.ld_in[l + 1] * 4); P.lnr_off[l] = take((size_t)N * 4); }
  // split-K of the dW GEMMs: chunks of >= 256 points, at most NSPLIT_MAX slabs
  int ns = (int)((N + 255) / 256);
  if (ns > NSPLIT_MAX) ns = NSPLIT_MAX;
  if (ns < 1) ns = 1;
  P.kchunk = (int)rup((N + ns - 1) / ns, BK);
  if (P.kchunk < BK) P.kchunk = BK;   // N == 0 (size query for an empty batch)
  P.nsplit = (int)((N + P.kchunk - 1) / P.kchunk);
  if (P.nsplit < 1) P.nsplit = 1;
  int64_t maxslab = 0;
  int maxout = 1;
  for (int l = 0; l < P.nl - 1; ++l) {
    const int64_t s = (int64_t)n->out_dim[l] * P.ld_in[l];
    if (s > maxslab) maxslab = s;
    if (n->out_dim[l] > maxout) maxout = n->out_dim[l];
  }
  P.slab = rup(maxslab, 64);
  P.slab_off = take((size_t)P.nsplit * P.slab * 4);
  P.mt = (int)((N + BM - 1) / BM);
  P.ldcs = (int)rup(maxout, 4);
  P.colsum_off = take((size_t)P.mt * P.ldcs * 4);
  if (n->ln_param_mask) { P.lnpg_off = take((size_t)LN_BLOCKS * P.ldcs * 4); P.lnpb_off = take((size_t)LN_BLOCKS * P.ldcs * 4); }
  P.last_blocks = (int)((N + 15) / 16);
  if (P.last_blocks > LAST_BLOCKS_MAX) P.last_blocks = LAST_BLOCKS_MAX;
  if (P.last_blocks < 1) P.last_blocks = 1;
  P.ld_part = 2 * P.ld_in[P.nl - 1];  // [dW_last | colsum_prev]
  // one row of head partials per block of last_layer_kernel (<= LAST_BLOCKS_MAX) OR per workgroup of the fused backward (N / 64,
  // unbounded): sized for the larger.  (Rounds 1-3 sized them by last_blocks alone: batches of more than 65536 points -- the shipped
  // 10 x 16000 -- let the fused head write its partials past these buffers, into part2 / partdb / partloss and the dP_0 buffer.)
  const size_t part_rows = (size_t)std::max<int64_t>(P.last_blocks, (N + frows - 1) / frows);
  P.part_off = take(part_rows * P.ld_part * 4);
  P.part2_off = take((size_t)LAST_GROUPS * P.ld_part * 4);
  P.partdb_off = take(part_rows * 4);
  P.partloss_off = take(part_rows * 4);
  P.segpart_off = take((size_t)LAT_SLICES_MAX * (R > 0 ? R : 1) * (n->latent_size > 0 ? n->latent_size : 1) * 4);   // (up to 8 partial copies)
  P.segnorm_off = take((size_t)(R > 0 ? R : 1) * 4);
  P.gnorm_off = take(1024 * 4);
  P.nwg = (int)((N + frows - 1) / frows);
  for (int l = 0; l < P.nl - 1; ++l) {
    P.dpl_off[l] = take((size_t)N * maxw * 4 + 4096);
    P.mask_off[l] = take((size_t)P.nwg * 256 * 16);
    P.cs_off[l] = take((size_t)P.nwg * P.ldcs * 4);
  }
  P.dw = dw_schedule(n, N, P.ld_in, segmode);
  P.dw_nb = nb < 2 ? 2 : (nb > DSDF_MAX_BUCKETS ? DSDF_MAX_BUCKETS : nb);
  dw_bucket_cuts(n, P.dw_nb, P.dw_cut);
  for (int t = 0; t < P.dw_nb; ++t)
    // (two buckets only: the riding roles take ~180 us, a bucket's launch must be at least that long or they become its tail --
    // measured: K = 2 1.3100 -> 1.3032 ms/step with the reserve, K = 4 1.3776 -> 1.4116)
    P.dwph[t] = dw_schedule(n, N, P.ld_in, segmode, P.dw_cut[t + 1], P.dw_cut[t],
                            t == 0 && P.dw_nb == 2 && segmode && P.nwg <= chip_waves() / 4 ? 64 : 0);
  for (int l = 0; l < P.nl - 1; ++l) {   // slabs sized for whichever schedule splits K finest: the layout does not depend on the phase
    int ns = P.dw.nsplit[l];
    for (int t = 0; t < P.dw_nb; ++t)
      if (P.dwph[t].nsplit[l] > ns) ns = P.dwph[t].nsplit[l];
    P.dwslab_off[l] = take((size_t)ns * P.dw.slab[l] * 4);
  }
  P.segmode = segmode ? 1 : 0;
  if (segmode) {
    P.ldu = P.ldcs;
    P.hoistU_off = take((size_t)(R > 0 ? R : 1) * 2 * P.ldu * 4);
    for (int t = 0; t < 2; ++t) P.xsum_off[t] = take((size_t)P.nwg * 4 * P.ldcs * 4);
    P.ldh = (int)rup(n->latent_size + n->geom_dim, 4);
    P.hstride = (long long)P.ldcs * P.ldh;
    P.hs_off = take((size_t)2 * P.hstride * 4);
    P.zr_off = take((size_t)(R > 0 ? R : 1) * (n->latent_size > 0 ? n->latent_size : 1) * 4);   // renormed latent row of every segment
  }
  P.total = o;
  return P;
}

template <typename T>
inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

// ---- launch helpers -------------------------------------------------------------------------------
template <int EPI>
int launch_nt(const NtArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return 0;
  if (a.K <= 0) return fail(DSDF_E_INVALID, "gemm_nt: K must be positive");
  if ((a.lda & 3) || (a.ldb & 3) || !aligned16(a.A) || !aligned16(a.B))
    return fail(DSDF_E_INVALID, "gemm_nt: operands must be 16-byte aligned with ld %% 4 == 0");
  if (a.lda < rup(a.K, 4) || a.ldb < rup(a.K, 4)) return fail(DSDF_E_INVALID, "gemm_nt: ld smaller than K rounded up to 4");
  const int grid = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  ProfScope ps(DSDF_PROF_GEMM_NT, 2.0 * a.M * a.N * a.K, st);
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(grid), dim3(256), 0, st, a);
  LAUNCH_OK("gemm_nt_kernel");
  return 0;
}

int launch_tn(const TnArgs& a, int nsplit, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || nsplit <= 0) return 0;
  if ((a.lda & 3) || (a.ldb & 3) || !aligned16(a.A) || !aligned16(a.B))
    return fail(DSDF_E_INVALID, "gemm_tn: operands must be 16-byte aligned with ld %% 4 == 0");
  if (a.lda < rup(a.M, 4) || a.ldb < rup(a.N, 4)) return fail(DSDF_E_INVALID, "gemm_tn: ld smaller than the tile width");
  if (a.kchunk % BK) return fail(DSDF_E_INVALID, "gemm_tn: kchunk must be a multiple of %d", BK);
  const int grid = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * nsplit;
  ProfScope ps(DSDF_PROF_GEMM_TN, 2.0 * a.M * a.N * a.K, st);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(grid), dim3(256), 0, st, a);
  LAUNCH_OK("gemm_tn_kernel");
  return 0;
}

template <int MODE>
int launch_last(const LastArgs& a, int blocks, hipStream_t st) {
  if (a.n <= 0) return 0;
  const int in = a.in;
  ProfScope ps(DSDF_PROF_LAST, (MODE == LAST_FWD ? 2.0 : 6.0) * a.n * a.in, st);
  if (in <= 256) hipLaunchKernelGGL((last_layer_kernel<MODE, 1>), dim3(blocks), dim3(256), 0, st, a);
  else if (in <= 512) hipLaunchKernelGGL((last_layer_kernel<MODE, 2>), dim3(blocks), dim3(256), 0, st, a);
  else if (in <= 1024) hipLaunchKernelGGL((last_layer_kernel<MODE, 4>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((last_layer_kernel<MODE, 8>), dim3(blocks), dim3(256), 0, st, a);
  LAUNCH_OK("last_layer_kernel");
  return 0;
}

// Adam constants of one parameter group (torch/optim/adam.py single-tensor form; bias corrections in double on the host)
AdamRide adam_ride(float* p, const float* g, float* m, float* v, int64_t n, float lr, const DsdfAdamCfg* c) {
  AdamRide a;
  memset(&a, 0, sizeof(a));
  const double bc1 = 1.0 - pow((double)c->beta1, (double)c->step), bc2 = 1.0 - pow((double)c->beta2, (double)c->step);
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
  a.blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  a.omb1 = 1.0f - c->beta1; a.b2 = c->beta2; a.omb2 = 1.0f - c->beta2;
  a.step_size = (float)((double)lr / bc1); a.bc2_sqrt = (float)sqrt(bc2); a.eps = c->eps;
  return a;
}

// ride != nullptr: a dense Adam update (the latent table) done by extra blocks of the wn_tiles launch
int materialize(const DsdfNet* net, const float* params, float* packed, hipStream_t st, bool scales_ready = false,
                const AdamRide* ride = nullptr) {
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  WnAll a;
  memset(&a, 0, sizeof(a));
  a.nl = net->n_layers;
  a.scale = packed + pk.scale_off;
  int rows = 0, tiles = 0;
  for (int l = 0; l < net->n_layers; ++l) {
    WnLayer& y = a.ly[l];
    y.v = params + L.v_off[l];
    y.g = L.g_off[l] >= 0 ? params + L.g_off[l] : nullptr;
    y.W = packed + pk.w_off[l];
    y.WT = (l == net->n_layers - 1) ? nullptr : packed + pk.wt_off[l];
    y.out = net->out_dim[l]; y.in = net->in_dim[l]; y.ldw = pk.ldw[l]; y.ldwt = pk.ldwt[l];
    y.row0 = rows; y.tile0 = tiles; y.tcols = (y.in + 31) / 32;
    const bool last = l == net->n_layers - 1;
    y.Wf = last ? nullptr : packed + pk.wf_off[l];
    y.WTf = last ? nullptr : packed + pk.wtf_off[l];
    y.Wfb = (last || !net->fwd_bf16) ? nullptr : reinterpret_cast<__bf16*>(packed + pk.wfb_off[l]);
    y.Uf = pk.uf[l]; y.UTf = pk.utf[l];
    const bool sp = net->gemm_split && !last;
    y.Ws = sp ? reinterpret_cast<__bf16*>(packed + pk.ws_off[l]) : nullptr;
    y.WTs = sp ? reinterpret_cast<__bf16*>(packed + pk.wts_off[l]) : nullptr;
    y.ws_plane = pk.ws_plane[l]; y.wts_plane = pk.wts_plane[l];
    rows += y.out;
    tiles += ((y.out + 31) / 32) * y.tcols;
  }
  a.total_rows = rows; a.total_tiles = tiles;
  if (!scales_ready) {   // dsdf_train_step's fused finalize+Adam already wrote the new row scales
    hipLaunchKernelGGL(wn_scale_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
    LAUNCH_OK("wn_scale_kernel");
  }
  if (ride != nullptr) a.adam = *ride;
  hipLaunchKernelGGL(wn_tiles_kernel, dim3(tiles + a.adam.blocks), dim3(256), 0, st, a);
  LAUNCH_OK("wn_tiles_kernel");
  return 0;
}

constexpr float LATENT_DROPOUT_P = 0.2f;          // nn.Dropout(0.2), deep_sdf_decoder.py:36
constexpr int LATENT_DROPOUT_KEY = DSDF_MAX_LAYERS - 1;   // slot of dropout_key[] (no hidden layer can have this index)
inline uint32_t latent_drop_thr() { return (uint32_t)lround((double)LATENT_DROPOUT_P * 65536.0); }
inline bool net_variant(const DsdfNet* n) { return n->latent_dropout || n->xyz_in_all || n->ln_param_mask != 0; }
inline bool ln_applied(const DsdfNet* n, int l) { return ((n->ln_param_mask >> l) & 1) && l < n->n_layers - 1; }   // hidden layers only


// x0 (+ skip copies, + the xyz columns of every layer of an xyz_in_all net) from either the latent table + segments or an
// explicit input; keys != nullptr && training: latent_dropout nets drop layer 0's latent columns on the way
int run_gather(const DsdfNet* net, const Plan& P, void* ws, const float* table, const DsdfBatch* b, const float* input,
               int64_t ld_in, int64_t n, hipStream_t st, int training = 0, const uint32_t* keys = nullptr, uint32_t row_offset = 0) {
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.table = table; g.L = net->latent_size; g.G = net->geom_dim;
  if (b) { g.xyz = b->xyz; g.seg_scene = b->seg_scene; g.seg_offset = b->seg_offset; g.R = (int)b->n_segments; }
  g.input = input; g.ld_in = ld_in; g.n = (int)n;
  const int W0 = net->latent_size + net->geom_dim;
  const bool drop = net->latent_dropout && training && keys != nullptr && net->latent_size > 0;
  if (drop) { g.drop_key = keys[LATENT_DROPOUT_KEY]; g.drop_thr = latent_drop_thr(); g.drop_scale = 1.0f / (1.0f - LATENT_DROPOUT_P); g.row_offset = row_offset; }
  g.ndst = 0;
  g.dst[g.ndst++] = GatherDst{at<float>(ws, P.in_off[0]), P.ld_in[0], 0, 0, W0, drop ? 1 : 0};
  for (int l = 1; l < net->n_layers; ++l) {
    if ((net->skip_mask >> l) & 1) g.dst[g.ndst++] = GatherDst{at<float>(ws, P.in_off[l]), P.ld_in[l], net->out_dim[l - 1], 0, W0, 0};
    else if (net->xyz_in_all) g.dst[g.ndst++] = GatherDst{at<float>(ws, P.in_off[l]), P.ld_in[l], net->out_dim[l - 1], net->latent_size, net->geom_dim, 0};
  }
  hipLaunchKernelGGL(gather_concat_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, g);
  LAUNCH_OK("gather_concat_kernel");
  return 0;
}

bool fused_enabled() {
  const char* e = getenv("DSDF_NO_FUSED");
  return !(e && e[0] == '1');
}

bool fused_eligible(const DsdfNet* net) {
  if (net_variant(net)) return false;   // latent_dropout / xyz_in_all live on the layer-by-layer kernels only
  if (net->in_dim[0] > FMAXW) return false;
  for (int l = 0; l < net->n_layers - 1; ++l)
    if (net->in_dim[l] > FMAXW || net->out_dim[l] > FMAXW) return false;
  return net->in_dim[net->n_layers - 1] <= FMAXW;
}

// Narrow nets (every layer input and hidden width <= 128): the fused kernels' n128 variants, two workgroups per CU (fused.hpp).
// fp32 MFMA only; DSDF_NO_NARROW=1 switches it off (A/B).
bool net_narrow(const DsdfNet* net) {
  const char* e = getenv("DSDF_NO_NARROW");
  if ((e && e[0] == '1') || !fused_enabled() || !fused_eligible(net) || net->gemm_split || net->fwd_bf16) return false;
  for (int l = 0; l < net->n_layers; ++l)
    if (net->in_dim[l] > 128 || (l < net->n_layers - 1 && net->out_dim[l] > 128)) return false;
  return true;
}

// Rows per workgroup of the fused kernels: 64, or 32 (fused_*_h32_kernel: fp32 MFMA, merged forward + backward or forward alone) when
// 32-row workgroups still fit one per CU -- i.e. when 64-row workgroups would leave at least half of the chip idle (BASELINE config 4:
// one shape x 8000 points).  DSDF_FROWS=64 switches it off (A/B).
// merged: the training step's one-launch forward + backward.  Nets of at most 32-wide layers take it WAVE-PRIVATE at every batch size
// (32 points per one-wave workgroup, fused_fwd_bwd_w32_kernel; DSDF_NO_W32=1: the 64-row narrow kernels instead).
int w32_width(const DsdfNet* net) {      // 0: no; 32 / 64: the wave-private kernel for nets of at most that width
  if (!net_narrow(net) || getenv("DSDF_NO_W32")) return 0;
  int wmax = 0;
  for (int l = 0; l < net->n_layers; ++l) {
    wmax = std::max(wmax, net->in_dim[l]);
    if (l < net->n_layers - 1) wmax = std::max(wmax, net->out_dim[l]);
  }
  if (wmax <= FWW) return FWW;
  if (wmax <= FWW2 && !getenv("DSDF_NO_W32X2")) return FWW2;
  return 0;
}
bool w32_wanted(const DsdfNet* net) { return w32_width(net) != 0; }
int pick_frows(const DsdfNet* net, int64_t n, bool merged = false) {
  const char* e = getenv("DSDF_FROWS");      // read per call: the tests switch it inside one process
  const bool off = e && !strcmp(e, "64");
  if (off || !fused_enabled() || !fused_eligible(net) || net->gemm_split || net->fwd_bf16) return FROWS;
  if (merged && n > 0 && w32_wanted(net)) return 32;
  return n > 0 && n <= 32ll * (chip_waves() / 4) ? 32 : FROWS;
}

// all hidden layers + the last layer's forward in ONE launch (fused.hpp).  store_act: keep global copies of the
// activations (training / module path) or not (inference).
// Segment mode: U[s][t][:] = W_t[:, latent columns] latent_s for layer 0 (t = 0) and the skip layer (t = 1), and the
// descriptor the fused forward needs to start its accumulators from them.
// renorm != nullptr (training steps): the launch also does the max-norm renorm of the looked-up rows into zr (max_norm <= 0: a plain
// copy) and zeroes the dense latent gradient (nzero floats of dlat; 0: leaves it) -- no latent_renorm_kernel launch in segment mode
struct HoistRenorm { float max_norm; float* dlat; long long nzero; };
int run_hoist(const DsdfNet* net, const Plan& P, void* ws, const float* packed, const float* table, const DsdfBatch* b,
              FusedSeg* seg, hipStream_t st, const HoistRenorm* renorm = nullptr) {
  const Packed pk = packed_layout(net);
  const int ks = skip_layer(net);
  HoistArgs h;
  memset(&h, 0, sizeof(h));
  memset(seg, 0, sizeof(*seg));
  h.nh = ks > 0 ? 2 : 1;
  h.W[0] = packed + pk.w_off[0]; h.ldw[0] = pk.ldw[0]; h.c0[0] = 0; h.out[0] = net->out_dim[0];
  seg->h[0].layer = 0; seg->h[0].wx = h.W[0] + net->latent_size; seg->h[0].ldw = pk.ldw[0];
  seg->h[1].layer = -1;
  if (ks > 0) {
    h.W[1] = packed + pk.w_off[ks]; h.ldw[1] = pk.ldw[ks]; h.c0[1] = net->out_dim[ks - 1]; h.out[1] = net->out_dim[ks];
    seg->h[1].layer = ks; seg->h[1].wx = h.W[1] + h.c0[1] + net->latent_size; seg->h[1].ldw = pk.ldw[ks];
  }
  h.L = net->latent_size; h.seg_scene = b->seg_scene; h.table = table; h.R = (int)b->n_segments;
  h.U = at<float>(ws, P.hoistU_off); h.ldu = P.ldu;
  h.bf16 = net->fwd_bf16 ? 1 : 0;
  if (renorm != nullptr) {
    h.max_norm = renorm->max_norm; h.zr = at<float>(ws, P.zr_off);
    h.dlat = renorm->nzero > 0 ? renorm->dlat : nullptr; h.nzero = renorm->nzero;
  }
  const int rows = h.out[0] + (ks > 0 ? h.out[1] : 0);
  const dim3 hgrid((unsigned)((rows + 3) / 4), (unsigned)((h.R + HOIST_SC - 1) / HOIST_SC));
  switch ((h.L + 63) >> 6) {               // k-units of 64 latent columns per lane (HOIST_MAXL = 512)
    case 1: hipLaunchKernelGGL(seg_hoist_kernel<1>, hgrid, dim3(256), 0, st, h); break;
    case 2: hipLaunchKernelGGL(seg_hoist_kernel<2>, hgrid, dim3(256), 0, st, h); break;
    case 3: case 4: hipLaunchKernelGGL(seg_hoist_kernel<4>, hgrid, dim3(256), 0, st, h); break;
    default: hipLaunchKernelGGL(seg_hoist_kernel<8>, hgrid, dim3(256), 0, st, h); break;
  }
  LAUNCH_OK("seg_hoist_kernel");
  seg->wg_per_seg = (int)(b->seg_len / P.frows);
  seg->xyz = b->xyz; seg->G = net->geom_dim; seg->U = h.U; seg->ldu = h.ldu;
  return 0;
}

// seg != nullptr: segment mode (fused.hpp FusedSeg) -- x0 is not read at all, seg->h[] / seg->U come from run_hoist
int run_fused_forward(const DsdfNet* net, const Plan& P, void* ws, const float* packed, const float* params, int64_t n,
                      int training, const uint32_t* keys, uint32_t row_offset, bool store_act, float* y_out, float* u_out,
                      hipStream_t st, const FusedSeg* seg = nullptr, FusedFwdArgs* defer = nullptr) {
  // defer != nullptr: fill *defer and launch nothing -- the caller hands it to run_backward_fused, which launches forward and
  // backward as ONE kernel (fused_fwd_bwd_kernel)
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int last = net->n_layers - 1;
  FusedFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.n_hidden = last; a.N = (int)n; a.W0 = net->in_dim[0];
  a.x0 = at<float>(ws, P.in_off[0]); a.ldx0 = P.ld_in[0];
  a.row_offset = row_offset;
  for (int l = 0; l < last; ++l) {
    FusedLayer& y = a.ly[l];
    y.wf = net->fwd_bf16 ? packed + pk.wfb_off[l] : (net->gemm_split ? packed + pk.ws_off[l] : packed + pk.wf_off[l]);   // (bf16 copy / split planes)
    y.wplane = (int)(pk.ws_plane[l] * 2);   // bytes per plane (gemm_split)
    y.wf32 = packed + pk.wf_off[l];
    y.bias = params + L.bias_off[l];
    y.out = store_act ? at<float>(ws, P.in_off[l + 1]) : nullptr;
    y.ld_out = P.ld_in[l + 1];
    y.in = net->in_dim[l]; y.out_dim = net->out_dim[l]; y.U = pk.uf[l];
    const bool drop = training && ((net->dropout_mask >> l) & 1) && net->dropout_p > 0.f;
    if (drop) {
      long thr = lround((double)net->dropout_p * 65536.0);
      if (thr > 65535) thr = 65535;
      y.drop_thr = (uint32_t)thr; y.drop_key = keys[l]; y.drop_scale = 1.0f / (1.0f - net->dropout_p);
    }
    y.x0_col = ((net->skip_mask >> (l + 1)) & 1) ? net->out_dim[l] : -1;
    y.maskbits = store_act ? at<uint32_t>(ws, P.mask_off[l]) : nullptr;
    if (seg != nullptr) {   // hoisted x0 columns: nothing left to contract for layer 0, only the previous layer for the skip layer
      y.x0_col = -1;
      if (l == 0) y.in = 0;
      else if ((net->skip_mask >> l) & 1) y.in = net->out_dim[l - 1];
    }
  }
  if (seg != nullptr) a.seg = *seg;
  a.w_last = packed + pk.w_off[last]; a.b_last = params + L.bias_off[last]; a.in_last = net->in_dim[last];
  a.use_tanh = net->use_tanh; a.y_out = y_out; a.u_out = u_out;
  if (defer != nullptr) {
    a.ly[last - 1].out = nullptr;   // the last hidden activation is consumed from the slab by the backward head: no global copy
    *defer = a;
    return 0;
  }
  double wmac = 0;
  for (int l = 0; l < last; ++l) wmac += (double)net->in_dim[l] * net->out_dim[l];
  ProfScope ps(DSDF_PROF_FUSED_FWD, 2.0 * (double)n * wmac, st);
  #ifdef DSDF_LAB
  static unsigned long long* dbg = nullptr;
  if (!dbg && getenv("DSDF_LAB_DBG")) { (void)hipMalloc(&dbg, 8192 * 64 * 8); }
  a.dbg = dbg;
#endif
  const dim3 grid((unsigned)((n + P.frows - 1) / P.frows));
  if (P.frows == 32)                 // small batch: 32 points per workgroup (pick_frows: fp32 MFMA only)
    hipLaunchKernelGGL(fused_forward_h32_kernel, grid, dim3(256), 0, st, a);
  else if (net_narrow(net))          // every layer <= 128 wide: two workgroups per CU
    hipLaunchKernelGGL(fused_forward_n128_kernel, grid, dim3(256), 0, st, a);
  else if (net->fwd_bf16 && !store_act)   // config 5, inference form: 8 staggered waves, transposed accumulators (fused_bf16x8.hpp)
    hipLaunchKernelGGL(fused_forward_bf16x8_kernel, grid, dim3(F8_THREADS), 0, st, a);
  else if (net->fwd_bf16)            // with activation copies (module path; training goes out merged with the backward)
    hipLaunchKernelGGL(fused_forward_bf16_kernel, grid, dim3(256), 0, st, a);
  else
    if (net->gemm_split) hipLaunchKernelGGL(fused_forward_split_kernel, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(fused_forward_kernel, grid, dim3(256), 0, st, a);
#ifdef DSDF_LAB
  if (dbg && getenv("DSDF_LAB_DBG")) {
    (void)hipDeviceSynchronize();
    static unsigned long long h[8192 * 64];
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    FILE* f = fopen(getenv("DSDF_LAB_DBG"), "wb");
    if (f) { fwrite(h, 1, sizeof(h), f); fclose(f); }
  }
#endif
  LAUNCH_OK("fused_forward_kernel");
  return 0;
}

// hidden layers 0..nl-2: in[l+1][:, :out_l] = dropout(relu(in[l] W_l^T + b_l)); LayerNorm layers: Linear (bias) into the xhat
// buffer, then ln_fwd_kernel normalises in place (kept for the backward when save_ln) and writes the activated output
int run_hidden_forward(const DsdfNet* net, const Plan& P, void* ws, const float* packed, const float* params, int64_t n,
                       int training, const uint32_t* keys, uint32_t row_offset, hipStream_t st, bool save_ln = true) {
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  for (int l = 0; l < net->n_layers - 1; ++l) {
    NtArgs a;
    memset(&a, 0, sizeof(a));
    a.A = at<float>(ws, P.in_off[l]); a.lda = P.ld_in[l];
    a.B = packed + pk.w_off[l]; a.ldb = pk.ldw[l];
    a.C = at<float>(ws, P.in_off[l + 1]); a.ldc = P.ld_in[l + 1];
    a.M = (int)n; a.N = net->out_dim[l]; a.K = net->in_dim[l];
    a.bias = params + L.bias_off[l];
    a.relu = 1;
    const bool drop = training && ((net->dropout_mask >> l) & 1) && net->dropout_p > 0.f;
    uint32_t thr = 0;
    if (drop) {
      long t = lround((double)net->dropout_p * 65536.0);
      if (t > 65535) t = 65535;
      thr = (uint32_t)t;
    }
    if (ln_applied(net, l)) {
      a.C = at<float>(ws, P.lnx_off[l]);
      TRY(launch_nt<EPI_PLAIN>(a, st));
      LnFwdArgs f;
      memset(&f, 0, sizeof(f));
      f.y = a.C; f.ldy = a.ldc; f.gamma = params + L.ln_w_off[l]; f.beta = params + L.ln_b_off[l];
      f.out = at<float>(ws, P.in_off[l + 1]); f.ldo = P.ld_in[l + 1]; f.n = (int)n; f.width = net->out_dim[l];
      f.save = save_ln ? 1 : 0; f.rstd = save_ln ? at<float>(ws, P.lnr_off[l]) : nullptr;
      if (drop) { f.drop_thr = thr; f.drop_key = keys[l]; f.drop_scale = 1.0f / (1.0f - net->dropout_p); f.row_offset = row_offset; }
      hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, f);
      LAUNCH_OK("ln_fwd_kernel");
      continue;
    }
    if (drop) {
      a.drop_thr = thr;
      a.drop_key = keys[l];
      a.drop_scale = 1.0f / (1.0f - net->dropout_p);
      a.row_offset = row_offset;
    }
    TRY(launch_nt<EPI_FWD>(a, st));
  }
  return 0;
}

float mask_scale_of(const DsdfNet* net, int layer, int training) {
  const bool drop = training && ((net->dropout_mask >> layer) & 1) && net->dropout_p > 0.f;
  return drop ? 1.0f / (1.0f - net->dropout_p) : 1.0f;
}

// shared backward over hidden layers, given dp of layer nl-2 in dp[0] and the last layer's partials.
// ncols_dz: how many leading x0 columns of d/dx0 are needed (L for training, W0 for the module path).
// keys / row_offset: latent_dropout nets mask the latent part of layer 0's dX (dzA) with the forward's hash.
// xyz_acc != nullptr (xyz_in_all, module path with d/d(input)): *xyz_acc = true when dxz_off[1] holds the sum of the xyz_in
// layers' d/d(xyz) (the caller adds it to the xyz columns of d_input).
int run_backward(const DsdfNet* net, const Plan& P, void* ws, const float* packed, const float* params, int64_t n,
                 int training, float* grads, int accumulate, int ncols_dz, bool* used_dzB, hipStream_t st, bool want_dw = true,
                 const uint32_t* keys = nullptr, uint32_t row_offset = 0, bool* xyz_acc = nullptr) {
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int nl = net->n_layers;
  const int last = nl - 1;
  // second stage of the last layer's partials
  if (want_dw) {
    const int w = P.ld_part;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((w + 63) / 64, LAST_GROUPS), dim3(256), 0, st,
                       ReduceRowsArgs{at<float>(ws, P.part_off), P.last_blocks, P.ld_part, w, at<float>(ws, P.part2_off), LAST_GROUPS});
    LAUNCH_OK("reduce_rows_kernel");
    FinArgs f;
    memset(&f, 0, sizeof(f));
    f.slabs = at<float>(ws, P.part2_off); f.nsplit = LAST_GROUPS; f.slab = P.ld_part; f.ldc = P.ld_part;
    f.colsum = at<float>(ws, P.partdb_off); f.npart = P.last_blocks; f.ldcs = 1;
    f.g = L.g_off[last] >= 0 ? params + L.g_off[last] : nullptr;
    f.v = params + L.v_off[last];
    f.dg = L.g_off[last] >= 0 ? grads + L.g_off[last] : nullptr;
    f.dv = grads + L.v_off[last];
    f.db = grads + L.bias_off[last];
    f.out = 1; f.in = net->in_dim[last]; f.accumulate = accumulate;
    hipLaunchKernelGGL(finalize_layer_kernel, dim3(1), dim3(256), 0, st, f);
    LAUNCH_OK("finalize_layer_kernel(last)");
    if ((net->ln_param_mask >> last) & 1) {   // bn module of the last Linear: created by the reference, never called: zero gradient
      LnGradArgs z;
      memset(&z, 0, sizeof(z));
      z.dgamma = grads + L.ln_w_off[last]; z.dbeta = grads + L.ln_b_off[last]; z.width = net->out_dim[last]; z.accumulate = accumulate; z.zero = 1;
      hipLaunchKernelGGL(ln_param_grad_kernel, dim3(1), dim3(64), 0, st, z);
      LAUNCH_OK("ln_param_grad_kernel(last)");
    }
  }
  *used_dzB = false;
  int cur = 0;
  for (int l = last - 1; l >= 0; --l) {
    float* dp = at<float>(ws, P.dp_off[cur]);
    // column-sum partials of dp as it arrives (from the output layer's kernel or the next layer's dX epilogue)
    const float* cs_ptr = l == last - 1 ? at<float>(ws, P.part2_off) + P.ld_in[last] : at<float>(ws, P.colsum_off);
    int cs_n = l == last - 1 ? LAST_GROUPS : P.mt, cs_ld = l == last - 1 ? P.ld_part : P.ldcs;
    const bool ln = ln_applied(net, l);
    if (ln) {   // dp is d/dz (after LayerNorm): turn it into d/d(Linear output) in place; gamma / beta / bias partials
      LnBwdArgs b;
      memset(&b, 0, sizeof(b));
      b.dz = dp; b.ldz = P.ld_dp; b.xhat = at<float>(ws, P.lnx_off[l]); b.ldx = P.ld_in[l + 1]; b.rstd = at<float>(ws, P.lnr_off[l]);
      b.gamma = params + L.ln_w_off[l]; b.n = (int)n; b.width = net->out_dim[l];
      b.part_dgamma = at<float>(ws, P.lnpg_off); b.part_db = at<float>(ws, P.lnpb_off); b.ldp = P.ldcs;
      int blocks = (int)((n + 3) / 4);
      if (blocks > LN_BLOCKS) blocks = LN_BLOCKS;
      hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, st, b);
      LAUNCH_OK("ln_bwd_kernel");
      if (want_dw) {
        LnGradArgs g;
        memset(&g, 0, sizeof(g));
        g.part_dgamma = b.part_dgamma; g.nblk = blocks; g.ldp = b.ldp; g.cs = cs_ptr; g.ncs = cs_n; g.ldcs = cs_ld;
        g.dgamma = grads + L.ln_w_off[l]; g.dbeta = grads + L.ln_b_off[l]; g.width = net->out_dim[l]; g.accumulate = accumulate;
        hipLaunchKernelGGL(ln_param_grad_kernel, dim3((g.width + 255) / 256), dim3(256), 0, st, g);
        LAUNCH_OK("ln_param_grad_kernel");
      }
      cs_ptr = b.part_db; cs_n = blocks; cs_ld = b.ldp;     // the Linear's bias gradient = column sums of dy
    }
    // dW_l = dp^T in_l  (split-K slabs)
    if (want_dw) {
    TnArgs t;
    memset(&t, 0, sizeof(t));
    t.A = dp; t.lda = P.ld_dp; t.B = at<float>(ws, P.in_off[l]); t.ldb = P.ld_in[l];
    t.C = at<float>(ws, P.slab_off); t.ldc = P.ld_in[l]; t.M = net->out_dim[l]; t.N = net->in_dim[l]; t.K = (int)n;
    t.kchunk = P.kchunk; t.slab = P.slab;
    TRY(launch_tn(t, P.nsplit, st));
    FinArgs f;
    memset(&f, 0, sizeof(f));
    f.slabs = t.C; f.nsplit = P.nsplit; f.slab = P.slab; f.ldc = t.ldc;
    f.colsum = cs_ptr; f.npart = cs_n; f.ldcs = cs_ld;
    f.g = L.g_off[l] >= 0 ? params + L.g_off[l] : nullptr;
    f.v = params + L.v_off[l];
    f.dg = L.g_off[l] >= 0 ? grads + L.g_off[l] : nullptr;
    f.dv = grads + L.v_off[l];
    f.db = grads + L.bias_off[l];
    f.out = net->out_dim[l]; f.in = net->in_dim[l]; f.accumulate = accumulate;
    hipLaunchKernelGGL(finalize_layer_kernel, dim3(f.out), dim3(256), 0, st, f);
    LAUNCH_OK("finalize_layer_kernel");
    }
    // dX
    NtArgs a;
    memset(&a, 0, sizeof(a));
    a.A = dp; a.lda = P.ld_dp; a.B = packed + pk.wt_off[l]; a.ldb = pk.ldwt[l];
    a.M = (int)n; a.K = net->out_dim[l];
    if (l > 0) {
      const bool skip = (net->skip_mask >> l) & 1;
      a.C = at<float>(ws, P.dp_off[cur ^ 1]); a.ldc = P.ld_dp;
      a.act = at<float>(ws, P.in_off[l]); a.ldact = P.ld_in[l];
      a.mask_cols = net->out_dim[l - 1];
      a.mask_scale = mask_scale_of(net, l - 1, training);
      const bool xyz_l = !skip && net->xyz_in_all;                 // this layer's input is [a || xyz]
      const bool want_xyz = xyz_l && xyz_acc != nullptr && ncols_dz > net->latent_size;
      a.N = skip ? (ncols_dz > 0 ? a.mask_cols + ncols_dz : a.mask_cols) : (xyz_l ? a.mask_cols + (want_xyz ? net->geom_dim : 0) : net->in_dim[l]);
      if (skip && ncols_dz > 0) { a.C2 = at<float>(ws, P.dzB_off); a.ldc2 = P.ldz; a.c2_cols = ncols_dz; *used_dzB = true; }
      if (want_xyz) { a.C2 = at<float>(ws, P.dxz_off[0]); a.ldc2 = 4; a.c2_cols = net->geom_dim; }
      a.colsum = at<float>(ws, P.colsum_off); a.ldcs = P.ldcs;
      TRY(launch_nt<EPI_BWD>(a, st));
      if (want_xyz) {
        const long long tot = (long long)n * net->geom_dim;
        hipLaunchKernelGGL(acc_cols_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, at<float>(ws, P.dxz_off[0]), 4,
                           at<float>(ws, P.dxz_off[1]), 4, 0, (int)n, net->geom_dim, *xyz_acc ? 1 : 0);
        LAUNCH_OK("acc_cols_kernel");
        *xyz_acc = true;
      }
      cur ^= 1;
    } else if (ncols_dz > 0) {
      a.C = at<float>(ws, P.dzA_off); a.ldc = P.ldz; a.N = ncols_dz;
      TRY(launch_nt<EPI_PLAIN>(a, st));
      if (net->latent_dropout && training && keys != nullptr && net->latent_size > 0) {   // layer 0 saw the DROPPED latent
        const long long tot = (long long)n * net->latent_size;
        hipLaunchKernelGGL(latent_drop_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, at<float>(ws, P.dzA_off), P.ldz,
                           (int)n, net->latent_size, keys[LATENT_DROPOUT_KEY], latent_drop_thr(), 1.0f / (1.0f - LATENT_DROPOUT_P), row_offset);
        LAUNCH_OK("latent_drop_bwd_kernel");
      }
    }
  }
  return 0;
}

struct FuseAdam { const DsdfAdamCfg* cfg; float* params; float* exp_avg; float* exp_avg_sq; float* packed; };

// Backward with the fused dX chain (fused.hpp): K3's second stage + last layer finalize, ONE launch for the whole
// dX chain (writes every dP_l, column sums, latent-gradient inputs), then dW (split-K) + finalize per layer.
// Segment mode (sb != nullptr): what the weight gradients of the hoisted layers need from the batch
struct SegBwd { const FusedSeg* seg; const int64_t* seg_scene; const float* table; int R;
                ScatterArgs* scatter; bool* scatter_done;       // (scatter->nslice is set here: the latent role decides it)
                const float* zr; };   // the segments' renormed latent rows (run_hoist), or nullptr: read table[seg_scene[r]]   // the dense latent-gradient scatter may ride on the finalize launch

int run_backward_fused(const DsdfNet* net, const Plan& P, void* ws, const float* packed, const float* params, int64_t n,
                       int training, float* grads, int accumulate, int ncols_dz, bool* used_dzB, hipStream_t st,
                       bool want_dw, const FusedBwdHead& head, const FuseAdam* fz = nullptr, const SegBwd* sb = nullptr,
                       const FusedFwdArgs* fwd = nullptr,     // fwd: the deferred forward of the same points -> one launch for both
                       int phase = 0) {                       // DsdfLossCfg.dw_phase: 0 = everything; p >= 1: dW + finalize of bucket p - 1
                                                              // only (p = 1: after the forward + backward launch and its roles)
  const DwSched& DS = phase == 0 ? P.dw : P.dwph[phase - 1];
  auto in_phase = [&](int l) { return phase == 0 || (l >= P.dw_cut[phase] && l < P.dw_cut[phase - 1]); };
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int nl = net->n_layers, last = nl - 1;
  *used_dzB = false;
  FusedBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.N = (int)n;
  a.head = head;
  const bool segmode = sb != nullptr;
  const int ks = skip_layer(net);
  if (segmode) { a.xyz = sb->seg->xyz; a.G = sb->seg->G; }
  int cnt = 0;
  double wmac = 0;
  for (int l = last - 1; l >= 0; --l) {
    if (l == 0 && ncols_dz <= 0) break;
    FusedBwdLayer& y = a.ly[cnt++];
    y.wtf = net->gemm_split ? packed + pk.wts_off[l] : packed + pk.wtf_off[l]; y.U = pk.utf[l]; y.K = net->out_dim[l];
    y.wplane = (int)(pk.wts_plane[l] * 2);
    y.wtf32 = packed + pk.wtf_off[l];
    if (l > 0) {
      const bool skip = (net->skip_mask >> l) & 1;
      y.mask_cols = net->out_dim[l - 1];
      y.mask_scale = mask_scale_of(net, l - 1, training);
      y.maskbits = at<uint32_t>(ws, P.mask_off[l - 1]);
      y.dp_out = at<float>(ws, P.dpl_off[l - 1]); y.ld_dp = P.ld_dp;
      y.colsum = at<float>(ws, P.cs_off[l - 1]); y.ldcs = P.ldcs;
      if (segmode && want_dw && (l - 1 == 0 || l - 1 == ks)) y.xsum = at<float>(ws, P.xsum_off[l - 1 == 0 ? 0 : 1]);
      if (segmode && l - 1 == 0) y.dp_out = nullptr;   // dP_0 is consumed through its column sums only
      if (skip && ncols_dz > 0) { y.dz_out = at<float>(ws, P.dzB_off); y.ldz = P.ldz; y.dz_cols = ncols_dz; *used_dzB = true; }
      y.ncols = y.mask_cols + y.dz_cols;
    } else {
      y.mask_cols = 0; y.mask_scale = 1.f;
      y.dz_out = at<float>(ws, P.dzA_off); y.ldz = P.ldz; y.dz_cols = ncols_dz; y.ncols = ncols_dz;
    }
    wmac += (double)y.K * y.ncols;
  }
  a.n_layers = cnt;
  if (last_layer_skip(net)) {   // the head hands the x0 columns' gradient du w[x0 cols] to the skip-layer buffer of d/dx0
    a.head.dz_out = ncols_dz > 0 ? at<float>(ws, P.dzB_off) : nullptr; a.head.ldz = P.ldz; a.head.dz_cols = ncols_dz;
    if (ncols_dz > 0) *used_dzB = true;
  }
  if (phase <= 1) {
    // algorithmic FLOPs of the dX chain (the reference back-propagates through every hidden layer down to x0); the
    // executed count `wmac` is smaller: layer 0's dX and the skip layer's x0 columns come from column sums instead
    double amac = 0;
    for (int l = 0; l < last; ++l) amac += (double)net->in_dim[l] * net->out_dim[l];
    (void)wmac;
    if (fwd != nullptr) {
#ifdef DSDF_LAB
      // lab: stamps of the MERGED launch (forward slots 0.., backward slots 32..), dumped after every launch
      static unsigned long long* mdbg = nullptr;
      FusedFwdArgs fwd_l = *fwd;
      if (!mdbg && getenv("DSDF_LAB_MDBG")) { (void)hipMalloc(&mdbg, 8192 * 64 * 8); }
      if (mdbg) { (void)hipMemsetAsync(mdbg, 0, 8192 * 64 * 8, st); fwd_l.dbg = mdbg; a.dbg = mdbg; fwd = &fwd_l; }
#endif
      ProfScope ps(DSDF_PROF_FUSED_FWD_BWD, 4.0 * (double)n * amac, st);   // forward + dX chain
      if (net->fwd_bf16 && net->gemm_split) hipLaunchKernelGGL(fused_fwd_bf16_bwd_split_kernel, dim3((unsigned)P.nwg), dim3(256), 0, st, *fwd, a);
      else if (net->fwd_bf16) hipLaunchKernelGGL(fused_fwd_bf16_bwd_kernel, dim3((unsigned)P.nwg), dim3(256), 0, st, *fwd, a);
      else if (net->gemm_split) hipLaunchKernelGGL(fused_fwd_bwd_split_kernel, dim3((unsigned)P.nwg), dim3(256), 0, st, *fwd, a);
      else if (P.frows == 32 && w32_width(net) == FWW) hipLaunchKernelGGL(fused_fwd_bwd_w32_kernel, dim3((unsigned)P.nwg), dim3(64), 0, st, *fwd, a);
      else if (P.frows == 32 && w32_width(net) == FWW2) hipLaunchKernelGGL(fused_fwd_bwd_w32x2_kernel, dim3((unsigned)P.nwg), dim3(64), 0, st, *fwd, a);
      else if (P.frows == 32) hipLaunchKernelGGL(fused_fwd_bwd_h32_kernel, dim3((unsigned)P.nwg), dim3(256), 0, st, *fwd, a);
      else if (net_narrow(net)) hipLaunchKernelGGL(fused_fwd_bwd_n128_kernel, dim3((unsigned)P.nwg), dim3(256), 0, st, *fwd, a);
      else hipLaunchKernelGGL(fused_fwd_bwd_kernel, dim3((unsigned)P.nwg), dim3(256), 0, st, *fwd, a);
      LAUNCH_OK("fused_fwd_bwd_kernel");
#ifdef DSDF_LAB
      if (mdbg && getenv("DSDF_LAB_MDBG")) {
        (void)hipDeviceSynchronize();
        static unsigned long long h[8192 * 64];
        (void)hipMemcpy(h, mdbg, sizeof(h), hipMemcpyDeviceToHost);
        FILE* f = fopen(getenv("DSDF_LAB_MDBG"), "wb");
        if (f) { fwrite(h, 1, sizeof(h), f); fclose(f); }
      }
#endif
    } else {
      ProfScope ps(DSDF_PROF_FUSED_BWD, 2.0 * (double)n * amac, st);
      if (P.frows != FROWS) return fail(DSDF_E_LAUNCH, "internal: the separate backward kernel has 64-row workgroups only");
      if (net->gemm_split) hipLaunchKernelGGL(fused_backward_split_kernel, dim3((unsigned)P.nwg), dim3(256), 0, st, a);
      else hipLaunchKernelGGL(fused_backward_kernel, dim3((unsigned)P.nwg), dim3(256), 0, st, a);
      LAUNCH_OK("fused_backward_kernel");
    }
  }
  const ReduceRowsArgs rr{at<float>(ws, P.part_off), P.nwg, P.ld_part, P.ld_part, at<float>(ws, P.part2_off), LAST_GROUPS};
  const int rr_bx = (P.ld_part + 63) / 64;
  if (!segmode && phase <= 1) {
    if (want_dw) {   // second stage of the head's per-workgroup partials
      hipLaunchKernelGGL(reduce_rows_kernel, dim3(rr_bx, LAST_GROUPS), dim3(256), 0, st, rr);
      LAUNCH_OK("reduce_rows_kernel");
    }
  }
  // segment mode: everything that consumes only the backward's per-workgroup partials (kernels.hpp post_bwd_role) -- run by
  // the workgroups the dW launch leaves idle, or by a launch of its own when there is no dW launch / no idle workgroup
  PostBwdArgs q;
  memset(&q, 0, sizeof(q));
  int lat_n = 0, lat_slices = 1;
  if (segmode) {
    if (want_dw) {
      q.rr = rr; q.rr_bx = rr_bx; q.rr_n = rr_bx * LAST_GROUPS;
      SegDwArgs& d = q.dw;
      d.nh = ks > 0 ? 2 : 1;
      d.cs[0] = at<float>(ws, P.cs_off[0]); d.xsum[0] = at<float>(ws, P.xsum_off[0]); d.out[0] = net->out_dim[0];
      if (ks > 0) { d.cs[1] = at<float>(ws, P.cs_off[ks]); d.xsum[1] = at<float>(ws, P.xsum_off[1]); d.out[1] = net->out_dim[ks]; }
      d.ldcs = P.ldcs; d.nwg = P.nwg; d.wg_per_seg = sb->seg->wg_per_seg; d.R = sb->R; d.L = net->latent_size; d.G = net->geom_dim;
      d.seg_scene = sb->seg_scene; d.table = sb->table; d.zr = sb->zr;
      d.HS = at<float>(ws, P.hs_off); d.ldh = P.ldh; d.hstride = P.hstride;
      // (block counts of the launch-of-its-own form; the riding form below has its own)
      q.dw_n = (d.out[0] + SDW_ROWS_WIDE - 1) / SDW_ROWS_WIDE + (ks > 0 ? (d.out[1] + SDW_ROWS_WIDE - 1) / SDW_ROWS_WIDE : 0);
      if (seg_dw_long_form(d.wg_per_seg, d.out[0], d.out[1])) q.dw_n = d.out[0] + d.out[1];      // long segments, narrow layers: one row per block (seg_dw_row_body)
    }
    SegLatArgs& g = q.lat;   // per-segment latent gradient from the column sums of dP_0 / dP_skip
    g.cs0 = at<float>(ws, P.cs_off[0]); g.ldcs = P.ldcs; g.out0 = net->out_dim[0];
    g.W0 = packed + pk.w_off[0]; g.ldw0 = pk.ldw[0];
    if (ks > 0) {
      g.csk = at<float>(ws, P.cs_off[ks]); g.outk = net->out_dim[ks];
      g.Wk = packed + pk.w_off[ks]; g.ldwk = pk.ldw[ks]; g.koff = net->out_dim[ks - 1];
    }
    g.wg_per_seg = sb->seg->wg_per_seg; g.R = sb->R; g.L = net->latent_size;
    g.seg_scene = sb->seg_scene; g.table = sb->table; g.zr = sb->zr;
    g.segpart = at<float>(ws, P.segpart_off); g.segnorm = at<float>(ws, P.segnorm_off);
    q.lat_bx = sb->R;
    g.nchunk = (net->latent_size + 15) / 16;
    lat_n = sb->R * g.nchunk;
    // few long segments in a launch of their own (config 4: one shape): cut each segment's workgroups into slices so that the launch
    // has blocks for the chip; the scatter adds the slices
    g.nslice = 1; g.slice_stride = (long long)sb->R * net->latent_size;
    while (g.nslice < LAT_SLICES_MAX && lat_n * g.nslice * 2 <= chip_waves() / 4 && g.wg_per_seg / (g.nslice * 2) >= 16) g.nslice *= 2;
    lat_n *= g.nslice;
    lat_slices = g.nslice;
  }
  auto tell_scatter = [&]() { if (segmode && sb->scatter != nullptr) { sb->scatter->nslice = lat_slices; sb->scatter->slice_stride = (long long)sb->R * net->latent_size; } };
  const int cus = chip_waves() / 4;
  const int dw_items = DS.n_full + DS.n_narrow;
  const int dw_busy = want_dw ? ((dw_items + 3) / 4 < cus ? (dw_items + 3) / 4 : cus) : 0;
  // the idle workgroups take the role blocks one after the other: that stays inside the dW time for batches of up to one
  // workgroup per CU (measured: 16384 points, 1168 role blocks on 16 workgroups, dW time unchanged); larger batches put
  // the roles on the critical path (65536 points: -5 %), so they get their own (wide) launch there
  static const bool no_ride = [] { const char* e = getenv("DSDF_NO_RIDE"); return e && e[0] == '1'; }();   // A/B switch
  // Round 4: stamps inside the launch (profiles/r04_dw_stamps_before.log) showed the riding roles to be its TAIL -- 391 us of role
  // blocks on the 16 spare workgroups against 377 us of MFMA items -- and their latency-bound chains to be what made the launch
  // 388 us on one box and 403 us on the next.  The roles were rebuilt for few workgroups (seg_dw_body: 32 rows per block;
  // seg_latgrad_all_body: one block per 16 latent columns takes all segments): they now end 242 us into the launch.
  // (gemm_split: measured again with the rebuilt roles -- they end 242 us into the launch, the split items 174 us: riding there made
  // the launch 253 us instead of 174 + 18 for a launch of their own, so they still do not ride in split mode)
  const bool post_rides = segmode && want_dw && cus - dw_busy >= 8 && P.nwg <= cus && !no_ride && !net->gemm_split && phase <= 1 &&
                          dw_items > 0;
  if (post_rides) {   // the few-workgroups forms: 32 weight-gradient rows per block, one latent-gradient block per 16 columns
    q.lat.nslice = 1; lat_slices = 1;
    q.lat_bx = 0;
    lat_n = (net->latent_size + 15) / 16;
    q.rr_n = q.rr_bx;               // one block per 64-column strip of the head's partials takes all groups (reduce_rows_strip_body)
    q.dw_n = (q.dw.out[0] + SDW_ROWS_RIDE - 1) / SDW_ROWS_RIDE + (ks > 0 ? (q.dw.out[1] + SDW_ROWS_RIDE - 1) / SDW_ROWS_RIDE : 0);
  }
  tell_scatter();
  if (segmode && !post_rides && phase <= 1) {
    hipLaunchKernelGGL(post_bwd_kernel, dim3((unsigned)(q.rr_n + q.dw_n + lat_n)), dim3(256), 0, st, q, lat_n);
    LAUNCH_OK("post_bwd_kernel");
  }
  if (want_dw && dw_items > 0) {   // all dW_l = dP_l^T a_l (of this phase's layers) in one launch
    DwArgs d;
    memset(&d, 0, sizeof(d));
    d.n_layers = last; d.n_full = DS.n_full; d.n_narrow = DS.n_narrow; d.N = (int)n;
    double fl = 0;
    for (int l = 0; l < last; ++l) {
      DwLayer& y = d.ly[l];
      y.dp = at<float>(ws, P.dpl_off[l]); y.ld_dp = P.ld_dp;
      y.act = at<float>(ws, P.in_off[l]); y.ld_act = P.ld_in[l];
      y.slabs = at<float>(ws, P.dwslab_off[l]); y.slab = DS.slab[l];
      y.M = net->out_dim[l]; y.Nc = dw_cols(net, l, segmode); y.ldc = P.ld_in[l];
      y.tiles_m = DS.tiles_m[l]; y.tiles_n = DS.tiles_n[l]; y.last_nj = DS.last_nj[l]; y.nfull_n = DS.nfull_n[l];
      y.nsplit = DS.nsplit[l]; y.kchunk = DS.kchunk[l]; y.full0 = DS.full0[l]; y.narrow0 = DS.narrow0[l];
      if (in_phase(l)) fl += 2.0 * (double)n * y.M * net->in_dim[l];   // algorithmic (segment mode executes fewer: hoisted x0 columns)
    }
    PostBwdArgs none;
    memset(&none, 0, sizeof(none));
    int grid = dw_busy < 1 ? 1 : dw_busy;
#ifdef DSDF_LAB
    static unsigned long long* dwdbg = nullptr;      // lab: per-wave stamps of the LAST dW launch, dumped at every launch
    if (!dwdbg && getenv("DSDF_LAB_DWDBG")) { (void)hipMalloc(&dwdbg, 1024 * 4 * 8 * 8); }
    if (dwdbg) (void)hipMemsetAsync(dwdbg, 0, 1024 * 4 * 8 * 8, st);
    d.dbg = dwdbg;
#endif
    ProfScope ps(DSDF_PROF_DW_STREAM, fl, st);
    if (net->gemm_split) {
      if (post_rides) hipLaunchKernelGGL(dw_stream_split_kernel, dim3(cus), dim3(256), 0, st, d, q, lat_n, dw_busy);
      else hipLaunchKernelGGL(dw_stream_split_kernel, dim3(grid), dim3(256), 0, st, d, none, 0, grid);
    } else if (post_rides) hipLaunchKernelGGL(dw_stream_kernel, dim3(cus), dim3(256), 0, st, d, q, lat_n, dw_busy);
    else hipLaunchKernelGGL(dw_stream_kernel, dim3(grid), dim3(256), 0, st, d, none, 0, grid);
    LAUNCH_OK("dw_stream_kernel");
#ifdef DSDF_LAB
    if (dwdbg && getenv("DSDF_LAB_DWDBG")) {
      (void)hipDeviceSynchronize();
      static unsigned long long h[1024 * 4 * 8];
      (void)hipMemcpy(h, dwdbg, sizeof(h), hipMemcpyDeviceToHost);
      FILE* f = fopen(getenv("DSDF_LAB_DWDBG"), "wb");
      if (f) { fwrite(h, 1, sizeof(h), f); fclose(f); }
    }
#endif
  }
  if (want_dw) {   // split-K sums, weight-norm backward and bias gradients of ALL layers (last layer included) in one launch
    FinAll fa;
    memset(&fa, 0, sizeof(fa));
    int rows = 0;
    for (int l = last; l >= 0; --l) {
      if (!in_phase(l)) continue;
      FinArgs& f = fa.f[fa.n];
      if (l == last) {
        f.slabs = at<float>(ws, P.part2_off); f.nsplit = LAST_GROUPS; f.slab = P.ld_part; f.ldc = P.ld_part;
        f.colsum = at<float>(ws, P.partdb_off); f.npart = P.nwg; f.ldcs = 1;
      } else {
        f.slabs = at<float>(ws, P.dwslab_off[l]); f.nsplit = DS.nsplit[l]; f.slab = DS.slab[l]; f.ldc = P.ld_in[l];
        if (l == last - 1) { f.colsum = at<float>(ws, P.part2_off) + P.ld_in[last]; f.npart = LAST_GROUPS; f.ldcs = P.ld_part; }
        else { f.colsum = at<float>(ws, P.cs_off[l]); f.npart = P.nwg; f.ldcs = P.ldcs; }
      }
      f.g = L.g_off[l] >= 0 ? params + L.g_off[l] : nullptr;
      f.v = params + L.v_off[l];
      f.dg = L.g_off[l] >= 0 ? grads + L.g_off[l] : nullptr;
      f.dv = grads + L.v_off[l];
      f.db = grads + L.bias_off[l];
      f.out = net->out_dim[l]; f.in = net->in_dim[l]; f.accumulate = accumulate;
      if (segmode && (l == 0 || l == ks)) {
        f.hoist = 1; f.lat0 = l == 0 ? 0 : net->out_dim[l - 1]; f.hW = net->latent_size + net->geom_dim; f.ldh = P.ldh;
        f.hs = at<float>(ws, P.hs_off) + (l == 0 ? 0 : P.hstride);
      }
      if (fz != nullptr) {
        const DsdfAdamCfg* c = fz->cfg;
        const double bc1 = 1.0 - pow((double)c->beta1, (double)c->step), bc2 = 1.0 - pow((double)c->beta2, (double)c->step);
        f.adam = 1;
        f.pb = fz->params + L.bias_off[l]; f.mb = fz->exp_avg + L.bias_off[l]; f.sb = fz->exp_avg_sq + L.bias_off[l];
        f.pv = fz->params + L.v_off[l];    f.mv = fz->exp_avg + L.v_off[l];    f.sv = fz->exp_avg_sq + L.v_off[l];
        if (L.g_off[l] >= 0) { f.pg = fz->params + L.g_off[l]; f.mg = fz->exp_avg + L.g_off[l]; f.sg = fz->exp_avg_sq + L.g_off[l]; }
        int r0 = 0;
        for (int q = 0; q < l; ++q) r0 += net->out_dim[q];
        f.scale_out = fz->packed + pk.scale_off + r0;
        f.omb1 = 1.0f - c->beta1; f.b2 = c->beta2; f.omb2 = 1.0f - c->beta2;
        f.step_size = (float)((double)c->lr_decoder / bc1); f.bc2_sqrt = (float)sqrt(bc2); f.eps = c->eps;
      }
      fa.row0[fa.n] = rows;
      rows += fin_blocks(f.out, f.in);     // (`rows` counts BLOCKS of the finalize launch)
      ++fa.n;
    }
    fa.row0[fa.n] = rows;
    if (rows == 0) return 0;   // an empty bucket (fewer layers than buckets)
    if (segmode && sb->scatter != nullptr && phase <= 1) {
      hipLaunchKernelGGL(finalize_scatter_kernel, dim3(rows + sb->R), dim3(256), 0, st, fa, *sb->scatter, rows);
      LAUNCH_OK("finalize_scatter_kernel");
      *sb->scatter_done = true;
    } else {
      hipLaunchKernelGGL(finalize_all_kernel, dim3(rows), dim3(256), 0, st, fa);
      LAUNCH_OK("finalize_all_kernel");
    }
  }
  return 0;
}

FusedBwdHead make_head(const DsdfNet* net, const Plan& P, void* ws, const float* packed, const float* params, int mode,
                       int training) {
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int last = net->n_layers - 1;
  FusedBwdHead h;
  memset(&h, 0, sizeof(h));
  h.mode = mode;
  h.a_last = at<float>(ws, P.in_off[last]); h.ld_a = P.ld_in[last]; h.in_last = net->in_dim[last];
  h.w_last = packed + pk.w_off[last]; h.b_last = params + L.bias_off[last]; h.use_tanh = net->use_tanh;
  h.mask_scale = mask_scale_of(net, last - 1, training);
  h.dp_out = at<float>(ws, P.dpl_off[last - 1]); h.ld_dp = P.ld_dp;
  h.part = at<float>(ws, P.part_off); h.ld_part = P.ld_part;
  h.part_db = at<float>(ws, P.partdb_off); h.part_loss = at<float>(ws, P.partloss_off);
  h.n_act = net->out_dim[last - 1];      // (< in_last only when latent_in names the output layer: run_backward_fused points dz_out)
  return h;
}

int check_common(const DsdfNet* net, const void* packed, const void* params, const void* ws) {
  TRY(validate(net));
  if (net->fwd_bf16 && !fused_enabled()) return fail(DSDF_E_INVALID, "fwd_bf16 exists only in the fused kernels (DSDF_NO_FUSED is set)");
  if (net->gemm_split && !fused_enabled()) return fail(DSDF_E_INVALID, "gemm_split exists only in the fused kernels (DSDF_NO_FUSED is set)");
  if (!packed || !params || !ws) return fail(DSDF_E_INVALID, "NULL packed/params/workspace pointer");
  if (!aligned16(packed) || !aligned16(params) || (reinterpret_cast<uintptr_t>(ws) & 255))
    return fail(DSDF_E_INVALID, "packed/params must be 16-byte and workspace 256-byte aligned");
  return 0;
}

}  // namespace

// ===================================================================================================
extern "C" {

int dsdf_abi_version(void) { return DSDF_ABI_VERSION; }
const char* dsdf_last_error(void) { return g_err; }

int dsdf_param_layout(const DsdfNet* net, DsdfParamLayout* out) {
  TRY(validate(net));
  if (!out) return fail(DSDF_E_INVALID, "out is NULL");
  param_layout(net, out);
  return 0;
}

int dsdf_packed_floats(const DsdfNet* net, int64_t* n_floats) {
  TRY(validate(net));
  if (!n_floats) return fail(DSDF_E_INVALID, "n_floats is NULL");
  *n_floats = packed_layout(net).total;
  return 0;
}

int dsdf_workspace_bytes(const DsdfNet* net, int64_t n_points, int64_t n_segments, size_t* bytes) {
  TRY(validate(net));
  if (!bytes || n_points < 0 || n_segments < 0) return fail(DSDF_E_INVALID, "bad arguments");
  if (n_points > (1ll << 30)) return fail(DSDF_E_INVALID, "n_points too large");
  size_t best = 0;
  for (int fr = 32; fr <= FROWS; fr += 32)      // (32-row workgroups double the per-workgroup partials: pick_frows decides per call)
    for (int seg = 0; seg < 2; ++seg)           // segment mode lays the workspace out differently
      best = std::max(best, make_plan(net, n_points, n_segments, false, seg != 0, 2, fr).total);
  *bytes = best;
  return 0;
}

int dsdf_workspace_bytes_buckets(const DsdfNet* net, int64_t n_points, int64_t n_segments, int32_t n_buckets, size_t* bytes) {
  TRY(validate(net));
  if (!bytes || n_points < 0 || n_segments < 0) return fail(DSDF_E_INVALID, "bad arguments");
  if (n_points > (1ll << 30)) return fail(DSDF_E_INVALID, "n_points too large");
  if (n_buckets < 0 || n_buckets > DSDF_MAX_BUCKETS) return fail(DSDF_E_INVALID, "n_buckets %d out of range [0, %d]", n_buckets, DSDF_MAX_BUCKETS);
  size_t best = 0;
  for (int fr = 32; fr <= FROWS; fr += 32)
    for (int seg = 0; seg < 2; ++seg) best = std::max(best, make_plan(net, n_points, n_segments, false, seg != 0, n_buckets, fr).total);
  *bytes = best;
  return 0;
}

int dsdf_dw_phase_supported(const DsdfNet* net) {
  TRY(validate(net));
  return fused_enabled() && fused_eligible(net) ? 1 : 0;
}

int dsdf_grad_buckets(const DsdfNet* net, int32_t n_buckets, int32_t* first_layer, int64_t* arena_off) {
  TRY(validate(net));
  if (!first_layer || !arena_off) return fail(DSDF_E_INVALID, "NULL argument");
  if (n_buckets < 2 || n_buckets > DSDF_MAX_BUCKETS) return fail(DSDF_E_INVALID, "n_buckets %d out of range [2, %d]", n_buckets, DSDF_MAX_BUCKETS);
  DsdfParamLayout L;
  param_layout(net, &L);
  int cut[DSDF_MAX_BUCKETS + 1];
  dw_bucket_cuts(net, n_buckets, cut);
  auto layer_start = [&](int k) {              // a layer's parameters are one contiguous block: its first offset
    if (k >= net->n_layers) return L.total;
    int64_t o = L.v_off[k];
    if (L.bias_off[k] >= 0 && L.bias_off[k] < o) o = L.bias_off[k];
    if (L.g_off[k] >= 0 && L.g_off[k] < o) o = L.g_off[k];
    if (L.ln_w_off[k] >= 0 && L.ln_w_off[k] < o) o = L.ln_w_off[k];
    return o;
  };
  arena_off[0] = L.total;                      // bucket b = layers [first_layer[b], ...) = arena floats [arena_off[b + 1], arena_off[b])
  for (int b = 0; b < n_buckets; ++b) { first_layer[b] = cut[b + 1]; arena_off[b + 1] = layer_start(cut[b + 1]); }
  return 0;
}

int dsdf_decode_workspace_bytes(const DsdfNet* net, int64_t n_points, size_t* bytes) {
  TRY(validate(net));
  if (!bytes || n_points < 0 || n_points > (1ll << 30)) return fail(DSDF_E_INVALID, "bad arguments");
  *bytes = make_plan(net, n_points, 0, true).total;
  return 0;
}

int dsdf_materialize_weights(const DsdfNet* net, const float* params, float* packed, void* stream) {
  TRY(validate(net));
  if (!params || !packed) return fail(DSDF_E_INVALID, "NULL pointer");
  return materialize(net, params, packed, (hipStream_t)stream);
}

int dsdf_decode(const DsdfNet* net, const float* packed, const float* params, const float* input, int64_t ld_in,
                int64_t n, float* sdf_out, void* ws, size_t ws_bytes, void* stream) {
  TRY(check_common(net, packed, params, ws));
  if (n == 0) return 0;
  if (!input || !sdf_out || n < 0 || ld_in < net->in_dim[0]) return fail(DSDF_E_INVALID, "bad input/sdf_out/ld_in");
  const Plan P = make_plan(net, n, 0, true, false, 2, pick_frows(net, n));
  if (ws_bytes < P.total) return fail(DSDF_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, P.total);
  hipStream_t st = (hipStream_t)stream;
  TRY(run_gather(net, P, ws, nullptr, nullptr, input, ld_in, n, st));
  if (fused_enabled() && fused_eligible(net))
    return run_fused_forward(net, P, ws, packed, params, n, 0, nullptr, 0, false, sdf_out, nullptr, st);
  TRY(run_hidden_forward(net, P, ws, packed, params, n, 0, nullptr, 0, st, false));
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int last = net->n_layers - 1;
  LastArgs a;
  memset(&a, 0, sizeof(a));
  a.a = at<float>(ws, P.in_off[last]); a.lda = P.ld_in[last]; a.in = net->in_dim[last];
  a.w = packed + pk.w_off[last]; a.b = params + L.bias_off[last]; a.n = (int)n; a.use_tanh = net->use_tanh;
  a.y_out = sdf_out;
  int blocks = (int)((n + 15) / 16);
  if (blocks > 2048) blocks = 2048;
  return launch_last<LAST_FWD>(a, blocks, st);
}

static bool decode_latent_ok(const DsdfNet* net) {
  return fused_enabled() && fused_eligible(net) && net->geom_dim <= FGEO && net->latent_size <= HOIST_MAXL &&
         net->latent_size >= 1 && net->n_layers >= 3 && !last_layer_skip(net);
}

int dsdf_decode_latent_supported(const DsdfNet* net) {
  TRY(validate(net));
  return decode_latent_ok(net) ? 1 : 0;
}

int dsdf_decode_latent(const DsdfNet* net, const float* packed, const float* params, const float* latent, const float* xyz,
                       int64_t n, float* sdf_out, void* ws, size_t ws_bytes, void* stream) {
  TRY(check_common(net, packed, params, ws));
  if (n == 0) return 0;
  if (!latent || !xyz || !sdf_out || n < 0) return fail(DSDF_E_INVALID, "bad latent/xyz/sdf_out");
  if (!decode_latent_ok(net))
    return fail(DSDF_E_INVALID, "dsdf_decode_latent needs the fused forward (widths <= 512, geom_dim <= 4): use dsdf_decode");
  Plan P = make_plan(net, n, 0, true, false, 2, pick_frows(net, n));
  const size_t need = P.total > 16384 ? P.total : 16384;
  if (ws_bytes < need) return fail(DSDF_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  // ONE segment covering every point: the segment-mode forward with the latent's products hoisted (fused.hpp FusedSeg)
  HIP_OK(hipMemsetAsync(ws, 0, 8, st));                 // seg_scene[0] = 0: the "table" is the single latent row
  P.hoistU_off = 256; P.ldu = FMAXW;                    // U [1][2][512] behind it
  DsdfBatch b;
  memset(&b, 0, sizeof(b));
  b.seg_scene = at<int64_t>(ws, 0); b.n_segments = 1; b.xyz = xyz; b.n_points = n; b.seg_len = n;
  FusedSeg seg;
  TRY(run_hoist(net, P, ws, packed, latent, &b, &seg, st));
  seg.wg_per_seg = (int)((n + P.frows - 1) / P.frows);  // every workgroup belongs to segment 0
  return run_fused_forward(net, P, ws, packed, params, n, 0, nullptr, 0, false, sdf_out, nullptr, st, &seg);
}

int dsdf_module_forward(const DsdfNet* net, const float* packed, const float* params, const float* input,
                        int64_t ld_in, int64_t n, int32_t training, const uint32_t* dropout_key, float* sdf_out,
                        void* ws, size_t ws_bytes, void* stream) {
  TRY(check_common(net, packed, params, ws));
  if (n == 0) return 0;
  if (!input || !sdf_out || n < 0 || ld_in < net->in_dim[0]) return fail(DSDF_E_INVALID, "bad input/sdf_out/ld_in");
  if (training && net->dropout_p > 0.f && net->dropout_mask && !dropout_key) return fail(DSDF_E_INVALID, "dropout_key is NULL");
  const Plan P = make_plan(net, n, 0, false);
  if (ws_bytes < P.total) return fail(DSDF_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, P.total);
  hipStream_t st = (hipStream_t)stream;
  TRY(run_gather(net, P, ws, nullptr, nullptr, input, ld_in, n, st, training, dropout_key, 0));
  if (fused_enabled() && fused_eligible(net))
    return run_fused_forward(net, P, ws, packed, params, n, training, dropout_key, 0, true, sdf_out, at<float>(ws, P.u_off), st);
  TRY(run_hidden_forward(net, P, ws, packed, params, n, training, dropout_key, 0, st));
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int last = net->n_layers - 1;
  LastArgs a;
  memset(&a, 0, sizeof(a));
  a.a = at<float>(ws, P.in_off[last]); a.lda = P.ld_in[last]; a.in = net->in_dim[last];
  a.w = packed + pk.w_off[last]; a.b = params + L.bias_off[last]; a.n = (int)n; a.use_tanh = net->use_tanh;
  a.y_out = sdf_out; a.u_save = at<float>(ws, P.u_off);
  return launch_last<LAST_FWD>(a, P.last_blocks, st);
}

int dsdf_module_backward(const DsdfNet* net, const float* packed, const float* params, const float* d_sdf, int64_t n,
                         int32_t training, const uint32_t* dropout_key, float* grads, int32_t accumulate, float* d_input,
                         int64_t ld_din, void* ws, size_t ws_bytes, void* stream) {
  TRY(check_common(net, packed, params, ws));
  if (n == 0) return 0;
  if (!d_sdf || !grads || n < 0) return fail(DSDF_E_INVALID, "bad d_sdf/grads");
  if (d_input && ld_din < net->in_dim[0]) return fail(DSDF_E_INVALID, "ld_din too small");
  if (training && net->latent_dropout && d_input && !dropout_key)
    return fail(DSDF_E_INVALID, "dropout_key is NULL (latent_dropout needs the forward's key for d/d(input))");
  const Plan P = make_plan(net, n, 0, false);
  if (ws_bytes < P.total) return fail(DSDF_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, P.total);
  hipStream_t st = (hipStream_t)stream;
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int last = net->n_layers - 1;
  LastArgs a;
  memset(&a, 0, sizeof(a));
  a.a = at<float>(ws, P.in_off[last]); a.lda = P.ld_in[last]; a.in = net->in_dim[last];
  a.w = packed + pk.w_off[last]; a.b = params + L.bias_off[last]; a.n = (int)n; a.use_tanh = net->use_tanh;
  a.d_sdf = d_sdf; a.u_in = at<float>(ws, P.u_off);
  const bool fusedb = fused_enabled() && fused_eligible(net);
  a.dp_prev = at<float>(ws, P.dp_off[0]);
  a.lddp = P.ld_dp; a.mask_scale = mask_scale_of(net, last - 1, training);
  a.part_dw = at<float>(ws, P.part_off); a.ld_part = P.ld_part;
  a.part_colsum = at<float>(ws, P.part_off) + P.ld_in[last];
  a.part_db = at<float>(ws, P.partdb_off); a.part_loss = at<float>(ws, P.partloss_off);
  a.n_act = net->out_dim[last - 1];
  if (last_layer_skip(net) && d_input) { a.dz = at<float>(ws, P.dzB_off); a.ldz = P.ldz; a.dz_cols = P.W0; }
  bool used_dzB = false;
  if (fusedb) {
    FusedBwdHead h = make_head(net, P, ws, packed, params, HEAD_EXT, training);
    h.d_sdf = d_sdf; h.u_in = at<float>(ws, P.u_off);
    TRY(run_backward_fused(net, P, ws, packed, params, n, training, grads, accumulate, d_input ? P.W0 : 0, &used_dzB, st, true, h));
  } else {
    TRY(launch_last<LAST_BWD_EXT>(a, P.last_blocks, st));
    bool xyz_acc = false;
    if (d_input && net->xyz_in_all && last > 0) {   // the last layer's input is [a || xyz] too: its own d/d(xyz) opens the running sum
      const long long tot = (long long)n * net->geom_dim;
      hipLaunchKernelGGL(last_xyz_grad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_sdf, at<float>(ws, P.u_off),
                         packed + pk.w_off[last] + net->out_dim[last - 1], net->geom_dim, net->use_tanh, at<float>(ws, P.dxz_off[1]), 4, 0,
                         (int)n, 0);
      LAUNCH_OK("last_xyz_grad_kernel");
      xyz_acc = true;
    }
    TRY(run_backward(net, P, ws, packed, params, n, training, grads, accumulate, d_input ? P.W0 : 0, &used_dzB, st, true, dropout_key, 0,
                     (d_input && net->xyz_in_all) ? &xyz_acc : nullptr));
    if (last_layer_skip(net) && d_input) used_dzB = true;      // (last_layer_kernel wrote it)
    if (d_input) {
      const long long tot = (long long)n * P.W0;
      hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, at<float>(ws, P.dzA_off), P.ldz,
                         used_dzB ? at<float>(ws, P.dzB_off) : nullptr, P.ldz, d_input, (long long)ld_din, (int)n, P.W0);
      LAUNCH_OK("add2_kernel");
      if (xyz_acc) {
        const long long tx = (long long)n * net->geom_dim;
        hipLaunchKernelGGL(acc_cols_kernel, dim3((unsigned)((tx + 255) / 256)), dim3(256), 0, st, at<float>(ws, P.dxz_off[1]), 4, d_input,
                           (int)ld_din, net->latent_size, (int)n, net->geom_dim, 1);
        LAUNCH_OK("acc_cols_kernel");
      }
    }
    return 0;
  }
  if (d_input) {
    const long long tot = (long long)n * P.W0;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, at<float>(ws, P.dzA_off), P.ldz,
                       used_dzB ? at<float>(ws, P.dzB_off) : nullptr, P.ldz, d_input, (long long)ld_din, (int)n, P.W0);
    LAUNCH_OK("add2_kernel");
  }
  return 0;
}

// Forward-mode tangent of the decoder at the point of the last dsdf_module_forward on this workspace:
//   t_0 = tangent;  t_{l+1} = (in_l-tangent W_l^T) * [a_{l+1} > 0] * mask_scale   (same ReLU / dropout decisions as the primal:
//   the mask is read off the stored activations);  skip layer: in-tangent = [t_l | tangent];  out = tanh' ... tanh' (t_last w_last)
// Layer-by-layer MFMA GEMM launches (gemm.hpp) -- this is the one-extra-pass tool of mesh.py:420, not the training hot path.
int dsdf_module_jvp(const DsdfNet* net, const float* packed, const float* params, const float* tangent, int64_t ld_t,
                    int64_t n, int32_t training, const uint32_t* dropout_key, float* jvp_out, void* ws, size_t ws_bytes, void* stream) {
  TRY(check_common(net, packed, params, ws));
  if (n == 0) return 0;
  if (!tangent || !jvp_out || n < 0 || ld_t < net->in_dim[0]) return fail(DSDF_E_INVALID, "bad tangent/jvp_out/ld_t");
  const bool lat_drop = net->latent_dropout && training && net->latent_size > 0;
  if (lat_drop && !dropout_key) return fail(DSDF_E_INVALID, "dropout_key is NULL (latent_dropout needs the forward's key)");
  const Plan P = make_plan(net, n, 0, false);
  if (ws_bytes < P.total) return fail(DSDF_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, P.total);
  hipStream_t st = (hipStream_t)stream;
  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int last = net->n_layers - 1;
  float* t0 = at<float>(ws, P.dzA_off);                      // the input tangent in a 16-byte-aligned, zero-padded layout
  HIP_OK(hipMemsetAsync(t0, 0, (size_t)n * P.ldz * 4, st));
  HIP_OK(hipMemsetAsync(at<float>(ws, P.dp_off[0]), 0, (size_t)n * P.ld_dp * 4, st));
  HIP_OK(hipMemsetAsync(at<float>(ws, P.dp_off[1]), 0, (size_t)n * P.ld_dp * 4, st));
  {
    const long long tot = (long long)n * P.W0;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, tangent, (int)ld_t, (const float*)nullptr, 0,
                       t0, (long long)P.ldz, (int)n, P.W0);
    LAUNCH_OK("add2_kernel(tangent)");
  }
  float* t0_layer0 = t0;                                     // latent_dropout: layer 0 sees the tangent of the DROPPED latent,
  if (lat_drop) {                                            // the skip layer the raw one (deep_sdf_decoder.py:79-89)
    t0_layer0 = at<float>(ws, P.dzB_off);
    HIP_OK(hipMemcpyAsync(t0_layer0, t0, (size_t)n * P.ldz * 4, hipMemcpyDeviceToDevice, st));
    const long long tot = (long long)n * net->latent_size;
    hipLaunchKernelGGL(latent_drop_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, t0_layer0, P.ldz, (int)n,
                       net->latent_size, dropout_key[LATENT_DROPOUT_KEY], latent_drop_thr(), 1.0f / (1.0f - LATENT_DROPOUT_P), 0u);
    LAUNCH_OK("latent_drop_bwd_kernel(tangent)");
  }
  auto append = [&](float* in_t, int l) -> int {             // what the forward concatenates to layer l's input, for the tangent
    const bool skip = (net->skip_mask >> l) & 1;
    if (!skip && !net->xyz_in_all) return 0;
    const int w = skip ? P.W0 : net->geom_dim;
    const float* src = skip ? t0 : t0 + net->latent_size;
    const long long tot = (long long)n * w;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, P.ldz, (const float*)nullptr, 0,
                       in_t + net->out_dim[l - 1], (long long)P.ld_dp, (int)n, w);
    return 0;
  };
  int cur = 0;
  for (int l = 0; l < last; ++l) {
    float* in_t = l == 0 ? t0_layer0 : at<float>(ws, P.dp_off[cur]);
    if (l > 0) { append(in_t, l); LAUNCH_OK("add2_kernel(concat tangent)"); }
    NtArgs a;
    memset(&a, 0, sizeof(a));
    a.A = in_t; a.lda = l == 0 ? P.ldz : P.ld_dp;
    a.B = packed + pk.w_off[l]; a.ldb = pk.ldw[l];
    a.C = at<float>(ws, P.dp_off[l == 0 ? 0 : cur ^ 1]); a.ldc = P.ld_dp;
    a.M = (int)n; a.N = net->out_dim[l]; a.K = net->in_dim[l];
    if (ln_applied(net, l)) {                                // Linear tangent, then LayerNorm + ReLU/dropout in one row pass
      TRY(launch_nt<EPI_PLAIN>(a, st));
      LnJvpArgs j;
      memset(&j, 0, sizeof(j));
      j.t = a.C; j.ldt = a.ldc; j.xhat = at<float>(ws, P.lnx_off[l]); j.ldx = P.ld_in[l + 1]; j.rstd = at<float>(ws, P.lnr_off[l]);
      j.gamma = params + L.ln_w_off[l]; j.act = at<float>(ws, P.in_off[l + 1]); j.ldact = P.ld_in[l + 1];
      j.mask_scale = mask_scale_of(net, l, training); j.n = (int)n; j.width = net->out_dim[l];
      hipLaunchKernelGGL(ln_jvp_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, j);
      LAUNCH_OK("ln_jvp_kernel");
    } else {
      a.act = at<float>(ws, P.in_off[l + 1]); a.ldact = P.ld_in[l + 1];
      a.mask_cols = net->out_dim[l]; a.mask_scale = mask_scale_of(net, l, training);
      TRY(launch_nt<EPI_BWD>(a, st));
    }
    if (l > 0) cur ^= 1;
  }
  if (last > 0) { append(at<float>(ws, P.dp_off[cur]), last); LAUNCH_OK("add2_kernel(concat tangent, last)"); }
  NtArgs a;                                                   // du = t_last . w_last
  memset(&a, 0, sizeof(a));
  a.A = at<float>(ws, P.dp_off[cur]); a.lda = P.ld_dp;
  a.B = packed + pk.w_off[last]; a.ldb = pk.ldw[last];
  a.C = at<float>(ws, P.y_off); a.ldc = 1;
  a.M = (int)n; a.N = 1; a.K = net->in_dim[last];
  TRY(launch_nt<EPI_PLAIN>(a, st));
  hipLaunchKernelGGL(jvp_tail_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, at<float>(ws, P.y_off),
                     at<float>(ws, P.u_off), jvp_out, (int)n, net->use_tanh);
  LAUNCH_OK("jvp_tail_kernel");
  return 0;
}

}  // extern "C"

namespace {
// returns 1 in *adam_fused when the decoder's Adam update (and the new weight-norm scales) was folded into the finalize pass
int train_fb_impl(const DsdfNet* net, const float* packed, const float* params, float* latent_table, int64_t n_scenes,
                  const DsdfBatch* b, const DsdfLossCfg* cfg, float* grads, float* dlat, float* loss_out, float* sdf_out,
                  int32_t accumulate, void* ws, size_t ws_bytes, void* stream, const FuseAdam* fz, int* adam_fused) {
  if (adam_fused) *adam_fused = 0;
  TRY(check_common(net, packed, params, ws));
  if (!b || !cfg || !latent_table || !grads || !dlat || !loss_out) return fail(DSDF_E_INVALID, "NULL argument");
  const int64_t n = b->n_points, R = b->n_segments;
  if (n <= 0 || R <= 0 || n_scenes <= 0) return fail(DSDF_E_INVALID, "empty batch (n_points %lld, n_segments %lld)", (long long)n, (long long)R);
  if (!b->seg_scene || !b->seg_offset || !b->xyz || !b->sdf_gt) return fail(DSDF_E_INVALID, "NULL batch pointer");
  if (b->n_norm <= 0) return fail(DSDF_E_INVALID, "n_norm must be positive");
  if (net->latent_size <= 0) return fail(DSDF_E_INVALID, "training needs latent_size > 0");
  // Segment mode: the batch is scenes x samples with every segment a whole number of 64-row workgroups, so a workgroup
  // sees ONE latent vector: its products with the weights are hoisted out of the per-point work (fused.hpp FusedSeg),
  // and the latent gradient / the x0 columns of the weight gradients come from per-workgroup column sums.
  const bool fusedb = fused_enabled() && fused_eligible(net);
  const int skip_l = skip_layer(net);
  // (32-row workgroups exist for the merged forward + backward launch only)
  const bool can_merge = fusedb && !getenv("DSDF_NO_MERGE")
#ifdef DSDF_LAB
                         && !getenv("DSDF_LAB_DBG")
#endif
      ;
  const int frows = can_merge ? pick_frows(net, n, true) : FROWS;
  const bool segsum = fusedb && b->seg_len > 0 && b->seg_len % frows == 0 && b->seg_len * R == n && net->n_layers > 2 &&
                      skip_l != net->n_layers - 2 &&   // the deepest hidden layer's dP column sums live in the head's partials
                      !last_layer_skip(net) &&
                      net->geom_dim <= FGEO && net->latent_size <= HOIST_MAXL;   // (config 5 too: bf16 rounding is element-wise
                                                                                  // on the operands, so the latent products still hoist)
  const int phase = cfg->dw_phase, nbk = cfg->dw_buckets;
  if (nbk < 0 || nbk > DSDF_MAX_BUCKETS) return fail(DSDF_E_INVALID, "dw_buckets %d out of range [0, %d]", nbk, DSDF_MAX_BUCKETS);
  if (phase < 0 || (nbk <= 1 ? phase != 0 : phase < 1 || phase > nbk))
    return fail(DSDF_E_INVALID, "dw_phase %d out of range for dw_buckets %d (0 without buckets, 1..K with K >= 2)", phase, nbk);
  const Plan P = make_plan(net, n, R, false, segsum, nbk, frows);
  if (ws_bytes < P.total) return fail(DSDF_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, P.total);
  hipStream_t st = (hipStream_t)stream;
  const int Lc = net->latent_size;
  if (phase != 0 && (!fusedb || cfg->frozen_decoder || accumulate || fz != nullptr))
    return fail(DSDF_E_INVALID, "dw_phase needs the fused kernels (dsdf_dw_phase_supported), a trainable decoder, accumulate = 0 and the two-call path");
  if (phase >= 2) {   // only the weight gradients of bucket phase - 1, from what the phase-1 call left in the workspace
    FusedSeg seg0;
    memset(&seg0, 0, sizeof(seg0));
    const FusedBwdHead h0 = make_head(net, P, ws, packed, params, HEAD_TRAIN, cfg->training);
    const SegBwd sb0{&seg0, b->seg_scene, latent_table, (int)R, nullptr, nullptr, nullptr};
    bool used = false;
    return run_backward_fused(net, P, ws, packed, params, n, cfg->training, grads, 0, segsum ? 0 : Lc, &used, st, true, h0, nullptr,
                              segsum ? &sb0 : nullptr, nullptr, phase);
  }

  if (!segsum && (cfg->code_bound > 0.f || !accumulate)) {   // max-norm renorm of the looked-up rows + zero of the dense latent gradient
    // (segment mode: both are part of the hoist launch -- seg_hoist_kernel)
    const long long nzero = accumulate ? 0 : (long long)n_scenes * Lc;
    long long blocks = (R + 3) / 4, zb = (nzero + 4095) / 4096;
    if (zb > 2048) zb = 2048;
    if (zb > blocks) blocks = zb;
    hipLaunchKernelGGL(latent_renorm_kernel, dim3((unsigned)blocks), dim3(256), 0, st, latent_table, Lc, b->seg_scene, (int)R,
                       cfg->code_bound > 0.f ? cfg->code_bound : 0.f, accumulate ? nullptr : dlat, nzero);
    LAUNCH_OK("latent_renorm_kernel");
  }
  FusedSeg seg;
  memset(&seg, 0, sizeof(seg));
  FusedFwdArgs fwd_args;                                   // fp32 fused path: forward + backward go out as ONE launch below
  const bool merged = can_merge;   // (DSDF_NO_MERGE -- lab / tests: forward and backward as two launches in fp32 too; lab builds with
                                   // DSDF_LAB_DBG: per-layer stamps are dumped after a forward launch of its own)
  if (segsum) {
    const HoistRenorm hr{cfg->code_bound > 0.f ? cfg->code_bound : 0.f, dlat, accumulate ? 0 : (long long)n_scenes * Lc};
    TRY(run_hoist(net, P, ws, packed, latent_table, b, &seg, st, &hr));
    TRY(run_fused_forward(net, P, ws, packed, params, n, cfg->training, cfg->dropout_key, (uint32_t)b->row_offset, true,
                          nullptr, nullptr, st, &seg, merged ? &fwd_args : nullptr));
  } else {
    TRY(run_gather(net, P, ws, latent_table, b, nullptr, 0, n, st, cfg->training, cfg->dropout_key, (uint32_t)b->row_offset));
    if (fusedb)
      TRY(run_fused_forward(net, P, ws, packed, params, n, cfg->training, cfg->dropout_key, (uint32_t)b->row_offset, true,
                            nullptr, nullptr, st, nullptr, merged ? &fwd_args : nullptr));
    else
      TRY(run_hidden_forward(net, P, ws, packed, params, n, cfg->training, cfg->dropout_key, (uint32_t)b->row_offset, st));
  }

  DsdfParamLayout L;
  param_layout(net, &L);
  const Packed pk = packed_layout(net);
  const int last = net->n_layers - 1;
  LastArgs a;
  memset(&a, 0, sizeof(a));
  a.a = at<float>(ws, P.in_off[last]); a.lda = P.ld_in[last]; a.in = net->in_dim[last];
  a.w = packed + pk.w_off[last]; a.b = params + L.bias_off[last]; a.n = (int)n; a.use_tanh = net->use_tanh;
  a.y_out = sdf_out; a.gt = b->sdf_gt; a.delta = cfg->clamp_dist; a.inv_n = 1.0f / (float)b->n_norm;
  a.dp_prev = at<float>(ws, P.dp_off[0]);
  a.lddp = P.ld_dp; a.mask_scale = mask_scale_of(net, last - 1, cfg->training);
  a.part_dw = at<float>(ws, P.part_off); a.ld_part = P.ld_part;
  a.part_colsum = at<float>(ws, P.part_off) + P.ld_in[last];
  a.part_db = at<float>(ws, P.partdb_off); a.part_loss = at<float>(ws, P.partloss_off);
  a.n_act = net->out_dim[last - 1];
  if (last_layer_skip(net)) { a.dz = at<float>(ws, P.dzB_off); a.ldz = P.ldz; a.dz_cols = Lc; }
  if (!fusedb) TRY(launch_last<LAST_TRAIN>(a, P.last_blocks, st));

  bool used_dzB = false;
  const bool want_dw = cfg->frozen_decoder == 0;
  ScatterArgs sc;   // dense latent gradient + regulariser + loss (block 0); launched below unless it rode on the finalize launch
  memset(&sc, 0, sizeof(sc));
  sc.segpart = at<float>(ws, P.segpart_off); sc.segnorm = at<float>(ws, P.segnorm_off);
  sc.seg_scene = b->seg_scene; sc.seg_offset = b->seg_offset;
  sc.R = (int)R; sc.L = Lc; sc.table = latent_table; sc.dlat = dlat;
  sc.zr = segsum ? at<float>(ws, P.zr_off) : nullptr;
  sc.creg = cfg->reg_coef / (float)b->n_norm;
  sc.part_loss = at<float>(ws, P.partloss_off); sc.n_part = fusedb ? P.nwg : P.last_blocks;
  sc.loss_scale = 1.0f / (float)b->n_norm; sc.loss_out = loss_out; sc.accumulate = accumulate;
  bool scatter_done = false;
  if (fusedb) {
    FusedBwdHead h = make_head(net, P, ws, packed, params, HEAD_TRAIN, cfg->training);
    h.gt = b->sdf_gt; h.delta = cfg->clamp_dist; h.inv_n = 1.0f / (float)b->n_norm; h.y_out = sdf_out;
    const FuseAdam* use = (fz != nullptr && want_dw && !accumulate) ? fz : nullptr;
    const SegBwd sb{&seg, b->seg_scene, latent_table, (int)R, &sc, &scatter_done, sc.zr};
    TRY(run_backward_fused(net, P, ws, packed, params, n, cfg->training, grads, accumulate, segsum ? 0 : Lc, &used_dzB, st, want_dw, h,
                           use, segsum ? &sb : nullptr, merged ? &fwd_args : nullptr, phase));
    if (use != nullptr && adam_fused) *adam_fused = 1;
  } else {
    TRY(run_backward(net, P, ws, packed, params, n, cfg->training, grads, accumulate, Lc, &used_dzB, st, want_dw, cfg->dropout_key,
                     (uint32_t)b->row_offset));
    if (last_layer_skip(net)) used_dzB = true;                 // (last_layer_kernel wrote it)
  }

  SegArgs s;
  memset(&s, 0, sizeof(s));
  s.dzA = at<float>(ws, P.dzA_off); s.dzB = used_dzB ? at<float>(ws, P.dzB_off) : nullptr; s.ldz = P.ldz;
  s.seg_scene = b->seg_scene; s.seg_offset = b->seg_offset; s.R = (int)R; s.L = Lc; s.table = latent_table;
  s.segpart = at<float>(ws, P.segpart_off); s.segnorm = at<float>(ws, P.segnorm_off);
  if (!segsum) {   // (segment mode: per-segment latent gradients came out of post_bwd_kernel)
    // few long segments: cut their rows into slices until the launch has blocks for the chip (the scatter adds the slices)
    int nsl = 1;
    const long long blocks = (long long)R * ((Lc + 63) / 64);
    while (nsl < LAT_SLICES_MAX && blocks * nsl * 2 <= chip_waves() / 4 && n / (R * (int64_t)nsl * 2) >= 128) nsl *= 2;
    s.nslice = nsl; s.slice_stride = (long long)R * Lc;
    sc.nslice = nsl; sc.slice_stride = s.slice_stride;
    hipLaunchKernelGGL(seg_reduce_kernel, dim3((unsigned)R, (Lc + 63) / 64, (unsigned)nsl), dim3(256), 0, st, s);
    LAUNCH_OK("seg_reduce_kernel");
  }
  if (!scatter_done) {
    hipLaunchKernelGGL(seg_scatter_kernel, dim3((unsigned)R), dim3(256), 0, st, sc);   // + the loss (block 0)
    LAUNCH_OK("seg_scatter_kernel");
  }
  return 0;
}
}  // namespace

extern "C" {

int dsdf_train_forward_backward(const DsdfNet* net, const float* packed, const float* params, float* latent_table,
                                int64_t n_scenes, const DsdfBatch* b, const DsdfLossCfg* cfg, float* grads, float* dlat,
                                float* loss_out, float* sdf_out, int32_t accumulate, void* ws, size_t ws_bytes,
                                void* stream) {
  return train_fb_impl(net, packed, params, latent_table, n_scenes, b, cfg, grads, dlat, loss_out, sdf_out, accumulate, ws,
                       ws_bytes, stream, nullptr, nullptr);
}

int dsdf_grad_norm(const float* grads, int64_t n, float max_norm, float* norm_out, float* coef_out, void* ws,
                   size_t ws_bytes, void* stream) {
  if (!grads || !norm_out || !coef_out || !ws || n <= 0) return fail(DSDF_E_INVALID, "bad arguments");
  int blocks = (int)((n + 4095) / 4096);
  if (blocks > 1024) blocks = 1024;
  if (ws_bytes < (size_t)blocks * 4) return fail(DSDF_E_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, grads, (long long)n, (float*)ws);
  LAUNCH_OK("sumsq_partial_kernel");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, blocks, max_norm, norm_out, coef_out);
  LAUNCH_OK("clip_coef_kernel");
  return 0;
}

static int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, const DsdfAdamCfg* c,
                       const float* gscale, hipStream_t st) {
  if (n <= 0) return 0;
  const double bc1 = 1.0 - pow((double)c->beta1, (double)c->step);
  const double bc2 = 1.0 - pow((double)c->beta2, (double)c->step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, (long long)n, 1.0f - c->beta1, c->beta2,
                     1.0f - c->beta2, step_size, bc2_sqrt, c->eps, gscale);
  LAUNCH_OK("adam_kernel");
  return 0;
}

int dsdf_adam_step(const DsdfNet* net, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   float* latent_table, const float* dlat, float* lat_exp_avg, float* lat_exp_avg_sq,
                   int64_t n_latent_floats, const DsdfAdamCfg* cfg, float* packed, void* stream) {
  TRY(validate(net));
  if (!params || !grads || !exp_avg || !exp_avg_sq || !cfg || !packed) return fail(DSDF_E_INVALID, "NULL argument");
  if (cfg->step < 1) return fail(DSDF_E_INVALID, "Adam step must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  DsdfParamLayout L;
  param_layout(net, &L);
  {   // decoder: Adam per (layer, row) + the row's new weight-norm scale in one pass
    const Packed pk = packed_layout(net);
    AdamRowsArgs a;
    memset(&a, 0, sizeof(a));
    a.nl = net->n_layers; a.p = params; a.g = grads; a.m = exp_avg; a.s = exp_avg_sq; a.scale = packed + pk.scale_off;
    const double bc1 = 1.0 - pow((double)cfg->beta1, (double)cfg->step), bc2 = 1.0 - pow((double)cfg->beta2, (double)cfg->step);
    a.omb1 = 1.0f - cfg->beta1; a.b2 = cfg->beta2; a.omb2 = 1.0f - cfg->beta2;
    a.step_size = (float)((double)cfg->lr_decoder / bc1); a.bc2_sqrt = (float)sqrt(bc2); a.eps = cfg->eps; a.gscale = cfg->grad_scale;
    int rows = 0;
    for (int l = 0; l < net->n_layers; ++l) {
      a.ly[l] = AdamRowsLayer{L.v_off[l], L.g_off[l], L.bias_off[l], net->out_dim[l], net->in_dim[l], rows};
      rows += net->out_dim[l];
    }
    a.total_rows = rows;
    hipLaunchKernelGGL(adam_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
    LAUNCH_OK("adam_rows_kernel");
    for (int l = 0; l < net->n_layers; ++l)     // LayerNorm variant: bn{l}.weight | bn{l}.bias are adjacent in the arena
      if (L.ln_w_off[l] >= 0)
        TRY(adam_launch(params + L.ln_w_off[l], grads + L.ln_w_off[l], exp_avg + L.ln_w_off[l], exp_avg_sq + L.ln_w_off[l],
                        2 * (int64_t)net->out_dim[l], cfg->lr_decoder, cfg, cfg->grad_scale, st));
  }
  if (n_latent_floats > 0) {
    if (!latent_table || !dlat || !lat_exp_avg || !lat_exp_avg_sq) return fail(DSDF_E_INVALID, "NULL latent argument");
    const AdamRide ride = adam_ride(latent_table, dlat, lat_exp_avg, lat_exp_avg_sq, n_latent_floats, cfg->lr_latent, cfg);
    return materialize(net, params, packed, st, true, &ride);
  }
  return materialize(net, params, packed, st, true);
}

int dsdf_train_step(const DsdfNet* net, float* packed, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                    float* latent_table, int64_t n_scenes, float* dlat, float* lat_exp_avg, float* lat_exp_avg_sq,
                    const DsdfBatch* b, const DsdfLossCfg* cfg, const DsdfAdamCfg* adam, float* loss_out, float* sdf_out,
                    void* ws, size_t ws_bytes, void* stream) {
  if (!adam || !exp_avg || !exp_avg_sq || !lat_exp_avg || !lat_exp_avg_sq) return fail(DSDF_E_INVALID, "NULL argument");
  if (adam->step < 1) return fail(DSDF_E_INVALID, "Adam step must be >= 1");
  FuseAdam fz{adam, params, exp_avg, exp_avg_sq, packed};
  int fused = 0;
  const bool can_fuse = adam->grad_scale == nullptr && cfg && !cfg->frozen_decoder;
  TRY(train_fb_impl(net, packed, params, latent_table, n_scenes, b, cfg, grads, dlat, loss_out, sdf_out, 0, ws, ws_bytes, stream,
                    can_fuse ? &fz : nullptr, &fused));
  const int64_t nlat = n_scenes * net->latent_size;
  if (!fused)
    return dsdf_adam_step(net, params, grads, exp_avg, exp_avg_sq, latent_table, dlat, lat_exp_avg, lat_exp_avg_sq, nlat, adam,
                          packed, stream);
  hipStream_t st = (hipStream_t)stream;
  const AdamRide ride = adam_ride(latent_table, dlat, lat_exp_avg, lat_exp_avg_sq, nlat, adam->lr_latent, adam);
  return materialize(net, params, packed, st, true, &ride);   // latent Adam + W / W^T / fragment copies in one launch
}

int dsdf_adam_latent_only(float* latent, const float* dlat, float* exp_avg, float* exp_avg_sq, int64_t n,
                          const DsdfAdamCfg* cfg, void* stream) {
  if (!latent || !dlat || !exp_avg || !exp_avg_sq || !cfg || n <= 0) return fail(DSDF_E_INVALID, "bad arguments");
  if (cfg->step < 1) return fail(DSDF_E_INVALID, "Adam step must be >= 1");
  return adam_launch(latent, dlat, exp_avg, exp_avg_sq, n, cfg->lr_latent, cfg, nullptr, (hipStream_t)stream);
}

int dsdf_adam_latent_sched(float* latent, const float* dlat, float* exp_avg, float* exp_avg_sq, int64_t n, const float* sched,
                           int64_t n_steps, int64_t* step_counter, float beta1, float beta2, float eps, float l2_coef, void* stream) {
  if (!latent || !dlat || !exp_avg || !exp_avg_sq || !sched || !step_counter || n <= 0 || n_steps <= 0)
    return fail(DSDF_E_INVALID, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_sched_kernel, dim3(blocks), dim3(256), 0, st, latent, dlat, exp_avg, exp_avg_sq, (long long)n, sched,
                     (long long)n_steps, (const long long*)step_counter, 1.0f - beta1, beta2, 1.0f - beta2, eps, l2_coef);
  LAUNCH_OK("adam_sched_kernel");
  hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, st, (long long*)step_counter);
  LAUNCH_OK("counter_inc_kernel");
  return 0;
}

static int sample_launch(const float* data, int32_t geom_dim, const int64_t* pos_start, const int64_t* n_pos, const int64_t* neg_start,
                         const int64_t* n_neg, const int64_t* scene_ids, int64_t n_batch_scenes, int64_t subsample, uint64_t key,
                         uint64_t key_step, const int64_t* counter, float* xyz_out, float* sdf_out, void* stream) {
  if (!data || !pos_start || !n_pos || !neg_start || !n_neg || !scene_ids || !xyz_out || !sdf_out)
    return fail(DSDF_E_INVALID, "NULL argument");
  if (geom_dim < 1 || geom_dim > 16) return fail(DSDF_E_INVALID, "geom_dim %d out of range", geom_dim);
  const int64_t S = 2 * (subsample / 2);
  if (n_batch_scenes <= 0 || S <= 0 || n_batch_scenes * S > (1ll << 31) - 1) return fail(DSDF_E_INVALID, "bad batch shape");
  SampleArgs a;
  memset(&a, 0, sizeof(a));
  a.data = data; a.row_floats = geom_dim + 1; a.G = geom_dim;
  a.pos_start = pos_start; a.n_pos = n_pos; a.neg_start = neg_start; a.n_neg = n_neg;
  a.scene_ids = scene_ids; a.B = (int)n_batch_scenes; a.S = (int)S; a.key = key; a.xyz = xyz_out; a.sdf = sdf_out;
  a.counter = (const long long*)counter; a.key_step = key_step;
  const long long tot = n_batch_scenes * S;
  hipLaunchKernelGGL(sample_batch_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  LAUNCH_OK("sample_batch_kernel");
  return 0;
}

int dsdf_sample_batch(const float* data, int32_t geom_dim, const int64_t* pos_start, const int64_t* n_pos,
                      const int64_t* neg_start, const int64_t* n_neg, const int64_t* scene_ids, int64_t n_batch_scenes,
                      int64_t subsample, uint64_t key, float* xyz_out, float* sdf_out, void* stream) {
  return sample_launch(data, geom_dim, pos_start, n_pos, neg_start, n_neg, scene_ids, n_batch_scenes, subsample, key, 0, nullptr, xyz_out,
                       sdf_out, stream);
}

int dsdf_sample_batch_seq(const float* data, int32_t geom_dim, const int64_t* pos_start, const int64_t* n_pos,
                          const int64_t* neg_start, const int64_t* n_neg, const int64_t* scene_ids, int64_t n_batch_scenes,
                          int64_t subsample, uint64_t key0, uint64_t key_step, const int64_t* counter, float* xyz_out, float* sdf_out,
                          void* stream) {
  if (!counter) return fail(DSDF_E_INVALID, "counter is NULL");
  return sample_launch(data, geom_dim, pos_start, n_pos, neg_start, n_neg, scene_ids, n_batch_scenes, subsample, key0, key_step, counter,
                       xyz_out, sdf_out, stream);
}

int dsdf_profile_enable(int32_t on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  for (int c = 0; c < Prof::NCLS; ++c) g_prof.flops[c] = 0;
  return 0;
}

int dsdf_profile_read(DsdfProfile* out) {
  if (!out) return fail(DSDF_E_INVALID, "out is NULL");
  memset(out, 0, sizeof(*out));
  Prof& P = g_prof;
  for (int i = 0; i < P.used; ++i) {
    HIP_OK(hipEventSynchronize(P.ev[i][1]));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, P.ev[i][0], P.ev[i][1]));
    out->ms[P.cls[i]] += ms;
    out->count[P.cls[i]] += 1;
  }
  for (int c = 0; c < Prof::NCLS; ++c) out->flops[c] = P.flops[c];
  out->dropped = P.used >= Prof::POOL;
  P.used = 0;
  for (int c = 0; c < Prof::NCLS; ++c) P.flops[c] = 0;
  return 0;
}

int dsdf_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                 int64_t K, const float* bias, void* stream) {
  if (!A || !B || !C) return fail(DSDF_E_INVALID, "NULL operand");
  NtArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.B = B; a.C = C; a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = (int)ldc; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.bias = bias;
  return launch_nt<EPI_PLAIN>(a, (hipStream_t)stream);
}

int dsdf_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                 int64_t K, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !C || !ws) return fail(DSDF_E_INVALID, "NULL operand");
  int ns = (int)((K + 255) / 256);
  if (ns > NSPLIT_MAX) ns = NSPLIT_MAX;
  if (ns < 1) ns = 1;
  const int kchunk = (int)rup((K + ns - 1) / ns, BK);
  const int nsplit = (int)((K + kchunk - 1) / kchunk);
  const long long slab = rup(M * ldc, 64);
  if (ws_bytes < (size_t)nsplit * slab * 4) return fail(DSDF_E_WORKSPACE, "gemm_tn needs %lld bytes of workspace", (long long)nsplit * slab * 4);
  TnArgs t;
  memset(&t, 0, sizeof(t));
  t.A = A; t.B = B; t.C = (float*)ws; t.lda = (int)lda; t.ldb = (int)ldb; t.ldc = (int)ldc; t.M = (int)M; t.N = (int)N;
  t.K = (int)K; t.kchunk = kchunk; t.slab = slab;
  hipStream_t st = (hipStream_t)stream;
  TRY(launch_tn(t, nsplit, st));
  // plain fixed-order slab sum (no weight norm): reuse the finalize kernel with g == NULL and no bias partials
  FinArgs f;
  memset(&f, 0, sizeof(f));
  if (N > 2048 || ldc != N) return fail(DSDF_E_INVALID, "gemm_tn test entry needs ldc == N <= 2048");
  f.slabs = t.C; f.nsplit = nsplit; f.slab = slab; f.ldc = (int)ldc; f.colsum = nullptr; f.npart = 0; f.ldcs = 0;
  f.dv = C; f.db = (float*)ws + (size_t)nsplit * slab;  // scratch row of M floats behind the slabs
  if (ws_bytes < ((size_t)nsplit * slab + M) * 4) return fail(DSDF_E_WORKSPACE, "gemm_tn workspace too small");
  f.out = (int)M; f.in = (int)N;
  hipLaunchKernelGGL(finalize_layer_kernel, dim3((unsigned)M), dim3(256), 0, st, f);
  LAUNCH_OK("finalize_layer_kernel(test)");
  return 0;
}

int dsdf_dropout_mask(uint32_t key, float p, int64_t rows, int64_t cols, int64_t row_offset, uint8_t* out, void* stream) {
  if (!out || rows <= 0 || cols <= 0) return fail(DSDF_E_INVALID, "bad arguments");
  long thr = lround((double)p * 65536.0);
  if (thr > 65535) thr = 65535;
  const long long tot = rows * cols;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, key,
                     (uint32_t)thr, (int)rows, (int)cols, (uint32_t)row_offset, out);
  LAUNCH_OK("dropout_mask_kernel");
  return 0;
}

}  // extern "C"

// Last function of the translation unit's device code: warm_own_code (common.hpp) clamps its reads to this address.
namespace dsdf {
__device__ __noinline__ void dsdf_text_end_marker() { asm volatile("s_nop 0"); }
}  // namespace dsdf
