// sample.hpp -- per-step subsampling of the SDF samples on the device (gfx950): the reference's
// unpack_sdf_samples (deep_sdf/data.py:74-110): per scene `half` positive and `half` negative rows drawn WITHOUT
// replacement (torch.randperm(len)[:n]), a shortfall of one sign taken from the other, positives first.
// torch's Philox stream cannot be reproduced, so the specification is a keyed pseudo-random PERMUTATION
// (oracle/deepsdf_oracle.py sample_perm): a 4-round Feistel network on the smallest even-bit domain >= len with cycle
// walking; slot i of a scene reads row perm(i).  One thread per output row; integer work, bit-exact with the oracle.
#pragma once
#include "common.hpp"

namespace dsdf {

__device__ __forceinline__ uint32_t sample_perm(uint32_t i, uint32_t len, uint32_t key) {
  uint32_t bits = 2;
  while (bits < 32 && (1u << bits) < len) ++bits;     // ceil(log2(len)), at least 2
  bits += bits & 1u;                                  // even: two halves of h bits  (len <= 2^30)
  const uint32_t h = bits >> 1, mask = (1u << h) - 1u;
  uint32_t x = i;
  do {
    uint32_t L = x >> h, R = x & mask;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
      const uint32_t F = lowbias32(R * 0x9E3779B1u + key + r * 0x85EBCA77u) & mask;
      const uint32_t t = L ^ F;
      L = R; R = t;
    }
    x = (L << h) | R;
  } while (x >= len);
  return x;
}
__device__ __forceinline__ uint32_t sample_key(uint64_t key, uint32_t scene, uint32_t sign) {
  return lowbias32((uint32_t)key ^ lowbias32((uint32_t)(key >> 32) + (2u * scene + sign) * 0x9E3779B1u));
}

struct SampleArgs {
  const float* data; int row_floats; int G;            // [rows][row_floats]: xyz (G) then sdf
  const int64_t* pos_start; const int64_t* n_pos; const int64_t* neg_start; const int64_t* n_neg;   // per scene of the cache
  const int64_t* scene_ids; int B; int S;              // S = 2 * (subsample / 2) rows per scene
  uint64_t key;
  const long long* counter; uint64_t key_step;         // optional (graph-captured loops): the draw key is key + *counter * key_step
  float* xyz; float* sdf;                              // [B*S][G], [B*S]
};
__global__ __launch_bounds__(256) void sample_batch_kernel(const SampleArgs p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.B * p.S) return;
  const int b = (int)(t / p.S), j = (int)(t - (long long)b * p.S);
  const int64_t sc = p.scene_ids[b];
  const int64_t np = p.n_pos[sc], nn = p.n_neg[sc];
  const int half = p.S >> 1;
  int cp = half, cn = half;                            // deep_sdf/data.py:83-91
  if (np < half) { cp = (int)np; cn = p.S - cp; }
  else if (nn < half) { cn = (int)nn; cp = p.S - cn; }
  const uint64_t key = p.counter != nullptr ? p.key + (uint64_t)*p.counter * p.key_step : p.key;
  int64_t row;
  if (j < cp) row = p.pos_start[sc] + sample_perm((uint32_t)j, (uint32_t)np, sample_key(key, (uint32_t)sc, 0u));
  else row = p.neg_start[sc] + sample_perm((uint32_t)(j - cp), (uint32_t)nn, sample_key(key, (uint32_t)sc, 1u));
  const float* src = p.data + (size_t)row * p.row_floats;
  for (int c = 0; c < p.G; ++c) p.xyz[(size_t)t * p.G + c] = src[c];
  p.sdf[t] = src[p.G];
}

}  // namespace dsdf
