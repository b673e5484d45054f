// common.hpp -- shared device helpers for libdsdf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 32-bit avalanche mix; specification: oracle/deepsdf_oracle.py _lowbias32
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// dropout keep bit of element (grow, col); ck = lowbias32(col * 0x85EBCA77 + key) is per column.
// One hash serves rows 2q and 2q+1 (low / high 16 bits).  Spec: oracle dropout_keep.
__device__ __forceinline__ uint32_t drop_col_key(uint32_t col, uint32_t key) {
  return lowbias32(col * 0x85EBCA77u + key);
}
__device__ __forceinline__ uint32_t drop_pair_hash(uint32_t ck, uint32_t grow) {
  return lowbias32(ck ^ ((grow >> 1) * 0x9E3779B1u));
}
__device__ __forceinline__ bool drop_keep(uint32_t h, uint32_t grow, uint32_t thr16) {
  const uint32_t bits = (grow & 1u) ? (h >> 16) : (h & 0xFFFFu);
  return bits >= thr16;
}

// Blocks b and b+8 share an XCD (round-robin dispatch, observed; speed only).  Give every XCD a
// contiguous range of logical ids so that neighbouring tiles (which share operand panels) hit one L2.
// Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7, s = b >> 3;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + s;
}

// Pull this kernel's own instructions into L2 as DATA at its very start (one 64-byte line per lane-load, results unused).
// The big kernels evict each other's code from the instruction cache AND from L2 (hundreds of MB stream through between
// two launches), and a lone wave per SIMD has nothing to hide an instruction-fetch miss behind; a miss that hits L2
// costs a fraction of one that goes to HBM.  `bytes` is clamped to the text that really follows (dsdf_text_end_marker).
__device__ __noinline__ void dsdf_text_end_marker();   // defined LAST in dsdf_api.hip: an address inside .text behind every kernel
__device__ __forceinline__ uint32_t warm_own_code(int bytes) {
  uint64_t pc;
  asm volatile("s_getpc_b64 %0" : "=s"(pc));
  const char* base = reinterpret_cast<const char*>(pc & ~63ull);
  // never read past the code object's text: clamp to the marker function (if the compiler ever places it in front of this
  // kernel the room is negative and nothing is read)
  const long long room = reinterpret_cast<const char*>(reinterpret_cast<void*>(&dsdf_text_end_marker)) - base;
  if (bytes > room) bytes = room > 0 ? (int)room : 0;
  uint32_t acc = 0;
  for (int off = (int)threadIdx.x * 64; off < bytes; off += 256 * 64) acc ^= *reinterpret_cast<const uint32_t*>(base + off);
  return acc;   // the caller keeps it alive (e.g. folds it into a store that never happens)
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sums of SIXTEEN per-lane values over the 64 lanes of a wave with 17 shuffles instead of 16 x 6: in every step a lane keeps the half
// of the values its lane-id bit selects and hands the other half to its partner.  Every lane returns the total of value index
//   wave_sum16_index(lane) = 8 b5 + 4 b4 + 2 b3 + b2          (b_i = bit i of the lane id)
// -- four lanes per index; lane wave_sum16_lane(q) is one that holds value q's total.  Fixed order (not wave_sum's).
__device__ __forceinline__ int wave_sum16_index(int lane) { return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1); }
__device__ __forceinline__ constexpr int wave_sum16_lane(int q) { return 32 * ((q >> 3) & 1) + 16 * ((q >> 2) & 1) + 8 * ((q >> 1) & 1) + 4 * (q & 1); }
__device__ __forceinline__ float wave_sum16(const float (&a)[16], int lane) {
  float b8[8], b4[4], b2[2];
  const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
  for (int j = 0; j < 8; ++j) b8[j] = (h5 ? a[8 + j] : a[j]) + __shfl_xor(h5 ? a[j] : a[8 + j], 32, 64);
#pragma unroll
  for (int j = 0; j < 4; ++j) b4[j] = (h4 ? b8[4 + j] : b8[j]) + __shfl_xor(h4 ? b8[j] : b8[4 + j], 16, 64);
#pragma unroll
  for (int j = 0; j < 2; ++j) b2[j] = (h3 ? b4[2 + j] : b4[j]) + __shfl_xor(h3 ? b4[j] : b4[2 + j], 8, 64);
  float r = (h2 ? b2[1] : b2[0]) + __shfl_xor(h2 ? b2[0] : b2[1], 4, 64);
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;
}

// deterministic block-wide sum for 256-thread blocks; result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats of LDS */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ void zero_tail4(float4& v, int rem) {  // keep the first `rem` elements
  if (rem < 4) v.w = 0.f;
  if (rem < 3) v.z = 0.f;
  if (rem < 2) v.y = 0.f;
  if (rem < 1) v.x = 0.f;
}

}  // namespace dsdf
