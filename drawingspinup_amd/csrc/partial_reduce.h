// Deferred reduction of per-workgroup partial gradient vectors: a backward kernel that keeps its
// parameter gradients in registers writes ONE partial vector per workgroup; summing them used to be
// a launch of its own per kernel (~7 us each, latency only).  The sums are now taken by extra
// workgroups of a launch that follows anyway (the table-gradient scatter of the geometry
// backward), described by this record (include/dsu_hip.h: dsu_partial_reduce).
#pragma once
#include "common.h"

namespace dsu_red {

// 64 elements per workgroup; the summation order of the stand-alone reduction kernels is kept
// (16 interleaved slices of the workgroup range, then the slices in order), whatever the block size.
__device__ __forceinline__ void reduce_block(const dsu_partial_reduce& r, int blk,
                                             float* red /* LDS, 16 * 64 floats */) {
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6, ng = blockDim.x >> 6;
  const int v = blk * 64 + e;
  const bool in_range = v < r.n;
  for (int sl = g; sl < 16; sl += ng) {
    float acc = 0.0f;
    if (in_range)
      for (int b = sl; b < r.nblocks; b += 16) acc += r.partials[(size_t)b * r.stride + v];
    red[sl * 64 + e] = acc;
  }
  __syncthreads();
  if (g != 0 || !in_range) return;
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k * 64 + e];
  const int32_t d = r.map[v];
  if (d >= 0) r.base[d] += s;
}

__host__ __device__ inline int blocks_of(const dsu_partial_reduce& r) { return (r.n + 63) / 64; }

}  // namespace dsu_red
