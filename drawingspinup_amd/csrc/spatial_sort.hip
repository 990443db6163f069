// Spatial binning of one optimisation step's sample positions (gfx950).
//
// The marcher hands the geometry network its samples ray by ray (nerfacc.ray_marching order,
// neus.py:119-129), and the rays of a batch are random pixels of six views: 64 neighbouring
// lanes of the hash-grid kernels then touch 64 unrelated neighbourhoods of every level.  Nothing
// in VolumeSDF.forward depends on the order of its points (geometry.py:135-187 is pointwise), so
// the step evaluates them in MORTON order of a 2^bits-per-axis lattice over the contracted unit
// cube instead and scatters the results back through the permutation:
//   * forward: the 8 x 7 x L gathers of a wave fall into a handful of cache lines,
//   * backward: the same-cell run merge (DPP) and the per-workgroup LDS gradient cache see
//     the SAME table entries again and again (one global atomic pair per entry per workgroup
//     instead of one per sample neighbourhood; no cache overflow on the fine levels).
//
// Counting sort: key = Morton(bin) -> returning atomic on the bin counter = rank inside the bin
// -> exclusive scan of the counters -> slot = start[key] + rank.  The order INSIDE a bin is the
// order in which the atomics retired (any order is as good as another for locality); the
// permutation is returned so that the caller can scatter per-point results back.
#include "common.h"

namespace {

constexpr int SCAN_BLOCK = 1024;   // counters per scan workgroup (256 threads x 4)

__device__ __forceinline__ uint32_t part1by2(uint32_t x) {   // spread 10 bits to every third bit
  x &= 0x3FFu;
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__device__ __forceinline__ uint32_t bin_key(const float* __restrict__ pts, int64_t i, float radius,
                                            int bits) {
  const float nb = (float)(1 << bits);
  const float inv = 1.0f / (2.0f * radius);
  uint32_t c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float u = (pts[i * 3 + a] + radius) * inv * nb;
    const float f = fminf(fmaxf(floorf(u), 0.0f), nb - 1.0f);   // NaN -> 0
    c[a] = (uint32_t)f;
  }
  return part1by2(c[0]) | (part1by2(c[1]) << 1) | (part1by2(c[2]) << 2);
}

__global__ __launch_bounds__(256) void bin_count_kernel(const float* __restrict__ pts, int64_t n,
                                                        float radius, int bits,
                                                        uint32_t* __restrict__ counters,
                                                        uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ ranks) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t k = bin_key(pts, i, radius, bits);
    keys[i] = k;
    ranks[i] = atomicAdd(&counters[k], 1u);
  }
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(v, off);
    if (lane >= off) v += o;
  }
  return v;
}

// sums[b] = sum of counters[b*1024 .. b*1024+1023]
__global__ __launch_bounds__(256) void bin_block_sum_kernel(const uint32_t* __restrict__ counters,
                                                            uint32_t* __restrict__ sums) {
  __shared__ uint32_t ws[4];
  const uint4 v = reinterpret_cast<const uint4*>(counters)[blockIdx.x * 256 + threadIdx.x];
  uint32_t s = v.x + v.y + v.z + v.w;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// counters -> exclusive prefix sums (in place); nblocks <= 2048
__global__ __launch_bounds__(256) void bin_scan_kernel(uint32_t* __restrict__ counters,
                                                       const uint32_t* __restrict__ sums) {
  __shared__ uint32_t red[4];
  __shared__ uint32_t wtot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // base = sum of the block sums before this block
  uint32_t b = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) b += sums[t];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) b += __shfl_xor(b, off);
  if (lane == 0) red[wave] = b;
  uint4 v = reinterpret_cast<uint4*>(counters)[blockIdx.x * 256 + threadIdx.x];
  const uint32_t tsum = v.x + v.y + v.z + v.w;
  const uint32_t incl = wave_incl_scan(tsum, lane);
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t base = red[0] + red[1] + red[2] + red[3];
  for (int w = 0; w < wave; ++w) base += wtot[w];
  uint32_t e = base + incl - tsum;
  uint4 o;
  o.x = e; e += v.x;
  o.y = e; e += v.y;
  o.z = e; e += v.z;
  o.w = e;
  reinterpret_cast<uint4*>(counters)[blockIdx.x * 256 + threadIdx.x] = o;
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(const float* __restrict__ pts, int64_t n,
                                                          const uint32_t* __restrict__ starts,
                                                          const uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ ranks,
                                                          int32_t* __restrict__ perm,
                                                          float* __restrict__ pts_sorted) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t slot = starts[keys[i]] + ranks[i];
    perm[slot] = (int32_t)i;
    pts_sorted[(size_t)slot * 3 + 0] = pts[i * 3 + 0];
    pts_sorted[(size_t)slot * 3 + 1] = pts[i * 3 + 1];
    pts_sorted[(size_t)slot * 3 + 2] = pts[i * 3 + 2];
  }
}

inline int64_t n_bins(int bits) { return (int64_t)1 << (3 * bits); }

}  // namespace

extern "C" {

int64_t dsu_spatial_sort_workspace_bytes(int64_t n, int32_t bits) {
  if (n < 0 || bits < 4 || bits > 7) return DSU_EINVAL;
  const int64_t nb = n_bins(bits);
  return (nb + nb / SCAN_BLOCK + 2 * n) * (int64_t)sizeof(uint32_t);
}

int dsu_spatial_sort(const float* pts, int64_t n, float radius, int32_t bits, int32_t* perm,
                     float* pts_sorted, void* workspace, int64_t workspace_bytes, void* stream) {
  if ((!pts && n) || (!perm && n) || (!pts_sorted && n) || n < 0 || n > 0x7FFFFFFF)
    return DSU_EINVAL;
  if (!(radius > 0.0f)) return DSU_EINVAL;
  const int64_t need = dsu_spatial_sort_workspace_bytes(n, bits);
  if (need < 0) return (int)need;
  if (n == 0) return DSU_OK;
  if (!workspace || workspace_bytes < need) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = n_bins(bits);
  const int nblk = (int)(nb / SCAN_BLOCK);
  uint32_t* counters = (uint32_t*)workspace;
  uint32_t* sums = counters + nb;
  uint32_t* keys = sums + nblk;
  uint32_t* ranks = keys + n;
  if (hipMemsetAsync(counters, 0, (size_t)nb * sizeof(uint32_t), s) != hipSuccess)
    return DSU_ELAUNCH;
  const int blocks = dsu_capped_blocks(n, 256, 2048);
  bin_count_kernel<<<dim3(blocks), dim3(256), 0, s>>>(pts, n, radius, bits, counters, keys, ranks);
  bin_block_sum_kernel<<<dim3(nblk), dim3(256), 0, s>>>(counters, sums);
  bin_scan_kernel<<<dim3(nblk), dim3(256), 0, s>>>(counters, sums);
  bin_scatter_kernel<<<dim3(blocks), dim3(256), 0, s>>>(pts, n, counters, keys, ranks, perm,
                                                        pts_sorted);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
