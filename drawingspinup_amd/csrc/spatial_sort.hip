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

// n_dev != nullptr: the point count lives on the device (n = n_dev[0] + n, capped at n_cap) — the
// prefetch path sorts on the side stream before the host has seen the sample total
__device__ __forceinline__ int64_t resolve_n(const int32_t* n_dev, int64_t n, int64_t n_cap) {
  if (n_dev) n += n_dev[0];
  return n < n_cap ? n : n_cap;
}

__global__ __launch_bounds__(256) void bin_count_kernel(const float* __restrict__ pts, int64_t n,
                                                        float radius, int bits,
                                                        uint32_t* __restrict__ counters,
                                                        uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ ranks,
                                                        const int32_t* __restrict__ n_dev,
                                                        int64_t n_cap) {
  n = resolve_n(n_dev, n, n_cap);
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
                                                          float* __restrict__ pts_sorted,
                                                          const int32_t* __restrict__ n_dev,
                                                          int64_t n_cap) {
  n = resolve_n(n_dev, n, n_cap);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t slot = starts[keys[i]] + ranks[i];
    perm[slot] = (int32_t)i;
    pts_sorted[(size_t)slot * 3 + 0] = pts[i * 3 + 0];
    pts_sorted[(size_t)slot * 3 + 1] = pts[i * 3 + 1];
    pts_sorted[(size_t)slot * 3 + 2] = pts[i * 3 + 2];
  }
}

// rows [total, total + n_r) = pts_random, [total + n_r, total + 2 n_r) = pts_random + alpha * perturb
// (neus.py:155-162: the random points of the sparsity / smoothness terms ride in the same
// geometry launch as the ray samples)
__global__ __launch_bounds__(256) void points_tail_kernel(float* __restrict__ points, int64_t cap_rows,
                                                          const int32_t* __restrict__ total_dev,
                                                          const float* __restrict__ pr,
                                                          const float* __restrict__ pe, int64_t n_r,
                                                          float alpha) {
  const int64_t total = total_dev[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_r * 3;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / 3;
    const float v = pr[i];
    if (total + row < cap_rows) points[(total) * 3 + i] = v;
    if (total + n_r + row < cap_rows) points[(total + n_r) * 3 + i] = v + alpha * pe[i];
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

int dsu_points_tail(float* points, int64_t capacity_rows, const int32_t* total_dev,
                    const float* pts_random, const float* perturb, int64_t n_random, float alpha,
                    void* stream) {
  if (!points || !total_dev || n_random < 0 || capacity_rows < 0 ||
      (n_random && (!pts_random || !perturb)))
    return DSU_EINVAL;
  if (n_random == 0) return DSU_OK;
  points_tail_kernel<<<dsu_capped_blocks(n_random * 3, 256, 256), 256, 0, (hipStream_t)stream>>>(
      points, capacity_rows, total_dev, pts_random, perturb, n_random, alpha);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

static int spatial_sort_impl(const float* pts, int64_t n, const int32_t* n_dev, int64_t n_cap,
                             float radius, int32_t bits, int32_t* perm, float* pts_sorted,
                             void* workspace, int64_t workspace_bytes, void* stream);

int dsu_spatial_sort(const float* pts, int64_t n, float radius, int32_t bits, int32_t* perm,
                     float* pts_sorted, void* workspace, int64_t workspace_bytes, void* stream) {
  return spatial_sort_impl(pts, n, nullptr, n, radius, bits, perm, pts_sorted, workspace,
                           workspace_bytes, stream);
}

int dsu_spatial_sort_dev(const float* pts, int64_t n_capacity, const int32_t* n_dev, int64_t n_add,
                         float radius, int32_t bits, int32_t* perm, float* pts_sorted,
                         void* workspace, int64_t workspace_bytes, void* stream) {
  if (!n_dev || n_add < 0) return DSU_EINVAL;
  return spatial_sort_impl(pts, n_add, n_dev, n_capacity, radius, bits, perm, pts_sorted, workspace,
                           workspace_bytes, stream);
}

static int spatial_sort_impl(const float* pts, int64_t n_host, const int32_t* n_dev, int64_t n,
                             float radius, int32_t bits, int32_t* perm, float* pts_sorted,
                             void* workspace, int64_t workspace_bytes, void* stream) {
  // n = capacity (the exact count when n_dev is null)
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
  bin_count_kernel<<<dim3(blocks), dim3(256), 0, s>>>(pts, n_host, radius, bits, counters, keys, ranks,
                                                      n_dev, n);
  bin_block_sum_kernel<<<dim3(nblk), dim3(256), 0, s>>>(counters, sums);
  bin_scan_kernel<<<dim3(nblk), dim3(256), 0, s>>>(counters, sums);
  bin_scatter_kernel<<<dim3(blocks), dim3(256), 0, s>>>(pts, n_host, counters, keys, ranks, perm,
                                                        pts_sorted, n_dev, n);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
