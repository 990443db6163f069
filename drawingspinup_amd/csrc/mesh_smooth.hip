// Constrained smoothing of the export's binary volume: the weighted-Jacobi iteration of
// mcubes.smooth (MarchingCubeHelper.forward, instant_nsr/models/geometry.py:57-58) on the
// compacted band voxels (gfx950, float64 as PyMCubes).
//
// Unknowns: the nv voxels with |signed distance| <= band radius, compacted in x-major order.
// nbr[6][nv] (int32): slot of the -/+ neighbour along x, y, z, or -1 when that neighbour is
// outside the band (or the volume): it then folds onto the diagonal of the 1-D second difference
//     (F_a v)(i) = cd_a(i) v(i) + v(n-_a(i)) + v(n+_a(i)),   cd_a = -2 + [no n-] + [no n+].
// Energy |F v|^2, Q = sum_a F_a^T F_a.  One iteration (PyMCubes: weight 0.5, projection onto the
// per-voxel bounds lower[i] <= x <= upper[i]: the initial distance on the voxel's own side, 0 for
// the voxels next to the surface, +-inf on the other side) is two passes over the band:
//     y_a = F_a x                                    (smooth_rows_kernel,   3 nv doubles out)
//     x  <- clamp(w * (-(Q x - d x) / d) + (1 - w) x)  with Q x = sum_a F_a^T y_a, d = diag Q
//                                                    (smooth_update_kernel)
// and every tenth iteration the energy x . Q x / 2 as per-workgroup partial sums in a fixed order
// (smooth_energy_kernel; the host adds the partials: the stopping test is deterministic).
// The iteration itself runs as ONE kernel per step (smooth_fused_kernel): the three rows a voxel
// needs per axis — its own and its two neighbours' — are recomputed from x with the very
// expressions of smooth_rows_kernel (same operands, same order: bit-identical values), so y is
// neither written nor read; x ping-pongs between the caller's x and the first nv doubles of y.
// Unique memory per iteration: slots 24 nv + x 8 nv in + 8 nv out + bounds 16 nv = 56 nv bytes
// against ~ (2 x 24 + 24 + 24 + 72 + 16 + 16) nv = 200 nv of the two-pass form; the band of a
// 512^3 export is 2-5 M voxels, i.e. it lives in the Infinity Cache across iterations.
#include "common.h"

namespace {

__device__ __forceinline__ double at_or_zero(const double* __restrict__ v, int s) {
  return s >= 0 ? v[s] : 0.0;
}

__global__ __launch_bounds__(256) void smooth_rows_kernel(const int32_t* __restrict__ nbr, int64_t nv,
                                                          const double* __restrict__ x,
                                                          double* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double xi = x[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int m = nbr[(size_t)(2 * a) * nv + i], p = nbr[(size_t)(2 * a + 1) * nv + i];
      const double cd = -2.0 + (m < 0 ? 1.0 : 0.0) + (p < 0 ? 1.0 : 0.0);
      y[(size_t)a * nv + i] = cd * xi + at_or_zero(x, m) + at_or_zero(x, p);
    }
  }
}

// (Q x)(i) and diag Q (i) from the rows y_a
__device__ __forceinline__ void q_and_diag(const int32_t* __restrict__ nbr, int64_t nv,
                                           const double* __restrict__ y, int64_t i, double& q,
                                           double& d) {
  q = 0.0;
  d = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int m = nbr[(size_t)(2 * a) * nv + i], p = nbr[(size_t)(2 * a + 1) * nv + i];
    const double hm = m < 0 ? 0.0 : 1.0, hp = p < 0 ? 0.0 : 1.0;
    const double cd = -2.0 + (1.0 - hm) + (1.0 - hp);
    const double* ya = y + (size_t)a * nv;
    q += cd * ya[i] + at_or_zero(ya, m) + at_or_zero(ya, p);
    d += cd * cd + hm + hp;
  }
}

__global__ __launch_bounds__(256) void smooth_update_kernel(const int32_t* __restrict__ nbr,
                                                            int64_t nv,
                                                            const double* __restrict__ y,
                                                            const double* __restrict__ lower,
                                                            const double* __restrict__ upper,
                                                            double weight, double* __restrict__ x) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * blockDim.x) {
    double q, d;
    q_and_diag(nbr, nv, y, i, q, d);
    const double xi = x[i];
    const double x1 = -(1.0 / d) * (q - d * xi);                 // -D^-1 R x
    double xn = weight * x1 + (1.0 - weight) * xi;
    xn = fmin(fmax(xn, lower[i]), upper[i]);                     // np.maximum(x, lower); np.minimum(x, upper)
    x[i] = xn;
  }
}

// y_a(j) = cd_a(j) x(j) + x(n-_a(j)) + x(n+_a(j)): smooth_rows_kernel's expression for voxel j
__device__ __forceinline__ double row_of(const int32_t* __restrict__ nbr, int64_t nv,
                                         const double* __restrict__ x, int a, int64_t j) {
  const int m = nbr[(size_t)(2 * a) * nv + j], p = nbr[(size_t)(2 * a + 1) * nv + j];
  const double cd = -2.0 + (m < 0 ? 1.0 : 0.0) + (p < 0 ? 1.0 : 0.0);
  return cd * x[j] + at_or_zero(x, m) + at_or_zero(x, p);
}

// one whole iteration: xo = clamp(w * (-(Q x - d x) / d) + (1 - w) x), Q x from rows recomputed
// on the fly (q_and_diag's sum, with y_a(.) = row_of(.))
__global__ __launch_bounds__(256) void smooth_fused_kernel(const int32_t* __restrict__ nbr, int64_t nv,
                                                           const double* __restrict__ x,
                                                           const double* __restrict__ lower,
                                                           const double* __restrict__ upper,
                                                           double weight, double* __restrict__ xo) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double xi = x[i];
    double q = 0.0, d = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int m = nbr[(size_t)(2 * a) * nv + i], p = nbr[(size_t)(2 * a + 1) * nv + i];
      const double hm = m < 0 ? 0.0 : 1.0, hp = p < 0 ? 0.0 : 1.0;
      const double cd = -2.0 + (1.0 - hm) + (1.0 - hp);
      const double cd_rows = -2.0 + (m < 0 ? 1.0 : 0.0) + (p < 0 ? 1.0 : 0.0);
      const double ya_i = cd_rows * xi + at_or_zero(x, m) + at_or_zero(x, p);
      const double ya_m = m >= 0 ? row_of(nbr, nv, x, a, m) : 0.0;
      const double ya_p = p >= 0 ? row_of(nbr, nv, x, a, p) : 0.0;
      q += cd * ya_i + ya_m + ya_p;
      d += cd * cd + hm + hp;
    }
    const double x1 = -(1.0 / d) * (q - d * xi);                 // -D^-1 R x
    double xn = weight * x1 + (1.0 - weight) * xi;
    xn = fmin(fmax(xn, lower[i]), upper[i]);
    xo[i] = xn;
  }
}

constexpr int EN_BLOCKS = 1024;

__global__ __launch_bounds__(256) void smooth_energy_kernel(const int32_t* __restrict__ nbr,
                                                            int64_t nv,
                                                            const double* __restrict__ y,
                                                            const double* __restrict__ x,
                                                            double* __restrict__ partials) {
  __shared__ double ws[4];
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * blockDim.x) {
    double q, d;
    q_and_diag(nbr, nv, y, i, q, d);
    acc += x[i] * q;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

}  // namespace

extern "C" {

int32_t dsu_smooth_energy_partials(void) { return EN_BLOCKS; }

int dsu_smooth_iterate(const int32_t* nbr, int64_t nv, const double* lower, const double* upper,
                       double weight, int32_t iters, double* x, double* y, void* stream) {
  if (nv < 0 || iters < 0 || (nv && (!nbr || !lower || !upper || !x || !y))) return DSU_EINVAL;
  if (nv == 0 || iters == 0) return DSU_OK;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = dsu_capped_blocks(nv, 256, 8192);
  if (dsu_ab_is("DSU_SMOOTH", "two_pass")) {      // the form of rounds 2-4 (variant builds, A/B)
    for (int it = 0; it < iters; ++it) {
      smooth_rows_kernel<<<dim3(blocks), dim3(256), 0, s>>>(nbr, nv, x, y);
      smooth_update_kernel<<<dim3(blocks), dim3(256), 0, s>>>(nbr, nv, y, lower, upper, weight, x);
    }
    DSU_CHECK_LAUNCH();
    return DSU_OK;
  }
  double* cur = x;
  double* nxt = y;                                  // first nv doubles of the scratch
  for (int it = 0; it < iters; ++it) {
    smooth_fused_kernel<<<dim3(blocks), dim3(256), 0, s>>>(nbr, nv, cur, lower, upper, weight, nxt);
    double* t = cur; cur = nxt; nxt = t;
  }
  DSU_CHECK_LAUNCH();
  if (cur != x &&
      hipMemcpyAsync(x, cur, (size_t)nv * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return DSU_ELAUNCH;
  return DSU_OK;
}

int dsu_smooth_energy(const int32_t* nbr, int64_t nv, const double* x, double* y,
                      double* partials, void* stream) {
  if (nv < 0 || !partials || (nv && (!nbr || !x || !y))) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = dsu_capped_blocks(nv > 0 ? nv : 1, 256, 8192);
  if (nv > 0) smooth_rows_kernel<<<dim3(blocks), dim3(256), 0, s>>>(nbr, nv, x, y);
  smooth_energy_kernel<<<dim3(EN_BLOCKS), dim3(256), 0, s>>>(nbr, nv, y, x, partials);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
