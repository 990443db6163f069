// Implicit umbrella-Laplacian smoothing of the export mesh on the device:
// trimesh.smoothing.filter_laplacian(mesh, lamb = 2, iterations = 5, implicit_time_integration =
// True), called by save_mesh (instant_nsr/utils/mesh_utils.py:42-45).  Each of its iterations solves
//     (I + lamb (I - L)) V' = V,        L = umbrella operator, row i = 1 / deg(i) on the neighbours of i
// which the reference (and nsr/mesh.laplacian_smooth_implicit on host arrays) hands to a sparse LU.
// The matrix is strictly diagonally dominant by rows — diagonal 1 + lamb against off-diagonal row
// sum lamb — so the Jacobi iteration
//     V'_{k+1}(i) = (V(i) + lamb * mean_{j in N(i)} V'_k(j)) / (1 + lamb)
// contracts the error by lamb / (1 + lamb) = 2/3 per sweep in the max norm: 96 sweeps from the
// starting guess V' = V leave 1e-17 of the initial difference, below float64 rounding of the
// direct solve.  float64 throughout; one launch per sweep (the sweep needs every vertex of the
// previous one), ping-pong between two buffers; 25 k vertices: ~2 us per sweep.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void umbrella_jacobi_kernel(const int32_t* __restrict__ off,
                                                              const int32_t* __restrict__ nbr, int64_t n,
                                                              double lamb, double inv_diag,
                                                              const double* __restrict__ b,
                                                              const double* __restrict__ x,
                                                              double* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = off[i], e = off[i + 1];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int32_t k = s; k < e; ++k) {                       // fixed (sorted) neighbour order
      const int64_t j = nbr[k];
      a0 += x[3 * j]; a1 += x[3 * j + 1]; a2 += x[3 * j + 2];
    }
    const double w = e > s ? lamb / (double)(e - s) : 0.0;
    y[3 * i] = (b[3 * i] + w * a0) * inv_diag;
    y[3 * i + 1] = (b[3 * i + 1] + w * a1) * inv_diag;
    y[3 * i + 2] = (b[3 * i + 2] + w * a2) * inv_diag;
  }
}

}  // namespace

extern "C" {

int dsu_umbrella_implicit_solve(const int32_t* offsets, const int32_t* neighbours, int64_t n_verts,
                                double lamb, const double* rhs, double* x, double* tmp, int32_t sweeps,
                                void* stream) {
  if (n_verts < 0 || !(lamb >= 0.0) || sweeps < 0 || (sweeps & 1) ||
      (n_verts && (!offsets || !neighbours || !rhs || !x || !tmp)))
    return DSU_EINVAL;
  if (n_verts == 0 || sweeps == 0) return DSU_OK;
  hipStream_t s = (hipStream_t)stream;
  const double inv_diag = 1.0 / (1.0 + lamb);
  const int blocks = dsu_capped_blocks(n_verts, 256);
  double* cur = x;
  double* nxt = tmp;
  for (int k = 0; k < sweeps; ++k) {                        // even count: the result ends in `x`
    umbrella_jacobi_kernel<<<blocks, 256, 0, s>>>(offsets, neighbours, n_verts, lamb, inv_diag, rhs, cur, nxt);
    double* t = cur; cur = nxt; nxt = t;
  }
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
