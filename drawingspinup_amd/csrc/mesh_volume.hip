// Dense-volume passes of the NSR export (gfx950): byte / integer work over the 512^3 lattice that the
// export ran as tensor programs before (nsr/mesh.py keeps those forms for host tensors and as the
// tests' comparison):
//   * dsu_volume_band_distance — the signed Euclidean distance transform in a band that
//     mcubes.smooth starts from (MarchingCubeHelper.forward, instant_nsr/models/geometry.py:57-58;
//     scipy.ndimage.distance_transform_edt on both classes restricted to the radius that matters):
//     exact wherever the other class is within `R` voxels.  Separable squared-distance transform in
//     integers: nearest other-class voxel along z (row in LDS), then min-plus sweeps over
//     |dy| <= R and |dx| <= R with the 2R+1 window of a line in registers (every voxel is read
//     once per sweep).  Both classes travel together as two bytes per voxel.  The last sweep
//     writes the float64 volume through a (2, cap2+1) table the caller forms on the host
//     (sqrt, far-field value, +-0.5, sign) and a byte mask of the band.
//     Traffic: 1 + 2 + 2 + 2 + 2 + 8 + 1 bytes per voxel (2.4 GB for 512^3) against ~60 passes over
//     4..8-byte volumes of the tensor-program form.
//   * dsu_mc_cube_index — marching cubes' per-cube configuration byte (marchingcubes.h of PyMCubes:
//     `if (v[m] <= isovalue) cubeindex |= 1 << m`, corner numbering of nsr/mesh.py) from the
//     float64 volume: one byte per cube instead of eight shifted int64 volumes.
#include "common.h"

namespace {

constexpr int VB_MAX_Z = 4096;
constexpr int VB_INF = 255;                    // "no such voxel on this line"

// pass 1: dz^2 (capped) to the nearest True (low byte) / False (high byte) voxel along z
__global__ __launch_bounds__(256) void band_z_kernel(const uint8_t* __restrict__ b, int64_t rows,
                                                     int Z, int R1, uint16_t* __restrict__ out) {
  __shared__ uint8_t row[VB_MAX_Z];
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const uint8_t* src = b + r * Z;
    for (int z = threadIdx.x; z < Z; z += blockDim.x) row[z] = src[z] ? 1 : 0;
    __syncthreads();
    for (int z = threadIdx.x; z < Z; z += blockDim.x) {
      int dt = R1, df = R1;                    // capped at R + 1
      for (int s = 0; s < R1; ++s) {
        const int lo = z - s, hi = z + s;
        bool t = false, f = false;
        if (lo >= 0) { t |= row[lo] != 0; f |= row[lo] == 0; }
        if (hi < Z) { t |= row[hi] != 0; f |= row[hi] == 0; }
        if (t && s < dt) dt = s;
        if (f && s < df) df = s;
      }
      out[r * Z + z] = (uint16_t)((dt * dt) | ((df * df) << 8));
    }
    __syncthreads();
  }
}

// passes 2, 3: out[i] = min(cap2, min_{|s| <= R} in[i + s] + s^2) along a strided line, for both
// bytes.  One thread per line; lanes run along the contiguous axis.  FINAL: instead of the packed
// pair, the float64 value table[class][d2 of the OTHER class's nearest voxel] and the band byte.
template <int R, bool FINAL>
__global__ __launch_bounds__(256) void band_line_kernel(
    const uint16_t* __restrict__ in, int64_t n_a, int64_t stride_a, int n_line, int64_t stride_line,
    int n_c, int cap2, uint16_t* __restrict__ out, const uint8_t* __restrict__ b,
    const double* __restrict__ table, const uint8_t* __restrict__ band_of, double* __restrict__ dist,
    uint8_t* __restrict__ band) {
  constexpr int W = 2 * R + 1;
  const int64_t cols = n_a * (int64_t)n_c;
  for (int64_t col = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; col < cols;
       col += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = col / n_c, c = col - a * n_c;
    const int64_t base = a * stride_a + c;
    int wt[W], wf[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const int i = k - R;                     // window of output 0: inputs -R .. R
      int t = VB_INF, f = VB_INF;
      if (i >= 0 && i < n_line) {
        const uint16_t v = in[base + (int64_t)i * stride_line];
        t = v & 255;
        f = v >> 8;
      }
      wt[k] = t;
      wf[k] = f;
    }
    // the value entering the window at output i + 1 is input i + 1 + R: fetched CH outputs ahead
    constexpr int CH = 8;
    for (int i0 = 0; i0 < n_line; i0 += CH) {
      uint16_t nx[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int j = i0 + u + 1 + R;
        nx[u] = j < n_line ? in[base + (int64_t)j * stride_line] : (uint16_t)0xffff;
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int i = i0 + u;
        if (i < n_line) {
          int mt = cap2, mf = cap2;
#pragma unroll
          for (int k = 0; k < W; ++k) {
            const int s2 = (k - R) * (k - R);
            mt = min(mt, wt[k] + s2);
            mf = min(mf, wf[k] + s2);
          }
          const int64_t o = base + (int64_t)i * stride_line;
          if (FINAL) {
            const bool inside = b[o] != 0;
            const int d2 = inside ? mf : mt;   // distance to the nearest voxel of the OTHER class
            dist[o] = table[(inside ? 0 : cap2 + 1) + d2];
            band[o] = band_of[d2];
          } else {
            out[o] = (uint16_t)(mt | (mf << 8));
          }
        }
#pragma unroll
        for (int k = 0; k + 1 < W; ++k) {
          wt[k] = wt[k + 1];
          wf[k] = wf[k + 1];
        }
        wt[W - 1] = nx[u] == 0xffff ? VB_INF : (nx[u] & 255);
        wf[W - 1] = nx[u] == 0xffff ? VB_INF : (nx[u] >> 8);
      }
    }
  }
}

template <int R>
int band_lines(const uint16_t* in, int X, int Y, int Z, int axis, int cap2, uint16_t* out, const uint8_t* b,
               const double* table, const uint8_t* band_of, double* dist, uint8_t* band, hipStream_t s) {
  const int64_t n_a = axis == 1 ? X : Y;
  const int64_t stride_a = axis == 1 ? (int64_t)Y * Z : Z;
  const int n_line = axis == 1 ? Y : X;
  const int64_t stride_line = axis == 1 ? Z : (int64_t)Y * Z;
  const int blocks = dsu_capped_blocks(n_a * Z, 256, 1 << 20);
  if (dist)
    band_line_kernel<R, true><<<dim3(blocks), dim3(256), 0, s>>>(in, n_a, stride_a, n_line, stride_line, Z,
                                                                 cap2, out, b, table, band_of, dist, band);
  else
    band_line_kernel<R, false><<<dim3(blocks), dim3(256), 0, s>>>(in, n_a, stride_a, n_line, stride_line, Z,
                                                                  cap2, out, b, table, band_of, dist, band);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

__global__ __launch_bounds__(256) void mc_cube_index_kernel(const double* __restrict__ v, int X, int Y,
                                                            int Z, double iso,
                                                            uint8_t* __restrict__ cube) {
  const int64_t cz = Z - 1, cy = Y - 1;
  const int64_t n = (int64_t)(X - 1) * cy * cz;
  const int64_t sy = Z, sx = (int64_t)Y * Z;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i % cz, j = (i / cz) % cy, x = i / (cz * cy);
    const double* p = v + x * sx + j * sy + k;
    // corners (Bourke / PyMCubes numbering): (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) (1,0,1) (1,1,1) (0,1,1)
    uint32_t c = 0;
    c |= (p[0] <= iso) ? 1u : 0u;
    c |= (p[sx] <= iso) ? 2u : 0u;
    c |= (p[sx + sy] <= iso) ? 4u : 0u;
    c |= (p[sy] <= iso) ? 8u : 0u;
    c |= (p[1] <= iso) ? 16u : 0u;
    c |= (p[sx + 1] <= iso) ? 32u : 0u;
    c |= (p[sx + sy + 1] <= iso) ? 64u : 0u;
    c |= (p[sy + 1] <= iso) ? 128u : 0u;
    cube[i] = (uint8_t)c;
  }
}

}  // namespace

extern "C" {

int64_t dsu_volume_band_distance_workspace_bytes(int32_t X, int32_t Y, int32_t Z) {
  if (X < 1 || Y < 1 || Z < 1) return DSU_EINVAL;
  return 2 * (int64_t)X * Y * Z * (int64_t)sizeof(uint16_t);
}

int dsu_volume_band_distance(const uint8_t* binary, int32_t X, int32_t Y, int32_t Z, int32_t R,
                             const double* value_table, const uint8_t* band_table, double* dist,
                             uint8_t* band, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!binary || !value_table || !band_table || !dist || !band || !workspace || X < 1 || Y < 1 || Z < 1)
    return DSU_EINVAL;
  if (Z > VB_MAX_Z || R < 1 || R > 8) return DSU_EUNSUP;      // (R + 1)^2 + R^2 < 255
  if (workspace_bytes < dsu_volume_band_distance_workspace_bytes(X, Y, Z)) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nvox = (int64_t)X * Y * Z;
  uint16_t* p0 = (uint16_t*)workspace;
  uint16_t* p1 = p0 + nvox;
  const int cap2 = (R + 1) * (R + 1);
  const int64_t rows = (int64_t)X * Y;
  band_z_kernel<<<dim3((unsigned)(rows < (1 << 20) ? rows : (1 << 20))), dim3(256), 0, s>>>(binary, rows, Z, R + 1, p0);
  DSU_CHECK_LAUNCH();
#define DSU_BAND_R(RR)                                                                               \
  case RR: {                                                                                         \
    int rc = band_lines<RR>(p0, X, Y, Z, 1, cap2, p1, nullptr, nullptr, nullptr, nullptr, nullptr, s); \
    if (rc) return rc;                                                                               \
    return band_lines<RR>(p1, X, Y, Z, 0, cap2, nullptr, binary, value_table, band_table, dist, band, s); \
  }
  switch (R) {
    DSU_BAND_R(1) DSU_BAND_R(2) DSU_BAND_R(3) DSU_BAND_R(4) DSU_BAND_R(5) DSU_BAND_R(6) DSU_BAND_R(7)
    DSU_BAND_R(8)
  }
#undef DSU_BAND_R
  return DSU_EUNSUP;
}

int dsu_mc_cube_index(const double* volume, int32_t X, int32_t Y, int32_t Z, double isovalue,
                      uint8_t* cube, void* stream) {
  if (!volume || !cube || X < 2 || Y < 2 || Z < 2) return DSU_EINVAL;
  const int64_t n = (int64_t)(X - 1) * (Y - 1) * (Z - 1);
  mc_cube_index_kernel<<<dim3(dsu_capped_blocks(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream>>>(
      volume, X, Y, Z, isovalue, cube);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
