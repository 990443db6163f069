// Mesh post-processing of the NSR export: the geometric queries behind
// color_projection (2_charactor_reconstructor/instant_nsr/utils/coloring_utils.py:91-138) and
// get_offset_mask (instant_nsr/utils/thinning_utils.py:96-193) on the device (gfx950).
//
// The reference answers them with three CPU / third-party pieces:
//   * mesh_raycast.raycast(source, (0,0,+-1), mesh=triangles) once per vertex in Python loops —
//     every ray is parallel to z, so the query is two-dimensional: which triangles cover (x, y),
//     and at which z.  Here: triangles are binned on a uniform xy grid (counting sort, two
//     kernels around the caller's prefix sum) and a ray only visits its cell's list.
//   * pytorch3d's MeshRasterizer used as a silhouette renderer (MaskRenderer.render:
//     zbuf > -1) — one kernel that marks the pixel centres every triangle covers.
//   * scipy cKDTree k-nearest neighbours in xy (interpolate_rgb) — the same grid, searched ring
//     by ring with an exact termination test.
// plus cv2.erode with the 19x19 elliptic element (load_color).
//
// Ray / triangle rule (one definition for both callers; mesh_raycast's source is not in the
// snapshot): the ray hits a triangle when (x, y) lies inside or on the boundary of its xy
// projection (edge functions in float64, degenerate projections skipped) and the hit is not behind
// the origin, t = (z_hit - z_origin) * sign >= 0.  A ray that starts AT a mesh vertex
// (self_vertex >= 0) meets the triangles incident to that vertex at distance exactly 0 — the
// reference relies on that (`farthest_result['distance'] == 0` = "nothing in front of me").
#include "common.h"

namespace {

struct ZGrid {
  float x0, y0, inv_cell;
  int32_t g;   // cells per axis
};

__device__ __forceinline__ int cell_of(float v, float v0, float inv_cell, int g) {
  int c = (int)floorf((v - v0) * inv_cell);
  return min(max(c, 0), g - 1);
}

// MODE 0: counts[cell] += 1 per overlapped cell; MODE 1: items[offsets[cell] + cursor[cell]++] = face
template <int MODE>
__global__ __launch_bounds__(256) void zgrid_bin_kernel(const float* __restrict__ tris, int64_t nf,
                                                        ZGrid gr, int32_t* __restrict__ counts,
                                                        const int32_t* __restrict__ offsets,
                                                        int32_t* __restrict__ items) {
  const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const float* t = tris + f * 9;
  const float xmin = fminf(fminf(t[0], t[3]), t[6]), xmax = fmaxf(fmaxf(t[0], t[3]), t[6]);
  const float ymin = fminf(fminf(t[1], t[4]), t[7]), ymax = fmaxf(fmaxf(t[1], t[4]), t[7]);
  const int cx0 = cell_of(xmin, gr.x0, gr.inv_cell, gr.g), cx1 = cell_of(xmax, gr.x0, gr.inv_cell, gr.g);
  const int cy0 = cell_of(ymin, gr.y0, gr.inv_cell, gr.g), cy1 = cell_of(ymax, gr.y0, gr.inv_cell, gr.g);
  for (int cy = cy0; cy <= cy1; ++cy)
    for (int cx = cx0; cx <= cx1; ++cx) {
      const int c = cy * gr.g + cx;
      const int k = atomicAdd(&counts[c], 1);
      if (MODE == 1) items[offsets[c] + k] = (int32_t)f;
    }
}

// one ray per thread
__global__ __launch_bounds__(256) void zray_cast_kernel(
    const float* __restrict__ tris, const int32_t* __restrict__ faces, ZGrid gr,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ items,
    const float* __restrict__ origins, int64_t n, float sign, const int32_t* __restrict__ self_vertex,
    int32_t* __restrict__ out_count, float* __restrict__ out_tmin, int32_t* __restrict__ out_fmin,
    float* __restrict__ out_tmax, int32_t* __restrict__ out_fmax) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float ox = origins[i * 3], oy = origins[i * 3 + 1], oz = origins[i * 3 + 2];
  const int sv = self_vertex ? self_vertex[i] : -1;
  const int c = cell_of(oy, gr.y0, gr.inv_cell, gr.g) * gr.g + cell_of(ox, gr.x0, gr.inv_cell, gr.g);
  int count = 0, fmin_ = -1, fmax_ = -1;
  float tmin_ = INFINITY, tmax_ = -INFINITY;
  for (int k = offsets[c]; k < offsets[c + 1]; ++k) {
    const int f = items[k];
    const float* t = tris + (int64_t)f * 9;
    float th;
    if (sv >= 0 && (faces[f * 3] == sv || faces[f * 3 + 1] == sv || faces[f * 3 + 2] == sv)) {
      th = 0.0f;                               // incident triangle: met at the origin itself
    } else {
      const double px = ox, py = oy;
      const double ax = t[0], ay = t[1], bx = t[3], by = t[4], cx = t[6], cy = t[7];
      const double w0 = (px - bx) * (cy - by) - (py - by) * (cx - bx);   // edge b->c, weight of a
      const double w1 = (px - cx) * (ay - cy) - (py - cy) * (ax - cx);   // edge c->a, weight of b
      const double w2 = (px - ax) * (by - ay) - (py - ay) * (bx - ax);   // edge a->b, weight of c
      const double area = w0 + w1 + w2;
      if (area == 0.0) continue;
      const bool inside = (w0 >= 0.0 && w1 >= 0.0 && w2 >= 0.0) || (w0 <= 0.0 && w1 <= 0.0 && w2 <= 0.0);
      if (!inside) continue;
      const double z = (w0 * (double)t[2] + w1 * (double)t[5] + w2 * (double)t[8]) / area;
      th = (float)((z - (double)oz) * (double)sign);
      if (!(th >= 0.0f)) continue;
    }
    ++count;
    if (th < tmin_ || (th == tmin_ && f < fmin_)) { tmin_ = th; fmin_ = f; }
    if (th > tmax_ || (th == tmax_ && f < fmax_)) { tmax_ = th; fmax_ = f; }
  }
  out_count[i] = count;
  out_tmin[i] = count ? tmin_ : 0.0f;
  out_tmax[i] = count ? tmax_ : 0.0f;
  out_fmin[i] = fmin_;
  out_fmax[i] = fmax_;
}

// MaskRenderer.render (coloring_utils.py:22-41): orthographic silhouette of the mesh seen from
// +z, `res` x `res`, pixel (row r, col c) centre at x = (2c + 1) / res - 1, y = 1 - (2r + 1) / res
// (pytorch3d: NDC +X left / +Y up, camera from look_at_view_transform(1, 0, 0) mirrors x back).
// One thread per triangle walks its pixel bounding box; writes are idempotent (255).
__global__ __launch_bounds__(256) void raster_mask_kernel(const float* __restrict__ tris,
                                                          int64_t nf, float scale, int32_t res,
                                                          uint8_t* __restrict__ mask) {
  const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const float* t = tris + f * 9;
  const double ax = (double)t[0] * scale, ay = (double)t[1] * scale;
  const double bx = (double)t[3] * scale, by = (double)t[4] * scale;
  const double cx = (double)t[6] * scale, cy = (double)t[7] * scale;
  const double area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
  if (area == 0.0) return;
  const double xmin = fmin(fmin(ax, bx), cx), xmax = fmax(fmax(ax, bx), cx);
  const double ymin = fmin(fmin(ay, by), cy), ymax = fmax(fmax(ay, by), cy);
  // x = (2c + 1) / res - 1  ->  c = ((x + 1) res - 1) / 2 ;  y = 1 - (2r + 1) / res  ->  r = ((1 - y) res - 1) / 2
  int c0 = (int)ceil(((xmin + 1.0) * res - 1.0) * 0.5), c1 = (int)floor(((xmax + 1.0) * res - 1.0) * 0.5);
  int r0 = (int)ceil(((1.0 - ymax) * res - 1.0) * 0.5), r1 = (int)floor(((1.0 - ymin) * res - 1.0) * 0.5);
  c0 = max(c0, 0); r0 = max(r0, 0); c1 = min(c1, res - 1); r1 = min(r1, res - 1);
  for (int r = r0; r <= r1; ++r) {
    const double py = 1.0 - (2.0 * r + 1.0) / res;
    for (int c = c0; c <= c1; ++c) {
      const double px = (2.0 * c + 1.0) / res - 1.0;
      const double w0 = (px - bx) * (cy - by) - (py - by) * (cx - bx);
      const double w1 = (px - cx) * (ay - cy) - (py - cy) * (ax - cx);
      const double w2 = (px - ax) * (by - ay) - (py - ay) * (bx - ax);
      if ((w0 >= 0.0 && w1 >= 0.0 && w2 >= 0.0) || (w0 <= 0.0 && w1 <= 0.0 && w2 <= 0.0))
        mask[(size_t)r * res + c] = 255;
    }
  }
}

// cv2.erode(mask, getStructuringElement(MORPH_ELLIPSE, (k, k)), iterations=1): minimum over the
// element's pixels; OpenCV's default border for erosion is +infinity (outside never lowers the
// minimum).  Element row i covers columns [c - dx, c + dx] with dx = round(c * sqrt(1 - dy^2 / r^2)),
// r = c = k / 2 (OpenCV's getStructuringElement).
__global__ __launch_bounds__(256) void erode_ellipse_kernel(const uint8_t* __restrict__ src, int32_t H,
                                                            int32_t W, int32_t k,
                                                            uint8_t* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int r = k / 2;
  const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
  int m = 255;
  for (int i = 0; i < k; ++i) {
    const int dy = i - r, yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    const int dx = (int)rint((double)r * sqrt(((double)r * r - (double)dy * dy) * inv_r2));
    const int xa = max(x - dx, 0), xb = min(x + dx, W - 1);
    for (int xx = xa; xx <= xb; ++xx) m = min(m, (int)src[(size_t)yy * W + xx]);
  }
  dst[(size_t)y * W + x] = (uint8_t)m;
}

// interpolate_rgb (coloring_utils.py:43-58): the K nearest known points in the xy plane (scipy
// cKDTree.query(k = 8)), weights 1 / (d + 1e-6) normalised, colours blended in float64.  Known
// points sit in the same kind of xy grid (point bins, assigned from the float32 cast of the
// coordinates; distances in float64 like scipy: front / back vertices of a flat mesh are almost
// coincident in xy and the weights 1 / (d + 1e-6) feel a float32 rounding of d); rings of cells are visited outward until
// the K-th distance found cannot be beaten by anything outside the visited square.
constexpr int KNN_K = 8;
__global__ __launch_bounds__(128) void knn_blend_kernel(const double* __restrict__ query, int64_t nq,
                                                        const double* __restrict__ known_xy,
                                                        const float* __restrict__ known_rgb,
                                                        int64_t nk, ZGrid gr,
                                                        const int32_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ items,
                                                        float* __restrict__ out_rgb) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const double qx = query[i * 2], qy = query[i * 2 + 1];
  const int cx = cell_of((float)qx, gr.x0, gr.inv_cell, gr.g), cy = cell_of((float)qy, gr.y0, gr.inv_cell, gr.g);
  double bd[KNN_K];
  int bi[KNN_K];
#pragma unroll
  for (int k = 0; k < KNN_K; ++k) { bd[k] = INFINITY; bi[k] = -1; }
  const double cell = 1.0 / (double)gr.inv_cell;
  const int kk = nk < KNN_K ? (int)nk : KNN_K;
  for (int ring = 0; ring < gr.g; ++ring) {
    for (int yy = cy - ring; yy <= cy + ring; ++yy) {
      if (yy < 0 || yy >= gr.g) continue;
      const bool edge_row = yy == cy - ring || yy == cy + ring;
      for (int xx = cx - ring; xx <= cx + ring; xx += (edge_row ? 1 : 2 * ring > 0 ? 2 * ring : 1)) {
        if (xx < 0 || xx >= gr.g) continue;
        const int c = yy * gr.g + xx;
        for (int p = offsets[c]; p < offsets[c + 1]; ++p) {
          const int j = items[p];
          const double dx = known_xy[j * 2] - qx, dy = known_xy[j * 2 + 1] - qy;
          const double d2 = dx * dx + dy * dy;
          if (d2 < bd[KNN_K - 1] || (d2 == bd[KNN_K - 1] && j < bi[KNN_K - 1])) {
            // insertion into the sorted list (ties by index: deterministic)
            int pos = KNN_K - 1;
#pragma unroll
            for (int k = KNN_K - 1; k > 0; --k) {
              const bool up = d2 < bd[k - 1] || (d2 == bd[k - 1] && j < bi[k - 1]);
              if (up) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; pos = k - 1; }
            }
            bd[pos] = d2;
            bi[pos] = j;
          }
        }
      }
    }
    // everything outside the visited (2 ring + 1)^2 square is farther than `reach`
    const double lx = qx - ((double)gr.x0 + (double)(cx - ring) * cell), hx = ((double)gr.x0 + (double)(cx + ring + 1) * cell) - qx;
    const double ly = qy - ((double)gr.y0 + (double)(cy - ring) * cell), hy = ((double)gr.y0 + (double)(cy + ring + 1) * cell) - qy;
    const double reach = fmax(fmin(fmin(lx, hx), fmin(ly, hy)), 0.0);
    if (bi[kk - 1] >= 0 && bd[kk - 1] <= reach * reach) break;
  }
  double wsum = 0.0, acc[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < kk; ++k) {
    if (bi[k] < 0) continue;
    const double w = 1.0 / (sqrt(bd[k]) + 1e-6);
    wsum += w;
    for (int ch = 0; ch < 3; ++ch) acc[ch] += w * (double)known_rgb[bi[k] * 3 + ch];
  }
  for (int ch = 0; ch < 3; ++ch) out_rgb[i * 3 + ch] = wsum > 0.0 ? (float)(acc[ch] / wsum) : 0.0f;
}

// point bins for the k-NN search: MODE as above
template <int MODE>
__global__ __launch_bounds__(256) void point_bin_kernel(const float* __restrict__ xy, int64_t n, ZGrid gr,
                                                        int32_t* __restrict__ counts,
                                                        const int32_t* __restrict__ offsets,
                                                        int32_t* __restrict__ items) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cell_of(xy[i * 2 + 1], gr.y0, gr.inv_cell, gr.g) * gr.g + cell_of(xy[i * 2], gr.x0, gr.inv_cell, gr.g);
  const int k = atomicAdd(&counts[c], 1);
  if (MODE == 1) items[offsets[c] + k] = (int32_t)i;
}

bool grid_ok(float cell, int32_t g) { return cell > 0.0f && g >= 1 && g <= 4096; }

}  // namespace

extern "C" {

int dsu_zgrid_count(const float* tris, int64_t n_faces, float x0, float y0, float cell, int32_t g,
                    int32_t* counts, void* stream) {
  if (n_faces < 0 || !grid_ok(cell, g) || !counts || (n_faces && !tris)) return DSU_EINVAL;
  if (n_faces == 0) return DSU_OK;
  const ZGrid gr{x0, y0, 1.0f / cell, g};
  zgrid_bin_kernel<0><<<dsu_blocks_for(n_faces, 256), 256, 0, (hipStream_t)stream>>>(
      tris, n_faces, gr, counts, nullptr, nullptr);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_zgrid_fill(const float* tris, int64_t n_faces, float x0, float y0, float cell, int32_t g,
                   const int32_t* offsets, int32_t* cursor, int32_t* items, void* stream) {
  if (n_faces < 0 || !grid_ok(cell, g) || !offsets || !cursor || (n_faces && (!tris || !items)))
    return DSU_EINVAL;
  if (n_faces == 0) return DSU_OK;
  const ZGrid gr{x0, y0, 1.0f / cell, g};
  zgrid_bin_kernel<1><<<dsu_blocks_for(n_faces, 256), 256, 0, (hipStream_t)stream>>>(
      tris, n_faces, gr, cursor, offsets, items);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_zray_cast(const float* tris, const int32_t* faces, int64_t n_faces, float x0, float y0,
                  float cell, int32_t g, const int32_t* offsets, const int32_t* items,
                  const float* origins, int64_t n_rays, int32_t sign, const int32_t* self_vertex,
                  int32_t* hit_count, float* t_near, int32_t* face_near, float* t_far,
                  int32_t* face_far, void* stream) {
  if (n_rays < 0 || n_faces < 0 || !grid_ok(cell, g) || (sign != 1 && sign != -1) || !offsets ||
      (n_faces && (!tris || !items)) || (self_vertex && !faces) ||
      (n_rays && (!origins || !hit_count || !t_near || !face_near || !t_far || !face_far)))
    return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  const ZGrid gr{x0, y0, 1.0f / cell, g};
  zray_cast_kernel<<<dsu_blocks_for(n_rays, 256), 256, 0, (hipStream_t)stream>>>(
      tris, faces, gr, offsets, items, origins, n_rays, (float)sign, self_vertex, hit_count, t_near,
      face_near, t_far, face_far);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_raster_mask(const float* tris, int64_t n_faces, float scale, int32_t res, uint8_t* mask,
                    void* stream) {
  if (n_faces < 0 || res < 1 || res > 16384 || !mask || (n_faces && !tris)) return DSU_EINVAL;
  if (n_faces == 0) return DSU_OK;
  raster_mask_kernel<<<dsu_blocks_for(n_faces, 256), 256, 0, (hipStream_t)stream>>>(tris, n_faces,
                                                                                     scale, res, mask);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_erode_ellipse_u8(const uint8_t* src, int32_t H, int32_t W, int32_t ksize, uint8_t* dst,
                         void* stream) {
  if (H < 1 || W < 1 || ksize < 1 || ksize > 255 || !(ksize & 1) || !src || !dst) return DSU_EINVAL;
  erode_ellipse_kernel<<<dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream>>>(src, H, W,
                                                                                        ksize, dst);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_point_bin_count(const float* xy, int64_t n, float x0, float y0, float cell, int32_t g,
                        int32_t* counts, void* stream) {
  if (n < 0 || !grid_ok(cell, g) || !counts || (n && !xy)) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  const ZGrid gr{x0, y0, 1.0f / cell, g};
  point_bin_kernel<0><<<dsu_blocks_for(n, 256), 256, 0, (hipStream_t)stream>>>(xy, n, gr, counts,
                                                                               nullptr, nullptr);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_point_bin_fill(const float* xy, int64_t n, float x0, float y0, float cell, int32_t g,
                       const int32_t* offsets, int32_t* cursor, int32_t* items, void* stream) {
  if (n < 0 || !grid_ok(cell, g) || !offsets || !cursor || (n && (!xy || !items))) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  const ZGrid gr{x0, y0, 1.0f / cell, g};
  point_bin_kernel<1><<<dsu_blocks_for(n, 256), 256, 0, (hipStream_t)stream>>>(xy, n, gr, cursor,
                                                                               offsets, items);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_knn8_blend(const double* query_xy, int64_t n_query, const double* known_xy,
                   const float* known_rgb, int64_t n_known, float x0, float y0, float cell,
                   int32_t g, const int32_t* offsets, const int32_t* items, float* out_rgb,
                   void* stream) {
  if (n_query < 0 || n_known < 1 || !grid_ok(cell, g) || !known_xy || !known_rgb || !offsets ||
      !items || (n_query && (!query_xy || !out_rgb)))
    return DSU_EINVAL;
  if (n_query == 0) return DSU_OK;
  const ZGrid gr{x0, y0, 1.0f / cell, g};
  knn_blend_kernel<<<dsu_blocks_for(n_query, 128), 128, 0, (hipStream_t)stream>>>(
      query_xy, n_query, known_xy, known_rgb, n_known, gr, offsets, items, out_rgb);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
