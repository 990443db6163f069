// Image-side host steps of the export's thinning (2_charactor_reconstructor/instant_nsr/utils/
// thinning_utils.py:205-239): cv2.distanceTransform(mask, cv2.DIST_L2, 5) and
// skimage.morphology.skeletonize(mask, method='lee') on the character mask.
//
// HOST code, as in the reference (OpenCV / scikit-image run these on the CPU): both are raster-order
// recurrences (the chamfer passes carry the running minimum along the row; the thinning re-checks
// its candidates one after the other), on a single 1024 x 1024 mask per drawing.
//
// Neither OpenCV nor scikit-image is installed in this image: PARITY UNPINNED.  Restated from the
// published algorithms:
//   * distance transform: Borgefors' two-pass 5x5 chamfer with OpenCV's documented DIST_L2
//     weights a = 1, b = 1.4, c = 2.1969 (horizontal/vertical, diagonal, knight move), in 16.16
//     fixed point, a two-pixel frame of "infinity" around the image, result scaled back to float;
//   * skeleton: Lee, Kashyap, Chu, "Building skeleton models via 3-D medial surface/axis thinning
//     algorithms" (CVGIP 1994) as scikit-image applies it to a 2-D image (a one-slice volume):
//     repeat over the six border directions — collect the border points of that direction that
//     are not end points, whose removal keeps the Euler characteristic and the number of
//     26-connected object components in their neighbourhood; then re-check the candidates one at
//     a time in raster order (components only) and delete.  In a one-slice volume the slices
//     above and below are background and connect every background neighbour, so the two
//     topological tests reduce to: ONE 8-connected object component among the 8 neighbours, and
//     at least one background pixel among the 4 edge neighbours (an interior pixel would open a
//     tunnel through the slice).  Direction order 4, 3, 2, 1, 5, 6 (scikit-image's), where
//     1..4 = the neighbour at column-1, column+1, row+1, row-1 is background and 5, 6 = above /
//     below (always background here).
#include "common.h"

#include <stdint.h>
#include <string.h>

#include <vector>

namespace {

// number of 8-connected object components among the 8 neighbours (ring order N, NE, E, SE, S, SW,
// W, NW; 1 = object), from a 256-entry table built by flood fill over the ring's own adjacency
// (two ring pixels touch when their row and column offsets differ by at most one)
struct RingTable {
  uint8_t comps[256];
  RingTable() {
    const int dr[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, dc[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    for (int m = 0; m < 256; ++m) {
      int label[8] = {0}, n = 0;
      for (int s0 = 0; s0 < 8; ++s0) {
        if (!((m >> s0) & 1) || label[s0]) continue;
        ++n;
        int stack[8], top = 0;
        stack[top++] = s0;
        label[s0] = n;
        while (top) {
          const int u = stack[--top];
          for (int w = 0; w < 8; ++w) {
            if (!((m >> w) & 1) || label[w]) continue;
            const int a = dr[u] - dr[w], b = dc[u] - dc[w];
            if (a >= -1 && a <= 1 && b >= -1 && b <= 1) { label[w] = n; stack[top++] = w; }
          }
        }
      }
      comps[m] = (uint8_t)n;
    }
  }
};
inline int ring_components(const uint8_t nb[8]) {
  static const RingTable table;
  int m = 0;
  for (int i = 0; i < 8; ++i) m |= (nb[i] ? 1 : 0) << i;
  return table.comps[m];
}

}  // namespace

extern "C" {

int dsu_distance_transform_l2_5x5(const uint8_t* mask, int32_t H, int32_t W, float* out) {
  if (H < 0 || W < 0 || ((int64_t)H * W > 0 && (!mask || !out))) return DSU_EINVAL;
  if ((int64_t)H * W == 0) return DSU_OK;
  const int B = 2;
  const int64_t step = W + 2 * B;
  const int32_t HV = 65536, DIAG = 91750, LONG = 143976;   // round(x * 2^16) of 1, 1.4, 2.1969
  const int32_t INF = INT32_MAX >> 2;
  std::vector<int32_t> t((size_t)(H + 2 * B) * step, INF);
  auto at = [&](int64_t i, int64_t j) -> int32_t& { return t[(size_t)(i + B) * step + (j + B)]; };
  // forward pass
  for (int32_t i = 0; i < H; ++i)
    for (int32_t j = 0; j < W; ++j) {
      if (!mask[(size_t)i * W + j]) { at(i, j) = 0; continue; }
      int32_t m = at(i - 2, j - 1) + LONG;
      int32_t v;
      v = at(i - 2, j + 1) + LONG; if (v < m) m = v;
      v = at(i - 1, j - 2) + LONG; if (v < m) m = v;
      v = at(i - 1, j - 1) + DIAG; if (v < m) m = v;
      v = at(i - 1, j) + HV;       if (v < m) m = v;
      v = at(i - 1, j + 1) + DIAG; if (v < m) m = v;
      v = at(i - 1, j + 2) + LONG; if (v < m) m = v;
      v = at(i, j - 1) + HV;       if (v < m) m = v;
      at(i, j) = m;
    }
  // backward pass
  for (int32_t i = H - 1; i >= 0; --i)
    for (int32_t j = W - 1; j >= 0; --j) {
      int32_t m = at(i, j);
      if (m > HV) {
        int32_t v;
        v = at(i + 2, j + 1) + LONG; if (v < m) m = v;
        v = at(i + 2, j - 1) + LONG; if (v < m) m = v;
        v = at(i + 1, j + 2) + LONG; if (v < m) m = v;
        v = at(i + 1, j + 1) + DIAG; if (v < m) m = v;
        v = at(i + 1, j) + HV;       if (v < m) m = v;
        v = at(i + 1, j - 1) + DIAG; if (v < m) m = v;
        v = at(i + 1, j - 2) + LONG; if (v < m) m = v;
        v = at(i, j + 1) + HV;       if (v < m) m = v;
        at(i, j) = m;
      }
      out[(size_t)i * W + j] = (float)m * (1.0f / 65536.0f);
    }
  return DSU_OK;
}

int dsu_skeletonize_lee_2d(const uint8_t* img, int32_t H, int32_t W, uint8_t* out) {
  if (H < 0 || W < 0 || ((int64_t)H * W > 0 && (!img || !out))) return DSU_EINVAL;
  if ((int64_t)H * W == 0) return DSU_OK;
  const int64_t step = W + 2;
  std::vector<uint8_t> a((size_t)(H + 2) * step, 0);
  for (int32_t i = 0; i < H; ++i)
    for (int32_t j = 0; j < W; ++j) a[(size_t)(i + 1) * step + j + 1] = img[(size_t)i * W + j] ? 1 : 0;
  // ring offsets N, NE, E, SE, S, SW, W, NW
  const int64_t ring[8] = {-step, -step + 1, 1, step + 1, step, step - 1, -1, -step - 1};
  // border direction d = 1..4 -> the neighbour that must be background
  const int64_t dir_off[5] = {0, -1, 1, step, -step};
  const int order[6] = {4, 3, 2, 1, 5, 6};
  std::vector<int64_t> cand;
  int unchanged = 0;
  while (unchanged < 6) {
    unchanged = 0;
    for (int oi = 0; oi < 6; ++oi) {
      const int d = order[oi];
      cand.clear();
      for (int32_t i = 1; i <= H; ++i)
        for (int32_t j = 1; j <= W; ++j) {
          const int64_t p = (int64_t)i * step + j;
          if (!a[p]) continue;
          if (d <= 4 && a[p + dir_off[d]]) continue;       // not a border point of this direction
          uint8_t nb[8];
          int n = 0;
          for (int k = 0; k < 8; ++k) { nb[k] = a[p + ring[k]]; n += nb[k]; }
          if (n == 1) continue;                             // end point
          // Euler characteristic kept: some edge neighbour is background (and the point is not
          // isolated: removing a lone pixel removes a component)
          if (n == 0 || (nb[0] && nb[2] && nb[4] && nb[6])) continue;
          if (ring_components(nb) != 1) continue;           // simple point
          cand.push_back(p);
        }
      bool changed = false;
      for (int64_t p : cand) {
        uint8_t nb[8];
        for (int k = 0; k < 8; ++k) nb[k] = a[p + ring[k]];
        if (ring_components(nb) <= 1) {
          a[p] = 0;
          changed = true;
        }
      }
      if (!changed) ++unchanged;
    }
  }
  for (int32_t i = 0; i < H; ++i)
    for (int32_t j = 0; j < W; ++j) out[(size_t)i * W + j] = a[(size_t)(i + 1) * step + j + 1] ? 255 : 0;
  return DSU_OK;
}

}  // extern "C"
