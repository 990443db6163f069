// cv2.inpaint(img, mask, radius, cv2.INPAINT_TELEA) for 8-bit 3-channel images — the tail of the
// contour remover (1_lama_contour_remover/predict.py:61-64: the predicted contour pixels and the
// background are filled from the surrounding character pixels).
//
// HOST code, as in the reference (OpenCV runs this on the CPU): Telea's fast-marching method is a
// strictly ordered front propagation (a priority queue pop decides every next pixel), there is no
// data-parallel form that reproduces its results, and at 512 x 512 it costs ~0.1 s of one core.
// It lives in libdsu_hip.so so that the stage has ONE native implementation behind the C ABI and
// fails loudly with the rest of the library.
//
// Restated from OpenCV's published algorithm (modules/photo/src/inpaint.cpp, 4.x: cvInpaint ->
// icvCalcFMM on the outside band -> icvTeleaInpaintFMM) — OpenCV is not installed in this image:
// PARITY UNPINNED.  Conventions kept: 1-pixel frame around the image (erows = rows + 2), flags
// KNOWN 0 / BAND 1 / INSIDE 2 / CHANGE 3, T = 1e6 where unset, the FIFO-stable ordered queue
// (equal T: first pushed, first popped; the initial band in row-major order), neighbours visited
// in the order up, left, down, right, the eikonal update FastMarching_solve in double, the
// weighting function of the paper with OpenCV's constants (|r|^-3 distance term, level term
// 1 / (1 + |dT|), direction term with the 0.01 -> 1e-6 floor), image gradients doubled for central
// differences, `Ia / s + (Jx + Jy) / (|J| + 1e-20) + 0.5` rounded and saturated, and the first
// image row / column never being filled (frame test `i <= 1`).
#include "common.h"

#include <math.h>
#include <queue>
#include <vector>

namespace {

enum : uint8_t { KNOWN = 0, BAND = 1, INSIDE = 2, CHANGE = 3 };

struct HeapElem {
  float T;
  uint64_t order;      // push counter: stable among equal T
  int i, j;
};
struct HeapCmp {
  bool operator()(const HeapElem& a, const HeapElem& b) const {
    return a.T > b.T || (a.T == b.T && a.order > b.order);
  }
};
struct Queue {
  std::priority_queue<HeapElem, std::vector<HeapElem>, HeapCmp> q;
  uint64_t n = 0;
  void push(int i, int j, float T) { q.push(HeapElem{T, n++, i, j}); }
  bool pop(int& i, int& j) {
    if (q.empty()) return false;
    i = q.top().i; j = q.top().j;
    q.pop();
    return true;
  }
};

float fmm_solve(int i1, int j1, int i2, int j2, const std::vector<uint8_t>& f,
                const std::vector<float>& t, int cols) {
  const double a11 = t[(size_t)i1 * cols + j1], a22 = t[(size_t)i2 * cols + j2];
  const double m12 = a11 < a22 ? a11 : a22;
  double sol;
  if (f[(size_t)i1 * cols + j1] != INSIDE) {
    if (f[(size_t)i2 * cols + j2] != INSIDE) {
      if (fabs(a11 - a22) >= 1.0) sol = 1 + m12;
      else sol = (a11 + a22 + sqrt((double)(2 - (a11 - a22) * (a11 - a22)))) * 0.5;
    } else {
      sol = 1 + a11;
    }
  } else if (f[(size_t)i2 * cols + j2] != INSIDE) {
    sol = 1 + a22;
  } else {
    sol = 1 + m12;
  }
  return (float)sol;
}

inline float min4(float a, float b, float c, float d) {
  a = a < b ? a : b;
  c = c < d ? c : d;
  return a < c ? a : c;
}

float fmm_dist(int i, int j, const std::vector<uint8_t>& f, const std::vector<float>& t, int cols) {
  return min4(fmm_solve(i - 1, j, i, j - 1, f, t, cols), fmm_solve(i + 1, j, i, j - 1, f, t, cols),
              fmm_solve(i - 1, j, i, j + 1, f, t, cols), fmm_solve(i + 1, j, i, j + 1, f, t, cols));
}

const int DI[4] = {-1, 0, 1, 0}, DJ[4] = {0, -1, 0, 1};

// distances of the KNOWN pixels around the hole (negated), marching outwards over `f == INSIDE`;
// er x ec = framed sizes (row stride ec)
void calc_fmm(std::vector<uint8_t>& f, std::vector<float>& t, int er, int ec, Queue& heap) {
  int ii, jj;
  while (heap.pop(ii, jj)) {
    f[(size_t)ii * ec + jj] = CHANGE;
    for (int q = 0; q < 4; ++q) {
      const int i = ii + DI[q], j = jj + DJ[q];
      if (i <= 0 || j <= 0 || i >= er - 1 || j >= ec - 1) continue;   // the frame is KNOWN
      if (f[(size_t)i * ec + j] == INSIDE) {
        const float dist = fmm_dist(i, j, f, t, ec);
        t[(size_t)i * ec + j] = dist;
        f[(size_t)i * ec + j] = BAND;
        heap.push(i, j, dist);
      }
    }
  }
  for (size_t k = 0; k < f.size(); ++k)
    if (f[k] == CHANGE) {
      f[k] = KNOWN;
      t[k] = -t[k];
    }
}

inline uint8_t sat_u8(float v) {
  const float r = nearbyintf(v);                 // cvRound: to nearest, ties to even
  return (uint8_t)(r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r));
}

}  // namespace

extern "C" {

int dsu_inpaint_telea_u8c3(const uint8_t* img, const uint8_t* mask, int32_t rows, int32_t cols,
                           int32_t radius, uint8_t* out) {
  if (!img || !mask || !out || rows < 1 || cols < 1) return DSU_EINVAL;
  if (rows < 3 || cols < 3) return DSU_EUNSUP;      // the border handling reads rows/columns 1 and n-2
  int range = radius;
  if (range < 1) range = 1;
  if (range > 100) range = 100;
  const int er = rows + 2, ec = cols + 2;
  const size_t ne = (size_t)er * ec;
  for (size_t k = 0; k < (size_t)rows * cols * 3; ++k) out[k] = img[k];

  std::vector<uint8_t> m(ne, KNOWN), band(ne, 0);
  std::vector<float> t(ne, 1.0e6f);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j)
      if (mask[(size_t)i * cols + j]) m[(size_t)(i + 1) * ec + j + 1] = INSIDE;
  // band = dilate(mask, 3x3 cross) - mask, frame excluded
  auto M = [&](int i, int j) -> uint8_t { return m[(size_t)i * ec + j]; };
  Queue heap;
  for (int i = 1; i < er - 1; ++i)
    for (int j = 1; j < ec - 1; ++j) {
      if (M(i, j) != KNOWN) continue;
      if (M(i - 1, j) || M(i + 1, j) || M(i, j - 1) || M(i, j + 1)) {
        band[(size_t)i * ec + j] = 1;
        t[(size_t)i * ec + j] = 0.0f;
        heap.push(i, j, 0.0f);
      }
    }
  for (size_t k = 0; k < ne; ++k)
    if (m[k] == INSIDE) t[k] = 1.0e6f;

  // ---- outside distances: region within `range` (square dilation) of the hole, minus hole and band
  {
    std::vector<uint8_t> o(ne, KNOWN);
    // separable square dilation of the hole by `range`
    std::vector<uint8_t> rowd(ne, 0);
    for (int i = 0; i < er; ++i) {
      int last = -1000000;
      for (int j = 0; j < ec; ++j) {                       // nearest hole pixel to the left
        if (m[(size_t)i * ec + j] == INSIDE) last = j;
        if (j - last <= range) rowd[(size_t)i * ec + j] = 1;
      }
      last = 1000000;
      for (int j = ec - 1; j >= 0; --j) {
        if (m[(size_t)i * ec + j] == INSIDE) last = j;
        if (last - j <= range) rowd[(size_t)i * ec + j] = 1;
      }
    }
    for (int j = 0; j < ec; ++j) {
      int last = -1000000;
      std::vector<uint8_t> col(er, 0);
      for (int i = 0; i < er; ++i) {
        if (rowd[(size_t)i * ec + j]) last = i;
        if (i - last <= range) col[i] = 1;
      }
      last = 1000000;
      for (int i = er - 1; i >= 0; --i) {
        if (rowd[(size_t)i * ec + j]) last = i;
        if (last - i <= range) col[i] = 1;
      }
      for (int i = 0; i < er; ++i)
        if (col[i] && m[(size_t)i * ec + j] != INSIDE && !band[(size_t)i * ec + j])
          o[(size_t)i * ec + j] = INSIDE;
    }
    for (int i = 0; i < er; ++i) o[(size_t)i * ec] = o[(size_t)i * ec + ec - 1] = KNOWN;
    for (int j = 0; j < ec; ++j) o[j] = o[(size_t)(er - 1) * ec + j] = KNOWN;
    Queue outq;
    for (int i = 1; i < er - 1; ++i)
      for (int j = 1; j < ec - 1; ++j)
        if (band[(size_t)i * ec + j]) outq.push(i, j, 0.0f);
    calc_fmm(o, t, er, ec, outq);
  }

  // distance term 1 / |r|^3 per window offset (same float/double operations as the in-loop form)
  std::vector<float> dst_tab((size_t)(2 * range + 1) * (2 * range + 1), 0.0f);
  for (int dk = -range; dk <= range; ++dk)
    for (int dl = -range; dl <= range; ++dl) {
      const float ry = (float)(-dk), rx = (float)(-dl);
      const float len2 = rx * rx + ry * ry;
      dst_tab[(size_t)(dk + range) * (2 * range + 1) + (dl + range)] =
          (float)(1. / (len2 * sqrt((double)len2)));
    }

  // ---- Telea: march into the hole (flags = the hole mask; band pixels are KNOWN there)
  std::vector<uint8_t>& f = m;
  auto OUT = [&](int i, int j, int c) -> int { return out[((size_t)i * cols + j) * 3 + c]; };
  int ii, jj;
  while (heap.pop(ii, jj)) {
    f[(size_t)ii * ec + jj] = KNOWN;
    for (int q = 0; q < 4; ++q) {
      const int i = ii + DI[q], j = jj + DJ[q];
      if (i <= 1 || j <= 1 || i > er - 1 || j > ec - 1) continue;
      if (i >= er - 1 || j >= ec - 1) continue;                 // the frame is never a hole
      if (f[(size_t)i * ec + j] != INSIDE) continue;
      const float dist = fmm_dist(i, j, f, t, ec);
      t[(size_t)i * ec + j] = dist;
      auto Fl = [&](int a, int b) -> uint8_t { return f[(size_t)a * ec + b]; };
      auto Tt = [&](int a, int b) -> float { return t[(size_t)a * ec + b]; };
      float gtx, gty;
      if (Fl(i, j + 1) != INSIDE) {
        if (Fl(i, j - 1) != INSIDE) gtx = (float)(Tt(i, j + 1) - Tt(i, j - 1)) * 0.5f;
        else gtx = (float)(Tt(i, j + 1) - Tt(i, j));
      } else {
        if (Fl(i, j - 1) != INSIDE) gtx = (float)(Tt(i, j) - Tt(i, j - 1));
        else gtx = 0;
      }
      if (Fl(i + 1, j) != INSIDE) {
        if (Fl(i - 1, j) != INSIDE) gty = (float)(Tt(i + 1, j) - Tt(i - 1, j)) * 0.5f;
        else gty = (float)(Tt(i + 1, j) - Tt(i, j));
      } else {
        if (Fl(i - 1, j) != INSIDE) gty = (float)(Tt(i, j) - Tt(i - 1, j));
        else gty = 0;
      }
      // the three colours share the weights (OpenCV recomputes them per colour: same values)
      float Ia[3] = {0, 0, 0}, Jx[3] = {0, 0, 0}, Jy[3] = {0, 0, 0}, s = 1.0e-20f;
      for (int k = i - range; k <= i + range; ++k) {
        const int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2);
        if (!(k > 0 && k < er - 1)) continue;
        for (int l = j - range; l <= j + range; ++l) {
          const int lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
          if (!(l > 0 && l < ec - 1)) continue;
          if (Fl(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
          const float ry = (float)(i - k), rx = (float)(j - l);
          const float dst = dst_tab[(size_t)(k - i + range) * (2 * range + 1) + (l - j + range)];
          const float lev = (float)(1. / (1. + (double)fabsf(Tt(k, l) - Tt(i, j))));
          float dir = rx * gtx + ry * gty;
          if ((double)fabsf(dir) <= 0.01) dir = 0.000001f;
          const float w = fabsf(dst * lev * dir);
          const bool xr = Fl(k, l + 1) != INSIDE, xl = Fl(k, l - 1) != INSIDE;
          const bool yd = Fl(k + 1, l) != INSIDE, yu = Fl(k - 1, l) != INSIDE;
          for (int color = 0; color <= 2; ++color) {
            float gix, giy;
            if (xr) {
              if (xl) gix = (float)(OUT(km, lp + 1, color) - OUT(km, lm - 1, color)) * 2.0f;
              else gix = (float)(OUT(km, lp + 1, color) - OUT(km, lm, color));
            } else {
              if (xl) gix = (float)(OUT(km, lp, color) - OUT(km, lm - 1, color));
              else gix = 0;
            }
            if (yd) {
              if (yu) giy = (float)(OUT(kp + 1, lm, color) - OUT(km - 1, lm, color)) * 2.0f;
              else giy = (float)(OUT(kp + 1, lm, color) - OUT(km, lm, color));
            } else {
              if (yu) giy = (float)(OUT(kp, lm, color) - OUT(km - 1, lm, color));
              else giy = 0;
            }
            Ia[color] += (float)w * (float)(OUT(km, lm, color));
            Jx[color] -= (float)w * (float)(gix * rx);
            Jy[color] -= (float)w * (float)(giy * ry);
          }
          s += w;
        }
      }
      for (int color = 0; color <= 2; ++color) {
        const float sat = Ia[color] / s + (Jx[color] + Jy[color]) /
                              (sqrtf(Jx[color] * Jx[color] + Jy[color] * Jy[color]) + 1.0e-20f) + 0.5f;
        out[((size_t)(i - 1) * cols + (j - 1)) * 3 + color] = sat_u8(sat);
      }
      f[(size_t)i * ec + j] = BAND;
      heap.push(i, j, dist);
    }
  }
  return DSU_OK;
}

}  // extern "C"
