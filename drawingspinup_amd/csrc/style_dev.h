// Device helpers shared by the style-translator kernels (style_conv.hip, style_train.hip).
#pragma once
#include "common.h"

namespace dsu_style {

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case 1: return fmaxf(v, 0.0f);
    case 2: return v > 0.0f ? v : 0.2f * v;
    case 3: return tanhf(v);
    default: return v;
  }
}

// bilinear sampling state of one (pixel, tap): clamped row/col offsets + 4 weights with the
// out-of-range corners zeroed (torchvision deform_conv2d bilinear_interpolate semantics).
struct Tap {
  int r0, r1, c0, c1;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ Tap make_tap(float h, float w, int H, int W) {
  Tap t;
  const bool inside = (h > -1.0f) && (w > -1.0f) && (h < (float)H) && (w < (float)W);
  const float hl = floorf(h), wl = floorf(w);
  const int h0 = (int)hl, w0 = (int)wl, h1 = h0 + 1, w1 = w0 + 1;
  const float lh = h - hl, lw = w - wl, hh = 1.0f - lh, hw = 1.0f - lw;
  const bool vh0 = inside && h0 >= 0, vh1 = inside && h1 <= H - 1;
  const bool vw0 = w0 >= 0, vw1 = w1 <= W - 1;
  t.w00 = (vh0 && vw0) ? hh * hw : 0.0f;
  t.w01 = (vh0 && vw1) ? hh * lw : 0.0f;
  t.w10 = (vh1 && vw0) ? lh * hw : 0.0f;
  t.w11 = (vh1 && vw1) ? lh * lw : 0.0f;
  t.r0 = min(max(h0, 0), H - 1) * W;
  t.r1 = min(max(h1, 0), H - 1) * W;
  t.c0 = min(max(w0, 0), W - 1);
  t.c1 = min(max(w1, 0), W - 1);
  return t;
}

}  // namespace dsu_style
