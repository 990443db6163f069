// torch.optim.AdamW on the hash-table parameters, shared by the stand-alone launch
// (dsu_table_adamw, nsr_step.hip) and the native step's last launch (nsr_driver.hip), where the
// table update runs in the workgroups beside the single-workgroup small-tensor update.
#pragma once
#include "common.h"
#include <hip/hip_fp16.h>

struct dsu_table_adamw_args {
  float4 *p, *g, *m, *v;
  __half2* img;
  int64_t n4;
  float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
};

// elements first, first + stride, ... of the float4 view: AdamW (betas, eps, decoupled weight
// decay), the f16 image the kernels read rewritten and the gradient zeroed in the same pass
__device__ __forceinline__ void dsu_table_adamw_range(const dsu_table_adamw_args& a, int64_t first,
                                                      int64_t stride) {
  const float step_size = a.lr / a.bc1;
  // 1 - beta from the double values torch uses (1 - 0.99f is 9.5e-7 off 0.01)
  const float omb1 = (float)(1.0 - (double)a.beta1), omb2 = (float)(1.0 - (double)a.beta2);
  for (int64_t i = first; i < a.n4; i += stride) {
    float4 P = a.p[i], G = a.g[i], M = a.m[i], V = a.v[i];
    float* pp = &P.x; float* gg = &G.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x = pp[k];
      x -= a.lr * a.wd * x;
      const float mk = mm[k] + (gg[k] - mm[k]) * omb1;                  // lerp, as torch
      const float vk = a.beta2 * vv[k] + omb2 * gg[k] * gg[k];
      const float denom = sqrtf(vk) / a.bc2_sqrt + a.eps;
      x -= step_size * mk / denom;
      pp[k] = x; mm[k] = mk; vv[k] = vk;
    }
    a.p[i] = P; a.m[i] = M; a.v[i] = V;
    a.g[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    a.img[2 * i] = __floats2half2_rn(P.x, P.y);
    a.img[2 * i + 1] = __floats2half2_rn(P.z, P.w);
  }
}
