// Device helpers shared by the hash-grid kernels (hashgrid.hip, hashgrid_mfma.hip).
// See hashgrid.hip for the arithmetic contract.
#pragma once
#include "common.h"
#include <math.h>

namespace dsu_hg {

struct GridMeta {
  uint32_t off[DSU_MAX_LEVELS + 1];
  uint32_t res[DSU_MAX_LEVELS];
  float scale[DSU_MAX_LEVELS];
  uint32_t hashed[DSU_MAX_LEVELS];
};

constexpr int HID = 64;   // n_neurons (neuralangelo-ortho-wmask.yaml:66)
constexpr int NOUT = 13;  // feature_dim (yaml:39)

__device__ __forceinline__ uint32_t grid_index(uint32_t hashed, uint32_t hsize, uint32_t res,
                                               uint32_t x, uint32_t y, uint32_t z) {
  uint32_t idx;
  if (hashed) {
    idx = x ^ (y * 2654435761u) ^ (z * 805459861u);
    return idx & (hsize - 1);  // hashed levels always have a power-of-two size
  }
  idx = x + y * res + z * res * res;
  if (idx >= hsize) idx %= hsize;
  return idx;
}

struct CellPos {
  uint32_t c[3];
  float f[3];
};

__device__ __forceinline__ CellPos cell_of(float scale, float x, float y, float z) {
  CellPos p;
  float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
  float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  p.c[0] = (uint32_t)(int)fx;
  p.c[1] = (uint32_t)(int)fy;
  p.c[2] = (uint32_t)(int)fz;
  p.f[0] = px - fx;
  p.f[1] = py - fy;
  p.f[2] = pz - fz;
  return p;
}

__device__ __forceinline__ float corner_weight(const CellPos& p, int c) {
  float w = 1.0f;
  w *= (c & 1) ? p.f[0] : 1.0f - p.f[0];
  w *= (c & 2) ? p.f[1] : 1.0f - p.f[1];
  w *= (c & 4) ? p.f[2] : 1.0f - p.f[2];
  return w;
}

// One level's trilinear lookup with tcnn's half-precision FMA chain.
__device__ __forceinline__ __half2 lookup_level(const __half2* __restrict__ table,
                                                const GridMeta& m, int l, float x, float y,
                                                float z) {
  const uint32_t hsize = m.off[l + 1] - m.off[l];
  const __half2* lvl = table + m.off[l];
  CellPos p = cell_of(m.scale[l], x, y, z);
  uint32_t idx[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
    idx[c] = grid_index(m.hashed[l], hsize, m.res[l], p.c[0] + (c & 1), p.c[1] + ((c >> 1) & 1),
                        p.c[2] + ((c >> 2) & 1));
  __half2 v[8];
#pragma unroll
#if defined(DSU_FWD_ABLATE) && (DSU_FWD_ABLATE & 2)      // probe builds: no table traffic
  for (int c = 0; c < 8; ++c) { uint32_t b = (idx[c] & 0x03FF03FFu) | 0x20002000u; v[c] = *reinterpret_cast<__half2*>(&b); }
#else
  for (int c = 0; c < 8; ++c) v[c] = lvl[idx[c]];  // 8 independent 4-byte gathers in flight
#endif
  __half2 acc = __float2half2_rn(0.0f);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float wf = corner_weight(p, c);
    // keep the f32 product rounded to f32 BEFORE the f16 conversion (tcnn: (T)weight); without
    // this the compiler folds mul+cvt into one v_fma_mixlo_f16 with a single rounding.
    asm volatile("" : "+v"(wf));
    __half w = __float2half_rn(wf);
    acc = __hfma2(__half2(w, w), v[c], acc);
  }
  return acc;
}

__device__ __forceinline__ float softplus100(float x) {
  // nn.Softplus(beta=100, threshold=20)  (network_utils.py:134-136)
  // evaluated as max(x,0) + log(1 + exp(-|100x|))/100 with the raw hardware exp2/log2
  // (v_exp_f32 / v_log_f32: the arguments stay in (-inf, 0] and [1, 2], so the denormal and range
  // fix-ups of the library forms are not needed): |error| <= ~1e-9 absolute, versus ~100 VALU
  // instructions for libm log1pf(expf()).  BRANCH-FREE: beyond the threshold (100x > 20) exp() is
  // below 2.1e-9, 1 + t rounds to 1 and the expression returns x exactly, as the reference's
  // threshold branch does; the explicit `bx > 20 ? x : ...` form compiled to one exec-mask branch
  // region per hidden unit (32 per lane per evaluation: 3x the instruction stream of the layer).
  const float t = __builtin_amdgcn_exp2f(fabsf(x) * -144.26950408889634f);   // exp(-|100 x|)
  return dsu_relu(x) + __builtin_amdgcn_logf(1.0f + t) * 0.0069314718055994531f;  // ln2 / 100
}
// Two values per lane in packed-f32 instructions (v_pk_mul_f32 / v_pk_add_f32; exp2 / log2 / max
// have no packed form): per value the same operations in the same order as softplus100 — the
// multiplication by 144.27 commutes with |.| and the sign bit for bit — hence the same result.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 softplus100_pair(f32x2 x) {
  const f32x2 y = x * splat2(144.26950408889634f);
  const f32x2 t = {__builtin_amdgcn_exp2f(-fabsf(y.x)), __builtin_amdgcn_exp2f(-fabsf(y.y))};
  const f32x2 u = splat2(1.0f) + t;
  const f32x2 lg = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
  const f32x2 mx = {dsu_relu(x.x), dsu_relu(x.y)};
  return mx + lg * splat2(0.0069314718055994531f);
}
__device__ __forceinline__ float softplus100_grad(float x) {
  // sigmoid(100 x) = 1 / (1 + exp(-100 x)); exp2(+large) = inf -> rcp(inf) = 0
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -144.26950408889634f));
}

__device__ __forceinline__ float contract(float p, float radius) {
  // scale_anything(x, (-r, r), (0, 1))  (instant_nsr/models/utils.py:101-106)
  float d = (p - (-radius)) / (radius - (-radius));
  return d * (1.0f - 0.0f) + 0.0f;
}

constexpr int GC_LOG2 = 12;
constexpr int GC_SLOTS = 1 << GC_LOG2;
constexpr uint32_t GC_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ void grad_cache_add(uint32_t* keys, float* vals,
                                               float* __restrict__ gtable, uint32_t entry,
                                               float v0, float v1) {
  uint32_t slot = (entry * 2654435761u) >> (32 - GC_LOG2);
#pragma unroll
  for (int probe = 0; probe < 3; ++probe) {
    const uint32_t old = atomicCAS(&keys[slot], GC_EMPTY, entry);
    if (old == GC_EMPTY || old == entry) {
      atomicAdd(&vals[2 * slot], v0);       // ds_add_f32
      atomicAdd(&vals[2 * slot + 1], v1);
      return;
    }
    slot = (slot + 1) & (GC_SLOTS - 1);
  }
  unsafeAtomicAdd(gtable + (size_t)entry * 2, v0);
  unsafeAtomicAdd(gtable + (size_t)entry * 2 + 1, v1);
}

__device__ __forceinline__ uint32_t grad_cache_slot(uint32_t entry) {
  return (entry * 2654435761u) >> (32 - GC_LOG2);
}

int make_meta(const dsu_hashgrid_cfg* cfg, GridMeta* m);

}  // namespace dsu_hg
