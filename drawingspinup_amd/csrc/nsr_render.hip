// Occupancy-grid ray marching and per-ray compositing for gfx950.
//
// Replaces the nerfacc==0.3.3 ops the reference calls at
// 2_charactor_reconstructor/instant_nsr/models/neus.py:53-57 (OccupancyGrid), :84
// (every_n_step), :119-129 (ray_marching), :147-152 (render_weight_from_alpha,
// accumulate_along_rays).  nerfacc is an un-vendored dependency (requirements.txt:14);
// the stepping rule below restates its published ray_marching kernel
// (nerfacc/cuda/csrc/ray_marching.cu @ v0.3.3) and is mirrored by oracle/nerfacc_ref.py.
#include "common.h"
#include <math.h>

namespace {

struct Aabb {
  float mn[3];
  float mx[3];
  float inv_ext[3];  // 1/(mx-mn), only used when pow2 != 0
  int pow2;          // every extent and the grid resolution are powers of two: x/ext == x*(1/ext)
                     // bit for bit, so the serial marching chain can avoid IEEE divisions
};

__device__ __forceinline__ float div_ext(float x, const Aabb& a, int d) {
  return a.pow2 ? x * a.inv_ext[d] : x / (a.mx[d] - a.mn[d]);
}
__device__ __forceinline__ float div_res(float x, const Aabb& a, int res, float inv_res) {
  return a.pow2 ? x * inv_res : x / (float)res;
}

__device__ __forceinline__ float signf1(float x) { return copysignf(1.0f, x); }

__device__ __forceinline__ bool occupied_at(const float p[3], const Aabb& a,
                                            const uint8_t* __restrict__ occ, int res) {
  if (p[0] < a.mn[0] || p[0] > a.mx[0] || p[1] < a.mn[1] || p[1] > a.mx[1] || p[2] < a.mn[2] ||
      p[2] > a.mx[2])
    return false;
  int ix[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float u = div_ext(p[d] - a.mn[d], a, d);
    int i = (int)(u * (float)res);
    ix[d] = min(max(i, 0), res - 1);
  }
  return occ[(ix[0] * res + ix[1]) * res + ix[2]] != 0;
}

__device__ __forceinline__ float dist_to_next_voxel(const float p[3], const float dir[3],
                                                    const float inv_dir[3], const Aabb& a,
                                                    int res) {
  float t = INFINITY;
  const float inv_res = 1.0f / (float)res;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float g = div_ext(p[d] - a.mn[d], a, d) * (float)res;
    float td = div_res((floorf(g + 0.5f + 0.5f * signf1(dir[d])) - g) * inv_dir[d], a, res,
                       inv_res) * (a.mx[d] - a.mn[d]);
    t = fminf(t, td);
  }
  return fmaxf(t, 0.0f);
}

// ray_aabb_intersect (nerfacc intersection.cu) + stratified jitter of t_min
// (nerfacc ray_marching.py: t_min = t_min + rand_like(t_min) * render_step_size).
__global__ void ray_aabb_kernel(const float* __restrict__ ro, const float* __restrict__ rd,
                                int64_t n, Aabb a, const float* __restrict__ jitter, float step,
                                float* __restrict__ tmin_o, float* __restrict__ tmax_o) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float o[3] = {ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2]};
  const float d[3] = {rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2]};
  float tmin = (a.mn[0] - o[0]) / d[0], tmax = (a.mx[0] - o[0]) / d[0];
  if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
  float near = 1e10f, far = 1e10f;
  bool hit = true;
  float tymin = (a.mn[1] - o[1]) / d[1], tymax = (a.mx[1] - o[1]) / d[1];
  if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
  if (tmin > tymax || tymin > tmax) hit = false;
  if (hit) {
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (a.mn[2] - o[2]) / d[2], tzmax = (a.mx[2] - o[2]) / d[2];
    if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) hit = false;
    if (hit) {
      if (tzmin > tmin) tmin = tzmin;
      if (tzmax < tmax) tmax = tzmax;
      near = tmin;
      far = tmax;
    }
  }
  if (jitter != nullptr) near = near + jitter[i] * step;
  tmin_o[i] = near;
  tmax_o[i] = far;
}

template <bool FILL>
__global__ void ray_march_kernel(const float* __restrict__ ro, const float* __restrict__ rd,
                                 const float* __restrict__ tmin, const float* __restrict__ tmax,
                                 int64_t n, Aabb a, const uint8_t* __restrict__ occ, int res,
                                 float step, const int32_t* __restrict__ offsets,
                                 int32_t* __restrict__ num_steps,
                                 int64_t* __restrict__ ray_indices, float* __restrict__ t_starts,
                                 float* __restrict__ t_ends) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float o[3] = {ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2]};
  const float d[3] = {rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2]};
  const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
  const float near = tmin[i], far = tmax[i];
  int64_t base = 0;
  if (FILL) base = offsets[i];
  const float dt = step;  // cone_angle == 0: calc_dt clamps to dt_min
  int j = 0;
  float t0 = near, t1 = t0 + dt, tm = (t0 + t1) * 0.5f;
  while (tm < far) {
    const float p[3] = {o[0] + tm * d[0], o[1] + tm * d[1], o[2] + tm * d[2]};
    if (occ == nullptr || occupied_at(p, a, occ, res)) {
      if (FILL) {
        t_starts[base + j] = t0;
        t_ends[base + j] = t1;
        ray_indices[base + j] = i;
      }
      ++j;
      t0 = t1;
      t1 = t0 + dt;
      tm = (t0 + t1) * 0.5f;
    } else {
      // advance_to_next_voxel: keep the sample lattice, skip to the first lattice point
      // at or beyond the next voxel boundary
      float target = tm + dist_to_next_voxel(p, d, inv, a, res);
      target = fminf(target, far);
      do { tm += dt; } while (tm < target);
      t0 = tm - dt * 0.5f;
      t1 = tm + dt * 0.5f;
    }
  }
  if (!FILL) num_steps[i] = j;
}

__global__ void weights_fwd_kernel(const float* __restrict__ alpha,
                                   const int32_t* __restrict__ off,
                                   const int32_t* __restrict__ cnt, int64_t n_rays,
                                   float* __restrict__ w) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const int64_t b = off[r];
  const int c = cnt[r];
  float T = 1.0f;
  for (int j = 0; j < c; ++j) {
    const float al = alpha[b + j];
    w[b + j] = al * T;
    T *= (1.0f - al);
  }
}

__global__ void weights_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ w,
                                   const float* __restrict__ gw,
                                   const int32_t* __restrict__ off,
                                   const int32_t* __restrict__ cnt, int64_t n_rays,
                                   float* __restrict__ galpha) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const int64_t b = off[r];
  const int c = cnt[r];
  float accum = 0.0f;
  for (int j = 0; j < c; ++j) accum += gw[b + j] * w[b + j];
  float T = 1.0f;
  for (int j = 0; j < c; ++j) {
    const float al = alpha[b + j];
    galpha[b + j] = (gw[b + j] * T - accum) / fmaxf(1.0f - al, 1e-10f);
    accum -= gw[b + j] * w[b + j];
    T *= (1.0f - al);
  }
}

// one thread per (ray, channel)
__global__ void accumulate_fwd_kernel(const float* __restrict__ w, const float* __restrict__ v,
                                      int ch, const int32_t* __restrict__ off,
                                      const int32_t* __restrict__ cnt, int64_t n_rays,
                                      float* __restrict__ out) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n_rays * ch) return;
  const int64_t r = t / ch;
  const int c = (int)(t % ch);
  const int64_t b = off[r];
  const int k = cnt[r];
  float acc = 0.0f;
  if (v == nullptr) {
    for (int j = 0; j < k; ++j) acc += w[b + j];
  } else {
    for (int j = 0; j < k; ++j) acc += w[b + j] * v[(b + j) * ch + c];
  }
  out[t] = acc;
}

__global__ void occ_ema_kernel(float* __restrict__ occs, const int64_t* __restrict__ idx,
                               const float* __restrict__ occ, int64_t n, float decay) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = idx ? idx[i] : i;
    occs[c] = fmaxf(occs[c] * decay, occ[i]);
  }
}

__global__ void occ_bin_kernel(const float* __restrict__ occs, int64_t n, float thre,
                               uint8_t* __restrict__ bin) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    bin[i] = occs[i] > thre ? 1 : 0;
}

bool is_pow2f(float x) {
  int e;
  return x > 0.0f && frexpf(x, &e) == 0.5f;
}

Aabb make_aabb(const float* a6, int res = 1) {
  Aabb a;
  a.pow2 = (res > 0 && (res & (res - 1)) == 0) ? 1 : 0;
  for (int d = 0; d < 3; ++d) {
    a.mn[d] = a6[d];
    a.mx[d] = a6[3 + d];
    const float ext = a.mx[d] - a.mn[d];
    a.inv_ext[d] = 1.0f / ext;
    if (!is_pow2f(ext)) a.pow2 = 0;
  }
  return a;
}

}  // namespace

extern "C" {

int dsu_ray_aabb(const float* rays_o, const float* rays_d, int64_t n_rays, const float* aabb6,
                 const float* jitter, float step, float* t_min, float* t_max, void* stream) {
  if (n_rays < 0 || !aabb6 || (n_rays && (!rays_o || !rays_d || !t_min || !t_max)))
    return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  ray_aabb_kernel<<<dsu_blocks_for(n_rays, 256), 256, 0, (hipStream_t)stream>>>(
      rays_o, rays_d, n_rays, make_aabb(aabb6), jitter, step, t_min, t_max);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ray_march_count(const float* rays_o, const float* rays_d, const float* t_min,
                        const float* t_max, int64_t n_rays, const float* aabb6,
                        const uint8_t* occ_binary, int32_t res, float step,
                        int32_t* num_steps, void* stream) {
  if (n_rays < 0 || !aabb6 || !(step > 0.0f) || (occ_binary && res <= 0)) return DSU_EINVAL;
  if (n_rays && (!rays_o || !rays_d || !t_min || !t_max || !num_steps)) return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  ray_march_kernel<false><<<dsu_blocks_for(n_rays, 64), 64, 0, (hipStream_t)stream>>>(
      rays_o, rays_d, t_min, t_max, n_rays, make_aabb(aabb6, occ_binary ? res : 1), occ_binary, res, step, nullptr,
      num_steps, nullptr, nullptr, nullptr);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ray_march_fill(const float* rays_o, const float* rays_d, const float* t_min,
                       const float* t_max, int64_t n_rays, const float* aabb6,
                       const uint8_t* occ_binary, int32_t res, float step,
                       const int32_t* offsets, int64_t* ray_indices, float* t_starts,
                       float* t_ends, void* stream) {
  if (n_rays < 0 || !aabb6 || !(step > 0.0f) || (occ_binary && res <= 0)) return DSU_EINVAL;
  if (n_rays && (!rays_o || !rays_d || !t_min || !t_max || !offsets)) return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  ray_march_kernel<true><<<dsu_blocks_for(n_rays, 64), 64, 0, (hipStream_t)stream>>>(
      rays_o, rays_d, t_min, t_max, n_rays, make_aabb(aabb6, occ_binary ? res : 1), occ_binary, res, step, offsets,
      nullptr, ray_indices, t_starts, t_ends);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_weights_from_alpha_fwd(const float* alpha, const int32_t* offsets, const int32_t* counts,
                               int64_t n_rays, float* weights, void* stream) {
  if (n_rays < 0 || (n_rays && (!offsets || !counts))) return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  weights_fwd_kernel<<<dsu_blocks_for(n_rays, 64), 64, 0, (hipStream_t)stream>>>(
      alpha, offsets, counts, n_rays, weights);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_weights_from_alpha_bwd(const float* alpha, const float* weights, const float* d_weights,
                               const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                               float* d_alpha, void* stream) {
  if (n_rays < 0 || (n_rays && (!offsets || !counts))) return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  weights_bwd_kernel<<<dsu_blocks_for(n_rays, 64), 64, 0, (hipStream_t)stream>>>(
      alpha, weights, d_weights, offsets, counts, n_rays, d_alpha);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_accumulate_fwd(const float* weights, const float* values, int32_t channels,
                       const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                       float* out, void* stream) {
  if (n_rays < 0 || channels <= 0 || (n_rays && (!offsets || !counts || !out)))
    return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  accumulate_fwd_kernel<<<dsu_blocks_for(n_rays * channels, 64), 64, 0, (hipStream_t)stream>>>(
      weights, values, channels, offsets, counts, n_rays, out);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_occgrid_ema(float* occs, const int64_t* idx, const float* occ, int64_t n, float decay,
                    void* stream) {
  if (n < 0 || (n && (!occs || !occ))) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  occ_ema_kernel<<<dsu_capped_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(occs, idx, occ, n,
                                                                            decay);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_occgrid_binarize(const float* occs, int64_t n_cells, float thre, uint8_t* binary,
                         void* stream) {
  if (n_cells < 0 || (n_cells && (!occs || !binary))) return DSU_EINVAL;
  if (n_cells == 0) return DSU_OK;
  occ_bin_kernel<<<dsu_capped_blocks(n_cells, 256), 256, 0, (hipStream_t)stream>>>(
      occs, n_cells, thre, binary);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
