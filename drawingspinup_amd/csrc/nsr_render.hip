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
#include <stdlib.h>

namespace {

struct Aabb {
  float mn[3];
  float mx[3];
  float inv_ext[3];  // 1/(mx-mn), only used when pow2 != 0
  int pow2;          // every extent and the grid resolution are powers of two: x/ext == x*(1/ext)
                     // bit for bit, so the serial marching chain can avoid IEEE divisions
};

template <bool P2>
__device__ __forceinline__ float div_ext(float x, const Aabb& a, int d) {
  return P2 ? x * a.inv_ext[d] : x / (a.mx[d] - a.mn[d]);
}
template <bool P2>
__device__ __forceinline__ float div_res(float x, const Aabb& a, int res, float inv_res) {
  return P2 ? x * inv_res : x / (float)res;
}

__device__ __forceinline__ float signf1(float x) { return copysignf(1.0f, x); }

// grid_occupied_at (nerfacc): false outside the roi, else the binary cell of the point.
// Split in two so that the marcher can issue many lookups back to back: occ_cell() is pure
// arithmetic (cell index clamped into range, so the load that follows is unconditional) and the
// caller combines the loaded byte with `inside`.
template <bool P2>
__device__ __forceinline__ int occ_cell(const float p[3], const Aabb& a, int res, bool& inside) {
  inside = !(p[0] < a.mn[0] || p[0] > a.mx[0] || p[1] < a.mn[1] || p[1] > a.mx[1] ||
             p[2] < a.mn[2] || p[2] > a.mx[2]);
  int ix[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float u = div_ext<P2>(p[d] - a.mn[d], a, d);
    u = inside ? u : 0.0f;                       // keep the int conversion defined outside the roi
    int i = (int)(u * (float)res);
    ix[d] = min(max(i, 0), res - 1);
  }
  return (ix[0] * res + ix[1]) * res + ix[2];
}

template <bool P2>
__device__ __forceinline__ float dist_to_next_voxel(const float p[3], const float dir[3],
                                                    const float inv_dir[3], const Aabb& a,
                                                    int res) {
  float t = INFINITY;
  const float inv_res = 1.0f / (float)res;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float g = div_ext<P2>(p[d] - a.mn[d], a, d) * (float)res;
    float td = div_res<P2>((floorf(g + 0.5f + 0.5f * signf1(dir[d])) - g) * inv_dir[d], a, res,
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

// advance_to_next_voxel (nerfacc): from the rejected lattice point tm, keep the sample lattice and
// move to the first lattice point at or beyond the next voxel boundary (pure arithmetic).
template <bool P2>
__device__ __forceinline__ float skip_voxel(float tm, const float o[3], const float d[3],
                                            const float inv[3], const Aabb& a, int res, float dt,
                                            float far) {
  const float p[3] = {o[0] + tm * d[0], o[1] + tm * d[1], o[2] + tm * d[2]};
  float target = tm + dist_to_next_voxel<P2>(p, d, inv, a, res);
  target = fminf(target, far);
  float tt = tm;
  do { tt += dt; } while (tt < target);
  return tt;
}

// MODE 0: count only; 1: fill at offsets[i]; 2: single pass — fill a fixed-capacity scratch row
// (i * cap) AND write the count, so the serial march runs once (compacted by ray_compact_kernel)
//
// One ray per lane, ~600 lattice points per ray, and the serial loop is ONE dependent occupancy
// byte load (~250 ns measured, L2 hit) per lattice point or per skipped voxel: pure latency with
// 32 waves on 256 CUs.  Both continuations of the recurrence are therefore computed ahead with
// the same f32 operations the serial loop would execute (so the bits are the same):
//   chain A: the next SPEC lattice points assuming every one is accepted,
//   chain B: the next SB voxel skips assuming every landing point is rejected,
// all SPEC+SB-1 occupancy bytes are fetched as independent loads, and the points are consumed in
// order along whichever chain the first byte selects until the prediction fails.
template <int MODE, bool HAS_OCC, bool P2, int SB>
__global__ void ray_march_kernel(const float* __restrict__ ro, const float* __restrict__ rd,
                                 const float* __restrict__ tmin, const float* __restrict__ tmax,
                                 int64_t n, Aabb a, const uint8_t* __restrict__ occ, int res,
                                 float step, const int32_t* __restrict__ offsets, int cap,
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
  if (MODE == 1) base = offsets[i];
  if (MODE == 2) base = i * (int64_t)cap;
  const float dt = step;  // cone_angle == 0: calc_dt clamps to dt_min
  int j = 0;
  float t0 = near, t1 = t0 + dt, tm = (t0 + t1) * 0.5f;
  constexpr int SPEC = 8;
  while (tm < far) {
    // ---- chain A: lattice points under "accepted"
    float At0[SPEC], At1[SPEC], Atm[SPEC];
    float a0 = t0, a1 = t1, am = tm;
#pragma unroll
    for (int b = 0; b < SPEC; ++b) {
      At0[b] = a0; At1[b] = a1; Atm[b] = am;
      a0 = a1;
      a1 = a0 + dt;
      am = (a0 + a1) * 0.5f;
    }
    bool Aocc[SPEC];
    if (HAS_OCC) {
      int Ai[SPEC];
      bool Ain[SPEC];
#pragma unroll
      for (int b = 0; b < SPEC; ++b) {
        const float p[3] = {o[0] + Atm[b] * d[0], o[1] + Atm[b] * d[1], o[2] + Atm[b] * d[2]};
        Ai[b] = occ_cell<P2>(p, a, res, Ain[b]);
      }
      uint32_t Av[SPEC];
#pragma unroll
      for (int b = 0; b < SPEC; ++b) Av[b] = occ[Ai[b]];
      // ---- chain B: landing points of successive voxel skips under "rejected"
      float Btm[SB + 1];
      uint32_t Bv[SB];
      bool Bin[SB];
      Btm[0] = tm;
      Bv[0] = 0;
      Bin[0] = false;
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        Btm[k + 1] = skip_voxel<P2>(Btm[k], o, d, inv, a, res, dt, far);
        if (k + 1 < SB) {
          const float t = Btm[k + 1];
          const float p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
          const int ci = occ_cell<P2>(p, a, res, Bin[k + 1]);
          Bv[k + 1] = occ[ci];
        }
      }
#pragma unroll
      for (int b = 0; b < SPEC; ++b) Aocc[b] = Ain[b] && Av[b] != 0;
      if (!Aocc[0]) {
        // the current point is rejected: walk the skip chain while the landing points stay
        // empty; stop on the first one that is occupied or past the far plane (it becomes the
        // current point of the next round)
        float nt = Btm[SB];
        bool alive = true;
#pragma unroll
        for (int k = 1; k < SB; ++k) {
          if (alive && (!(Btm[k] < far) || (Bin[k] && Bv[k] != 0))) {
            nt = Btm[k];
            alive = false;
          }
        }
        tm = nt;
        t0 = tm - dt * 0.5f;
        t1 = tm + dt * 0.5f;
        continue;
      }
    } else {
#pragma unroll
      for (int b = 0; b < SPEC; ++b) Aocc[b] = true;
    }
    bool alive = true;
#pragma unroll
    for (int b = 0; b < SPEC; ++b) {
      if (alive) {
        if (!(Atm[b] < far)) {          // the serial loop's exit test
          t0 = At0[b]; t1 = At1[b]; tm = Atm[b];
          alive = false;
        } else if (Aocc[b]) {
          if (MODE == 1 || (MODE == 2 && j < cap)) {
            t_starts[base + j] = At0[b];
            t_ends[base + j] = At1[b];
            if (MODE == 1) ray_indices[base + j] = i;
          }
          ++j;
        } else {
          tm = skip_voxel<P2>(Atm[b], o, d, inv, a, res, dt, far);
          t0 = tm - dt * 0.5f;
          t1 = tm + dt * 0.5f;
          alive = false;
        }
      }
    }
    if (alive) { t0 = a0; t1 = a1; tm = am; }   // all SPEC points accepted
  }
  if (MODE != 1) num_steps[i] = j;
}

// one wave per ray: scratch rows -> packed samples
__global__ __launch_bounds__(256) void ray_compact_kernel(const float* __restrict__ t0s,
                                                          const float* __restrict__ t1s, int cap,
                                                          const int32_t* __restrict__ off,
                                                          const int32_t* __restrict__ cnt,
                                                          int64_t n_rays,
                                                          int64_t* __restrict__ ray_indices,
                                                          float* __restrict__ t_starts,
                                                          float* __restrict__ t_ends) {
  const int lane = threadIdx.x & 63;
  const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const int64_t b = off[r];
  const int c = cnt[r];
  for (int j = lane; j < c; j += 64) {
    t_starts[b + j] = t0s[r * cap + j];
    t_ends[b + j] = t1s[r * cap + j];
    ray_indices[b + j] = r;
  }
}

// compaction that also emits the sample positions the reference builds with
// rays_o[ray_indices] + rays_d[ray_indices] * ((t_starts + t_ends) / 2)   (neus.py:131-134)
__global__ __launch_bounds__(256) void ray_compact_points_kernel(
    const float* __restrict__ t0s, const float* __restrict__ t1s, int cap,
    const int32_t* __restrict__ off, const int32_t* __restrict__ cnt, int64_t n_rays,
    const float* __restrict__ ro, const float* __restrict__ rd, float* __restrict__ t_starts,
    float* __restrict__ t_ends, float* __restrict__ positions, int64_t out_cap) {
  const int lane = threadIdx.x & 63;
  const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const int64_t b = off[r];
  int c = cnt[r];
  // fixed-capacity outputs (prefetch path: the total is not known to the host yet): rows beyond
  // the capacity are dropped here and the caller, who sees the total later, re-packs
  if (b + c > out_cap) c = b < out_cap ? (int)(out_cap - b) : 0;
  const float o[3] = {ro[r * 3], ro[r * 3 + 1], ro[r * 3 + 2]};
  const float d[3] = {rd[r * 3], rd[r * 3 + 1], rd[r * 3 + 2]};
  for (int j = lane; j < c; j += 64) {
    const float t0 = t0s[r * cap + j], t1 = t1s[r * cap + j];
    t_starts[b + j] = t0;
    t_ends[b + j] = t1;
    const float mid = (t0 + t1) / 2.0f;
    positions[(b + j) * 3 + 0] = o[0] + d[0] * mid;
    positions[(b + j) * 3 + 1] = o[1] + d[1] * mid;
    positions[(b + j) * 3 + 2] = o[2] + d[2] * mid;
  }
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

__global__ void occ_ema_kernel(float* __restrict__ occs, const float* __restrict__ occ, int64_t n,
                               float decay) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    occs[i] = fmaxf(occs[i] * decay, occ[i]);
}

// Indexed form in three passes, so that a cell listed twice is decayed once: candidates from the
// untouched grid, the listed cells to zero, atomic max of the candidates (values >= 0: the bit
// pattern orders like the value).
__global__ void occ_ema_candidates_kernel(const float* __restrict__ occs,
                                          const int64_t* __restrict__ idx,
                                          const float* __restrict__ occ, int64_t n, float decay,
                                          float* __restrict__ cand) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    cand[i] = fmaxf(fmaxf(occs[idx[i]] * decay, occ[i]), 0.0f);
}

__global__ void occ_ema_clear_kernel(float* __restrict__ occs, const int64_t* __restrict__ idx,
                                     int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    occs[idx[i]] = 0.0f;
}

__global__ void occ_ema_max_kernel(float* __restrict__ occs, const int64_t* __restrict__ idx,
                                   const float* __restrict__ cand, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    atomicMax(reinterpret_cast<int*>(&occs[idx[i]]), __float_as_int(cand[i]));
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


// ---------------------------------------------------------------------------------------------
// Fused NeuS shading + compositing (neus.py:90-112 get_alpha, :143-153 weights/accumulate).
// One wave per ray; lanes walk the ray's samples 64 at a time; transmittance by a wave-level
// product scan carried across chunks.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_incl_scan_mul(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o);
    if (lane >= o) v *= t;
  }
  return v;
}
__device__ __forceinline__ float wave_incl_scan_add(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct AlphaOut {
  float alpha, pc, nc, iter_cos, true_cos;
  bool clipped;
};

__device__ __forceinline__ AlphaOut neus_alpha(float sdf, const float n[3], const float d[3],
                                               float dist, float inv_s, float car) {
  AlphaOut a;
  a.true_cos = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];
  a.iter_cos = -(fmaxf(-a.true_cos * 0.5f + 0.5f, 0.0f) * (1.0f - car) +
                 fmaxf(-a.true_cos, 0.0f) * car);
  const float nx = sdf + a.iter_cos * dist * 0.5f, pv = sdf - a.iter_cos * dist * 0.5f;
  a.pc = sigmoidf_(pv * inv_s);
  a.nc = sigmoidf_(nx * inv_s);
  const float raw = ((a.pc - a.nc) + 1e-5f) / (a.pc + 1e-5f);
  a.clipped = raw < 0.0f || raw > 1.0f;
  a.alpha = fminf(fmaxf(raw, 0.0f), 1.0f);
  return a;
}

// outputs per ray: opacity(1) depth(1) rgb(3) normal(3) -> comp (R,8); per sample: alpha, weights
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float* __restrict__ sdf, const float* __restrict__ normal,
    const float* __restrict__ rgb, const float* __restrict__ rays_d,
    const float* __restrict__ t_starts, const float* __restrict__ t_ends,
    const int32_t* __restrict__ off, const int32_t* __restrict__ cnt, int64_t n_rays,
    const float* __restrict__ inv_s_p, float car, float* __restrict__ alpha_o,
    float* __restrict__ w_o, float* __restrict__ comp) {
  const float inv_s = fminf(fmaxf(*inv_s_p, 1e-6f), 1e6f);   // .clip(1e-6, 1e6)  (neus.py:91)
  const int lane = threadIdx.x & 63;
  const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rays) return;
  const int64_t b = off[r];
  const int c = cnt[r];
  const float d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
  float T = 1.0f;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
  for (int j0 = 0; j0 < c; j0 += 64) {
    const int j = j0 + lane;
    const bool v = j < c;
    float al = 0.0f, mid = 0.0f, n[3] = {0.f, 0.f, 0.f}, col[3] = {0.f, 0.f, 0.f};
    if (v) {
      const int64_t i = b + j;
      n[0] = normal[i * 3]; n[1] = normal[i * 3 + 1]; n[2] = normal[i * 3 + 2];
      const float ts = t_starts[i], te = t_ends[i];
      mid = (ts + te) / 2.0f;
      al = neus_alpha(sdf[i], n, d, te - ts, inv_s, car).alpha;
      col[0] = rgb[i * 3]; col[1] = rgb[i * 3 + 1]; col[2] = rgb[i * 3 + 2];
    }
    const float incl = wave_incl_scan_mul(1.0f - al, lane);
    float excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0f;
    const float w = al * (T * excl);
    T *= __shfl(incl, 63);
    if (v) {
      alpha_o[b + j] = al;
      w_o[b + j] = w;
      acc[0] += w;
      acc[1] += w * mid;
      acc[2] += w * col[0]; acc[3] += w * col[1]; acc[4] += w * col[2];
      acc[5] += w * n[0]; acc[6] += w * n[1]; acc[7] += w * n[2];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = wave_sum(acc[k]);
  if (lane < 8) {
    float v = acc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) v = lane == k ? acc[k] : v;
    comp[r * 8 + lane] = v;
  }
}

// d_comp (R,8) -> d_sdf (N), d_normal (N,3), d_rgb (N,3); d_inv_s accumulated (1 float)
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float* __restrict__ sdf, const float* __restrict__ normal,
    const float* __restrict__ rgb, const float* __restrict__ rays_d,
    const float* __restrict__ t_starts, const float* __restrict__ t_ends,
    const int32_t* __restrict__ off, const int32_t* __restrict__ cnt, int64_t n_rays,
    const float* __restrict__ inv_s_p, float car, const float* __restrict__ alpha_s,
    const float* __restrict__ w_s,
    const float* __restrict__ d_comp, const float* __restrict__ d_w_extra,
    float* __restrict__ d_sdf, float* __restrict__ d_normal, float* __restrict__ d_rgb,
    float* __restrict__ d_inv_s) {
  const float inv_raw = *inv_s_p;
  const float inv_s = fminf(fmaxf(inv_raw, 1e-6f), 1e6f);
  const bool inv_pass = inv_raw >= 1e-6f && inv_raw <= 1e6f;   // clip passes the gradient inside
  const int lane = threadIdx.x & 63;
  const int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  float dinv = 0.0f;
  if (r < n_rays) {
    const int64_t b = off[r];
    const int c = cnt[r];
    const float d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
    float dc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dc[k] = d_comp[r * 8 + k];
    // pass 1: total = sum_j gw_j * w_j
    float total = 0.0f;
    for (int j = lane; j < c; j += 64) {
      const int64_t i = b + j;
      const float mid = (t_starts[i] + t_ends[i]) / 2.0f;
      float gw = dc[0] + dc[1] * mid + dc[2] * rgb[i * 3] + dc[3] * rgb[i * 3 + 1] +
                 dc[4] * rgb[i * 3 + 2] + dc[5] * normal[i * 3] + dc[6] * normal[i * 3 + 1] +
                 dc[7] * normal[i * 3 + 2];
      if (d_w_extra) gw += d_w_extra[i];
      total += gw * w_s[i];
    }
    total = wave_sum(total);
    // pass 2: nerfacc's weight backward, prefix form
    float T = 1.0f, done = 0.0f;
    for (int j0 = 0; j0 < c; j0 += 64) {
      const int j = j0 + lane;
      const bool v = j < c;
      float al = 0.0f, w = 0.0f, gw = 0.0f, n[3] = {0.f, 0.f, 0.f};
      int64_t i = b + (v ? j : 0);
      float mid = 0.0f, dist = 0.0f, s = 0.0f;
      if (v) {
        al = alpha_s[i]; w = w_s[i];
        n[0] = normal[i * 3]; n[1] = normal[i * 3 + 1]; n[2] = normal[i * 3 + 2];
        const float ts = t_starts[i], te = t_ends[i];
        mid = (ts + te) / 2.0f; dist = te - ts; s = sdf[i];
        gw = dc[0] + dc[1] * mid + dc[2] * rgb[i * 3] + dc[3] * rgb[i * 3 + 1] +
             dc[4] * rgb[i * 3 + 2] + dc[5] * n[0] + dc[6] * n[1] + dc[7] * n[2];
        if (d_w_extra) gw += d_w_extra[i];
      }
      const float incl = wave_incl_scan_mul(1.0f - al, lane);
      float excl = __shfl_up(incl, 1);
      if (lane == 0) excl = 1.0f;
      const float Tj = T * excl;
      const float gww = gw * w;
      const float pin = wave_incl_scan_add(gww, lane);      // inclusive prefix of gw*w
      const float before = done + (pin - gww);              // sum_{k<j}
      T *= __shfl(incl, 63);
      done += __shfl(pin, 63);
      if (v) {
        const float accum = total - before;                 // sum_{k>=j} gw_k w_k
        const float galpha = (gw * Tj - accum) / fmaxf(1.0f - al, 1e-10f);
        // alpha = clip(((pc - nc) + 1e-5) / (pc + 1e-5))
        const AlphaOut a = neus_alpha(s, n, d, dist, inv_s, car);
        float dsdf = 0.0f, dn[3] = {0.f, 0.f, 0.f};
        if (!a.clipped) {
          const float den = a.pc + 1e-5f;
          const float d_pc = galpha * (1.0f / den - ((a.pc - a.nc) + 1e-5f) / (den * den));
          const float d_nc = -galpha / den;
          const float d_pv = d_pc * a.pc * (1.0f - a.pc);   // wrt (prev * inv_s)
          const float d_nx = d_nc * a.nc * (1.0f - a.nc);
          const float pv = s - a.iter_cos * dist * 0.5f, nx = s + a.iter_cos * dist * 0.5f;
          dinv += d_pv * pv + d_nx * nx;
          dsdf = (d_pv + d_nx) * inv_s;
          const float d_ic = (d_nx - d_pv) * inv_s * dist * 0.5f;
          // iter_cos = -(relu(-tc*0.5+0.5)*(1-car) + relu(-tc)*car)
          float d_tc = 0.0f;
          if (-a.true_cos * 0.5f + 0.5f > 0.0f) d_tc += 0.5f * (1.0f - car);
          if (-a.true_cos > 0.0f) d_tc += car;
          d_tc *= d_ic;
          dn[0] = d_tc * d[0]; dn[1] = d_tc * d[1]; dn[2] = d_tc * d[2];
        }
        d_sdf[i] = dsdf;
        d_normal[i * 3] = dn[0] + w * dc[5];
        d_normal[i * 3 + 1] = dn[1] + w * dc[6];
        d_normal[i * 3 + 2] = dn[2] + w * dc[7];
        d_rgb[i * 3] = w * dc[2]; d_rgb[i * 3 + 1] = w * dc[3]; d_rgb[i * 3 + 2] = w * dc[4];
      }
    }
  }
  dinv = wave_sum(dinv);
  if (lane == 0 && dinv != 0.0f && inv_pass) unsafeAtomicAdd(d_inv_s, dinv);
}

// normal = grad / max(|grad|, 1e-12) ; tex_in = [feature(13), normal(3)]   (neus.py:143,145)
__global__ void shade_prep_fwd_kernel(const float* __restrict__ grad,
                                      const float* __restrict__ feat, int64_t n,
                                      float* __restrict__ normal, float* __restrict__ tex_in) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g0 = grad[i * 3], g1 = grad[i * 3 + 1], g2 = grad[i * 3 + 2];
  const float inv = 1.0f / fmaxf(sqrtf(g0 * g0 + g1 * g1 + g2 * g2), 1e-12f);
  const float n0 = g0 * inv, n1 = g1 * inv, n2 = g2 * inv;
  normal[i * 3] = n0; normal[i * 3 + 1] = n1; normal[i * 3 + 2] = n2;
#pragma unroll
  for (int k = 0; k < 13; ++k) tex_in[i * 16 + k] = feat[i * 13 + k];
  tex_in[i * 16 + 13] = n0; tex_in[i * 16 + 14] = n1; tex_in[i * 16 + 15] = n2;
}

// d_grad = (dn - n (n.dn)) / |grad| with dn = d_normal + d_tex_in[13:16]; d_feat = d_tex_in[:13]
__global__ void shade_prep_bwd_kernel(const float* __restrict__ grad,
                                      const float* __restrict__ d_normal,
                                      const float* __restrict__ d_tex_in, int64_t n,
                                      int64_t tail, float* __restrict__ d_grad,
                                      float* __restrict__ d_feat) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) {
    // rows of points that are not ray samples (the regulariser points evaluated in the same
    // geometry launch): no feature gradient reaches them
    if (i < n + tail)
      for (int k = 0; k < 13; ++k) d_feat[i * 13 + k] = 0.0f;
    return;
  }
  const float g0 = grad[i * 3], g1 = grad[i * 3 + 1], g2 = grad[i * 3 + 2];
  const float len = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
  const float inv = 1.0f / fmaxf(len, 1e-12f);
  const float n0 = g0 * inv, n1 = g1 * inv, n2 = g2 * inv;
  float d0 = d_tex_in[i * 16 + 13], d1 = d_tex_in[i * 16 + 14], d2 = d_tex_in[i * 16 + 15];
  if (d_normal) { d0 += d_normal[i * 3]; d1 += d_normal[i * 3 + 1]; d2 += d_normal[i * 3 + 2]; }
  const float dot = n0 * d0 + n1 * d1 + n2 * d2;
  const bool ok = len > 1e-12f;
  d_grad[i * 3] = ok ? (d0 - n0 * dot) * inv : d0 * inv;
  d_grad[i * 3 + 1] = ok ? (d1 - n1 * dot) * inv : d1 * inv;
  d_grad[i * 3 + 2] = ok ? (d2 - n2 * dot) * inv : d2 * inv;
#pragma unroll
  for (int k = 0; k < 13; ++k) d_feat[i * 13 + k] = d_tex_in[i * 16 + k];
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

// march kernel variant: occupancy grid present or not, power-of-two box (exact reciprocal
// multiplies instead of IEEE divisions), skip-chain length (DSU_MARCH_SB=4|8, default 4)
static int march_sb() {
  static int v = 0;
  if (!v) v = dsu_ab_int("DSU_MARCH_SB", 4) == 8 ? 8 : 4;
  return v;
}
// rays per workgroup (= lanes of the one wave that marches them): DSU_MARCH_THREADS=16|32|64.  A
// wave executes the union of its rays' control paths round by round, so fewer rays per wave means
// fewer rounds (and more, shorter waves).  16: with the march of every 16th step on the critical
// path (the step behind an occupancy refresh cannot prefetch its samples) the NSR stage measured
// 1.152 ms per step against 1.175 with 64 (three interleaved runs each, same box).
static int march_threads() {
  static int v = 0;
  if (!v) {
    const int t = dsu_ab_int("DSU_MARCH_THREADS", 16);
    v = (t == 64 || t == 32) ? t : 16;
  }
  return v;
}
#define DSU_MARCH_ONE(M, O, P, S, nr, st, ...)                                              \
  ray_march_kernel<M, O, P, S><<<dsu_blocks_for(nr, march_threads()), march_threads(), 0,   \
                                 (hipStream_t)st>>>(__VA_ARGS__)
#define DSU_LAUNCH_MARCH(M, nr, st, box, occp, ...)                                   \
  do {                                                                                \
    const bool p2_ = (box).pow2 != 0;                                                 \
    if (!(occp)) {                                                                    \
      if (p2_) DSU_MARCH_ONE(M, false, true, 4, nr, st, __VA_ARGS__);                 \
      else DSU_MARCH_ONE(M, false, false, 4, nr, st, __VA_ARGS__);                    \
    } else if (march_sb() == 8) {                                                     \
      if (p2_) DSU_MARCH_ONE(M, true, true, 8, nr, st, __VA_ARGS__);                  \
      else DSU_MARCH_ONE(M, true, false, 8, nr, st, __VA_ARGS__);                     \
    } else {                                                                          \
      if (p2_) DSU_MARCH_ONE(M, true, true, 4, nr, st, __VA_ARGS__);                  \
      else DSU_MARCH_ONE(M, true, false, 4, nr, st, __VA_ARGS__);                     \
    }                                                                                 \
  } while (0)

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
  const Aabb box = make_aabb(aabb6, occ_binary ? res : 1);
  DSU_LAUNCH_MARCH(0, n_rays, stream, box, occ_binary, rays_o, rays_d, t_min, t_max,
                   n_rays, box,
                   occ_binary, res, step, nullptr, 0, num_steps, nullptr, nullptr, nullptr);
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
  const Aabb box = make_aabb(aabb6, occ_binary ? res : 1);
  DSU_LAUNCH_MARCH(1, n_rays, stream, box, occ_binary, rays_o, rays_d, t_min, t_max,
                   n_rays, box,
                   occ_binary, res, step, offsets, 0, nullptr, ray_indices, t_starts, t_ends);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ray_march_scratch(const float* rays_o, const float* rays_d, const float* t_min,
                          const float* t_max, int64_t n_rays, const float* aabb6,
                          const uint8_t* occ_binary, int32_t res, float step, int32_t capacity,
                          int32_t* num_steps, float* scratch_t_starts, float* scratch_t_ends,
                          void* stream) {
  if (n_rays < 0 || !aabb6 || !(step > 0.0f) || (occ_binary && res <= 0) || capacity <= 0)
    return DSU_EINVAL;
  if (n_rays && (!rays_o || !rays_d || !t_min || !t_max || !num_steps || !scratch_t_starts ||
                 !scratch_t_ends))
    return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  const Aabb box = make_aabb(aabb6, occ_binary ? res : 1);
  DSU_LAUNCH_MARCH(2, n_rays, stream, box, occ_binary, rays_o, rays_d, t_min, t_max,
                   n_rays, box,
                   occ_binary, res, step, nullptr, capacity, num_steps, nullptr, scratch_t_starts, scratch_t_ends);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ray_compact(const float* scratch_t_starts, const float* scratch_t_ends, int32_t capacity,
                    const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                    int64_t* ray_indices, float* t_starts, float* t_ends, void* stream) {
  if (n_rays < 0 || capacity <= 0) return DSU_EINVAL;
  if (n_rays && (!scratch_t_starts || !scratch_t_ends || !offsets || !counts)) return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  ray_compact_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      scratch_t_starts, scratch_t_ends, capacity, offsets, counts, n_rays, ray_indices, t_starts,
      t_ends);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ray_compact_points_cap(const float* scratch_t_starts, const float* scratch_t_ends,
                               int32_t capacity, const int32_t* offsets, const int32_t* counts,
                               int64_t n_rays, const float* rays_o, const float* rays_d,
                               float* t_starts, float* t_ends, float* positions,
                               int64_t out_capacity, void* stream) {
  if (n_rays < 0 || capacity <= 0 || out_capacity < 0) return DSU_EINVAL;
  if (n_rays && (!scratch_t_starts || !scratch_t_ends || !offsets || !counts || !rays_o ||
                 !rays_d || !t_starts || !t_ends || !positions))
    return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  ray_compact_points_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      scratch_t_starts, scratch_t_ends, capacity, offsets, counts, n_rays, rays_o, rays_d,
      t_starts, t_ends, positions, out_capacity);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ray_compact_points(const float* scratch_t_starts, const float* scratch_t_ends,
                           int32_t capacity, const int32_t* offsets, const int32_t* counts,
                           int64_t n_rays, const float* rays_o, const float* rays_d,
                           float* t_starts, float* t_ends, float* positions, void* stream) {
  return dsu_ray_compact_points_cap(scratch_t_starts, scratch_t_ends, capacity, offsets, counts,
                                    n_rays, rays_o, rays_d, t_starts, t_ends, positions,
                                    INT64_MAX, stream);
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
                    float* scratch, void* stream) {
  if (n < 0 || (n && (!occs || !occ)) || (n && idx && !scratch)) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = dsu_capped_blocks(n, 256);
  if (!idx) {
    occ_ema_kernel<<<blocks, 256, 0, s>>>(occs, occ, n, decay);
  } else {
    occ_ema_candidates_kernel<<<blocks, 256, 0, s>>>(occs, idx, occ, n, decay, scratch);
    occ_ema_clear_kernel<<<blocks, 256, 0, s>>>(occs, idx, n);
    occ_ema_max_kernel<<<blocks, 256, 0, s>>>(occs, idx, scratch, n);
  }
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

int dsu_neus_composite_fwd(const float* sdf, const float* normal, const float* rgb,
                           const float* rays_d, const float* t_starts, const float* t_ends,
                           const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                           const float* inv_s, float cos_anneal_ratio, float* alpha,
                           float* weights, float* comp, void* stream) {
  if (n_rays < 0 || !inv_s || (n_rays && (!rays_d || !offsets || !counts || !comp)))
    return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  composite_fwd_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      sdf, normal, rgb, rays_d, t_starts, t_ends, offsets, counts, n_rays, inv_s,
      cos_anneal_ratio, alpha, weights, comp);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_neus_composite_bwd(const float* sdf, const float* normal, const float* rgb,
                           const float* rays_d, const float* t_starts, const float* t_ends,
                           const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                           const float* inv_s, float cos_anneal_ratio, const float* alpha,
                           const float* weights, const float* d_comp, const float* d_weights,
                           float* d_sdf, float* d_normal, float* d_rgb, float* d_inv_s,
                           void* stream) {
  if (n_rays < 0 || !inv_s || (n_rays && (!rays_d || !offsets || !counts || !d_comp || !d_inv_s)))
    return DSU_EINVAL;
  if (n_rays == 0) return DSU_OK;
  composite_bwd_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      sdf, normal, rgb, rays_d, t_starts, t_ends, offsets, counts, n_rays, inv_s,
      cos_anneal_ratio, alpha, weights, d_comp, d_weights, d_sdf, d_normal, d_rgb, d_inv_s);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_shade_prep_fwd(const float* grad, const float* feature, int64_t n, float* normal,
                       float* tex_in, void* stream) {
  if (n < 0 || (n && (!grad || !feature || !normal || !tex_in))) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  shade_prep_fwd_kernel<<<dsu_blocks_for(n, 256), 256, 0, (hipStream_t)stream>>>(grad, feature, n,
                                                                               normal, tex_in);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_shade_prep_bwd(const float* grad, const float* d_normal, const float* d_tex_in, int64_t n,
                       float* d_grad, float* d_feature, void* stream) {
  return dsu_shade_prep_bwd_tail(grad, d_normal, d_tex_in, n, 0, d_grad, d_feature, stream);
}

int dsu_shade_prep_bwd_tail(const float* grad, const float* d_normal, const float* d_tex_in,
                            int64_t n, int64_t tail_rows, float* d_grad, float* d_feature,
                            void* stream) {
  if (n < 0 || tail_rows < 0 || (n && (!grad || !d_tex_in || !d_grad)) ||
      ((n || tail_rows) && !d_feature))
    return DSU_EINVAL;
  if (n + tail_rows == 0) return DSU_OK;
  shade_prep_bwd_kernel<<<dsu_blocks_for(n + tail_rows, 256), 256, 0, (hipStream_t)stream>>>(
      grad, d_normal, d_tex_in, n, tail_rows, d_grad, d_feature);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
