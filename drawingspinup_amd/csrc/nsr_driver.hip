// Native driver of one NSR optimisation step (gfx950 host code + three small kernels).
//
// OrthoNeuSSystem.training_step (2_charactor_reconstructor/instant_nsr/systems/neus_ortho.py:79-169)
// over NeuSModelTextureMLP.forward_ (instant_nsr/models/neus.py:114-196), preprocess_data
// (neus_ortho.py:26-77) and the AdamW step of configs/neuralangelo-ortho-wmask.yaml:96-127 as ONE
// call of the C ABI.  The kernels are the library's own (dsu_sdf_fd_*_sorted, dsu_texture_*,
// dsu_neus_composite_*, dsu_ray_*, dsu_*_losses ...); what this file adds is the host sequencing
// that used to be ~1.4 ms of Python per step — as long as a step's device time — plus
//   * nsr_draw_kernel: the step's random draws (view / pixel triples, stratified jitter, the 2048
//     regulariser points and their perturbation: neus_ortho.py:31-41, neus.py:155-160) from a
//     counter-based Philox4x32-10 stream: one launch instead of six torch RNG launches;
//   * small_update_kernel: weight-norm backward (network_utils.py:130-131), grad of the variance
//     scalar, torch.optim.AdamW for the 13 small tensors of the three parameter groups, weight-norm
//     forward and inv_s = exp(10 variance) for the NEXT step, and the reset of the small gradient
//     accumulators: one single-workgroup launch instead of ~12 torch launches.
//
// Streams: the step's kernels run on `main`; the next step's draws, ray batch, march, packing and
// Morton sort run on a driver-owned side stream behind the geometry forward (they depend on the
// occupancy grid and the RNG, not on the parameters), and hand the sample total to the host
// through pinned memory.  All device memory is the caller's: one workspace carved here.
#include "common.h"
#include "adamw_dev.h"
// Priority of the driver's side stream (march + packing of the next step's samples): 1 high, 2 normal, 0 low —
// process-wide, dsu_set_nsr_side_stream_priority (default high).  Found at the end of round 6: a driver is created per
// drawing, and from the fourth side stream of a process on a stream of NON-DEFAULT priority (high or low alike) puts the
// whole drawing into a slow mode — every kernel of the main stream 10-100 % longer (geometry forward 116 -> 135 us, texture
// forward 50 -> 90, table update 26 -> 52, the march itself 357 -> 331), the optimisation 2.31 -> 3.11 s; at normal
// priority 10 of 10 drawings fast, 6 of 15 slow otherwise (profiles/round6_side_stream_priority.txt; it lives in the
// runtime's queues for such streams, not in this code).  One drawing at a time therefore wants NORMAL (bench.py
// --inflight 1 sets it).  With three drawings in flight the bench line measured 0.317 / 0.337 at normal against
// 0.341 / 0.347 at high (different boxes; the optimisation loses its precedence over the other drawings' diffusion
// launches: 5.7 vs 4.8 s per drawing in the stage, the diffusion 1.8 vs 2.4): the in-flight default stays high.

#include <math.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace {

constexpr int N_GEO = 64 * 23 + 64 + 13 * 64 + 13;            // effective-weight gradients of the SDF MLP
constexpr int N_TEX = 64 * 16 + 64 + 64 * 64 + 64 + 3 * 64 + 3;
constexpr int N_SMALL = (64 * 23 + 64 + 64) + (13 * 64 + 13 + 13) + N_TEX + 1;   // optimizer elements

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11): counter (index, stream, step lo, step hi), key = seed
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void nsr_draw_kernel(uint64_t seed, int64_t step, int32_t n_rays,
                                                       int32_t V, int32_t H, int32_t W,
                                                       int64_t* __restrict__ index,
                                                       int64_t* __restrict__ px,
                                                       int64_t* __restrict__ py,
                                                       float* __restrict__ jitter, int32_t n_random,
                                                       float* __restrict__ pts_random,
                                                       float* __restrict__ perturb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t s0 = (uint32_t)step, s1 = (uint32_t)((uint64_t)step >> 32);
  if (i < n_rays) {
    const uint4 r = philox(make_uint4((uint32_t)i, 0u, s0, s1), key);
    index[i] = (int64_t)(r.x % (uint32_t)V);          // torch.randint(0, n_views)
    px[i] = (int64_t)(r.y % (uint32_t)W);
    py[i] = (int64_t)(r.z % (uint32_t)H);
    jitter[i] = u01(r.w);
  }
  if (i < n_random) {
    const uint4 a = philox(make_uint4((uint32_t)i, 1u, s0, s1), key);
    const uint4 b = philox(make_uint4((uint32_t)i, 2u, s0, s1), key);
    pts_random[3 * i] = u01(a.x) * 2.0f - 1.0f;         // torch.rand([2048, 3]) * 2 - 1
    pts_random[3 * i + 1] = u01(a.y) * 2.0f - 1.0f;
    pts_random[3 * i + 2] = u01(a.z) * 2.0f - 1.0f;
    // torch.randn_like: Box-Muller on (0, 1] uniforms
    const float u1 = 1.0f - u01(b.x), u2 = u01(b.y), u3 = 1.0f - u01(b.z), u4 = u01(b.w);
    const float r1 = sqrtf(-2.0f * logf(u1)), r2 = sqrtf(-2.0f * logf(u3));
    perturb[3 * i] = r1 * cosf(6.28318530717958648f * u2);
    perturb[3 * i + 1] = r1 * sinf(6.28318530717958648f * u2);
    perturb[3 * i + 2] = r2 * cosf(6.28318530717958648f * u4);
  }
}

// ------------------------------------------------------------------------------------------------
// weight-norm backward + AdamW of the small tensors + weight-norm forward for the next step
// ------------------------------------------------------------------------------------------------
struct SmallArgs {
  // parameters
  float *w0_v, *w0_g, *b0, *w1_v, *w1_g, *b1;
  float* tex[6];
  float* variance;
  // gradients: effective-weight gradients of the SDF MLP [g_w0 | g_b0 | g_w1 | g_b1], texture
  // [w0 b0 w1 b1 w2 b2], d loss / d inv_s — one contiguous block, zeroed here after use
  float* g_geo;
  float* g_tex;
  float* d_inv;
  // optimizer moments, N_SMALL floats each, tensor order: w0_v w0_g b0 w1_v w1_g b1 tex[0..5] variance
  float *m, *v;
  // outputs for the next step
  float *w0_eff, *w1_eff, *inv_s;
  float* zero_terms;   // the three sample-loss accumulators of the NEXT step
  const float* terms;  // this step's eight loss terms ...
  float* terms_out;    // ... and where the caller wants a copy of them (may be null)
  float lr_geo, lr_tex, lr_var, beta1, beta2, eps, wd, bc1, bc2_sqrt;
  int32_t update;   // 0: forward part only (first call)
};

__device__ __forceinline__ void adamw_elem(float& x, float g, float& m, float& v, float lr,
                                           const SmallArgs& a) {
  const float omb1 = (float)(1.0 - (double)a.beta1), omb2 = (float)(1.0 - (double)a.beta2);
  x -= lr * a.wd * x;
  m = m + (g - m) * omb1;
  v = a.beta2 * v + omb2 * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  x -= (lr / a.bc1) * m / denom;
}

// Workgroup 0: the small tensors (below).  Workgroups 1..: AdamW of the hash table's active levels
// (adamw_dev.h) — the two used to be consecutive launches (9 + 18 us); side by side the table update
// disappears behind the single-workgroup chain.
__global__ __launch_bounds__(1024) void small_update_kernel(SmallArgs a, dsu_table_adamw_args ta) {
  if (blockIdx.x > 0) {
    dsu_table_adamw_range(ta, (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x,
                          (int64_t)(gridDim.x - 1) * 1024);
    return;
  }
  // Everything the kernel touches more than once sits in LDS: the first version walked the rows of
  // the weight-norm reductions with dependent global loads from 77 threads (36 us per step for
  // 8 k elements of work).
  __shared__ float s_g[N_GEO];                       // [g_w0 | g_b0 | g_w1 | g_b1] effective-weight grads
  __shared__ float s_v0[64 * 23], s_v1[13 * 64];     // weight_v (updated in place below)
  __shared__ float s_gain[77], s_norm[77], s_dot[77];   // weight_g, |v|_row, <grad_w, v>_row
  const int t = threadIdx.x;
  for (int i = t; i < 64 * 23; i += 1024) s_v0[i] = a.w0_v[i];
  for (int i = t; i < 13 * 64; i += 1024) s_v1[i] = a.w1_v[i];
  if (t < 77) s_gain[t] = t < 64 ? a.w0_g[t] : a.w1_g[t - 64];
  if (a.update)
    for (int i = t; i < N_GEO; i += 1024) s_g[i] = a.g_geo[i];
  __syncthreads();
  if (a.update) {
    // ---- torch._weight_norm_interface_backward (dim 0): rows 0..63 of layer 0, 64..76 of layer 1
    if (t < 77) {
      const bool l0 = t < 64;
      const int r = l0 ? t : t - 64, cols = l0 ? 23 : 64;
      const float* vrow = (l0 ? s_v0 : s_v1) + r * cols;
      const float* grow = (l0 ? s_g : s_g + 64 * 23 + 64) + r * cols;
      float nn = 0.0f, dot = 0.0f;
      for (int c = 0; c < cols; ++c) {
        nn += vrow[c] * vrow[c];
        dot += grow[c] * vrow[c];
      }
      s_norm[t] = sqrtf(nn);
      s_dot[t] = dot;
    }
    __syncthreads();
    // ---- AdamW, tensor by tensor (moments laid out in the same order); updated weight_v /
    // weight_g also go back to LDS for the forward part
    int off = 0;
    for (int i = t; i < 64 * 23; i += 1024) {          // layer-0 weight_v
      const int r = i / 23;
      const float nrm = s_norm[r];
      const float g = (s_gain[r] / nrm) * (s_g[i] - s_v0[i] * s_dot[r] / (nrm * nrm));
      float x = s_v0[i], m = a.m[off + i], v = a.v[off + i];
      adamw_elem(x, g, m, v, a.lr_geo, a);
      a.w0_v[i] = x; a.m[off + i] = m; a.v[off + i] = v;
      s_v0[i] = x;                                     // same thread reads / writes element i
    }
    off += 64 * 23;
    float new_gain = 0.0f;
    if (t < 64) {                                      // layer-0 weight_g
      float x = s_gain[t], m = a.m[off + t], v = a.v[off + t];
      adamw_elem(x, s_dot[t] / s_norm[t], m, v, a.lr_geo, a);
      a.w0_g[t] = x; a.m[off + t] = m; a.v[off + t] = v;
      new_gain = x;
    }
    off += 64;
    if (t < 64) {                                      // layer-0 bias
      float x = a.b0[t], m = a.m[off + t], v = a.v[off + t];
      adamw_elem(x, s_g[64 * 23 + t], m, v, a.lr_geo, a);
      a.b0[t] = x; a.m[off + t] = m; a.v[off + t] = v;
    }
    off += 64;
    for (int i = t; i < 13 * 64; i += 1024) {          // layer-1 weight_v
      const int r = i / 64;
      const float nrm = s_norm[64 + r];
      const float g = (s_gain[64 + r] / nrm) *
                      (s_g[64 * 23 + 64 + i] - s_v1[i] * s_dot[64 + r] / (nrm * nrm));
      float x = s_v1[i], m = a.m[off + i], v = a.v[off + i];
      adamw_elem(x, g, m, v, a.lr_geo, a);
      a.w1_v[i] = x; a.m[off + i] = m; a.v[off + i] = v;
      s_v1[i] = x;
    }
    off += 13 * 64;
    float new_gain1 = 0.0f;
    if (t < 13) {                                      // layer-1 weight_g
      float x = s_gain[64 + t], m = a.m[off + t], v = a.v[off + t];
      adamw_elem(x, s_dot[64 + t] / s_norm[64 + t], m, v, a.lr_geo, a);
      a.w1_g[t] = x; a.m[off + t] = m; a.v[off + t] = v;
      new_gain1 = x;
    }
    off += 13;
    if (t < 13) {                                      // layer-1 bias
      float x = a.b1[t], m = a.m[off + t], v = a.v[off + t];
      adamw_elem(x, s_g[64 * 23 + 64 + 13 * 64 + t], m, v, a.lr_geo, a);
      a.b1[t] = x; a.m[off + t] = m; a.v[off + t] = v;
    }
    off += 13;
    const int tn[6] = {64 * 16, 64, 64 * 64, 64, 3 * 64, 3};
    int go = 0;
    for (int k = 0; k < 6; ++k) {                      // texture MLP
      float* p = a.tex[k];
      for (int i = t; i < tn[k]; i += 1024) {
        float x = p[i], m = a.m[off + i], v = a.v[off + i];
        adamw_elem(x, a.g_tex[go + i], m, v, a.lr_tex, a);
        p[i] = x; a.m[off + i] = m; a.v[off + i] = v;
      }
      off += tn[k];
      go += tn[k];
    }
    if (t == 1023) {
      // d inv_s / d variance = 10 exp(10 variance)
      float x = a.variance[0], m = a.m[off], v = a.v[off];
      const float g = a.d_inv[0] * expf(x * 10.0f) * 10.0f;
      adamw_elem(x, g, m, v, a.lr_var, a);
      a.variance[0] = x; a.m[off] = m; a.v[off] = v;
      a.inv_s[0] = expf(x * 10.0f);
      a.d_inv[0] = 0.0f;
    }
    if (t < 3) a.zero_terms[t] = 0.0f;
    if (t >= 8 && t < 16 && a.terms_out) a.terms_out[t - 8] = a.terms[t - 8];
    __syncthreads();                                   // every read of s_gain / g_tex is done
    if (t < 64) s_gain[t] = new_gain;
    if (t < 13) s_gain[64 + t] = new_gain1;
    // ---- the accumulators of the next step start from zero
    for (int i = t; i < N_GEO; i += 1024) a.g_geo[i] = 0.0f;
    for (int i = t; i < N_TEX; i += 1024) a.g_tex[i] = 0.0f;
    __syncthreads();
  } else if (t == 1023) {
    a.inv_s[0] = expf(a.variance[0] * 10.0f);
  }
  // ---- torch._weight_norm (dim 0): w = g * v / |v|_row
  if (t < 77) {
    const bool l0 = t < 64;
    const int r = l0 ? t : t - 64, cols = l0 ? 23 : 64;
    const float* vrow = (l0 ? s_v0 : s_v1) + r * cols;
    float nn = 0.0f;
    for (int c = 0; c < cols; ++c) nn += vrow[c] * vrow[c];
    s_norm[t] = s_gain[t] / sqrtf(nn);
  }
  __syncthreads();
  for (int i = t; i < 64 * 23; i += 1024) a.w0_eff[i] = s_v0[i] * s_norm[i / 23];
  for (int i = t; i < 13 * 64; i += 1024) a.w1_eff[i] = s_v1[i] * s_norm[64 + i / 64];
}

// ------------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------------
struct Carve {
  char* base;
  int64_t off = 0;
  template <typename T>
  T* take(int64_t count) {
    off = (off + 255) / 256 * 256;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * (int64_t)sizeof(T);
    return p;
  }
};

struct Prefetch {   // everything one step's sample set consists of (three sets, by step % 3)
  int64_t *index, *px, *py;
  float *jitter, *rays, *rgb, *normal, *mask, *cosines, *vw, *tmin, *tmax;
  int32_t *counts, *offsets, *stats;
  float *rays_o, *rays_d;
  float *pts_random, *perturb;
  float *points, *t_starts, *t_ends, *sorted;
  int32_t* perm;
  void* sort_ws;
};

struct Layout {
  Prefetch pf[3];
  float *scratch_ts, *scratch_te;
  float *a_sdf, *a_grad, *a_feat;
  void* enc_cache;
  float *normal, *rgb, *alpha, *w, *comp, *d_comp, *terms;
  float *d_sdf_all, *d_grad_all, *d_feat_all, *d_normal, *d_rgb;
  void *tex_ws, *sdf_ws;
  int32_t* tex_map;                   // dsu_texture_partial_map, uploaded at creation
  uint32_t* tex_mask;                 // (cap_points, 2): layer 1's ReLU pattern, texture forward -> backward
  float *g_geo, *g_tex, *d_inv;      // contiguous: zeroed as one block
  float *w0_eff, *w1_eff, *inv_s, *adam_m, *adam_v;
  int64_t tex_ws_bytes, sdf_ws_bytes, sort_ws_bytes, enc_cache_bytes, total;
};

int32_t march_row_capacity(float radius, float step) {
  const double diag = sqrt(3.0 * (2.0 * radius) * (2.0 * radius));
  return (int32_t)(diag / step) + 8;
}

int carve(const dsu_nsr_driver_cfg& c, char* base, Layout& L) {
  Carve k{base};
  const int64_t R = c.cap_rays, N = c.cap_points, nr = c.n_random, rows = N + 2 * nr;
  const int32_t rowcap = march_row_capacity(c.radius, c.render_step_size);
  L.sort_ws_bytes = c.sort_bits ? dsu_spatial_sort_workspace_bytes(rows, c.sort_bits) : 0;
  L.tex_ws_bytes = dsu_texture_bwd_workspace_bytes(N);
  L.sdf_ws_bytes = dsu_sdf_fd_bwd_workspace_bytes(&c.grid, rows);
  L.enc_cache_bytes = dsu_sdf_fd_enc_cache_bytes(rows, c.grid.n_levels);
  if (L.sort_ws_bytes < 0 || L.tex_ws_bytes < 0 || L.sdf_ws_bytes < 0 || L.enc_cache_bytes < 0)
    return DSU_EINVAL;
  for (int p = 0; p < 3; ++p) {
    Prefetch& f = L.pf[p];
    f.index = k.take<int64_t>(R); f.px = k.take<int64_t>(R); f.py = k.take<int64_t>(R);
    f.jitter = k.take<float>(R); f.rays = k.take<float>(R * 6); f.rgb = k.take<float>(R * 4);
    f.normal = k.take<float>(R * 3); f.mask = k.take<float>(R); f.cosines = k.take<float>(R);
    f.vw = k.take<float>(R); f.tmin = k.take<float>(R); f.tmax = k.take<float>(R);
    f.counts = k.take<int32_t>(R); f.offsets = k.take<int32_t>(R); f.stats = k.take<int32_t>(4);
    f.rays_o = k.take<float>(R * 3); f.rays_d = k.take<float>(R * 3);
    f.pts_random = k.take<float>(nr * 3); f.perturb = k.take<float>(nr * 3);
    f.points = k.take<float>(rows * 3); f.t_starts = k.take<float>(N); f.t_ends = k.take<float>(N);
    f.sorted = k.take<float>(rows * 3); f.perm = k.take<int32_t>(rows);
    f.sort_ws = k.take<char>(L.sort_ws_bytes > 4 ? L.sort_ws_bytes : 4);
  }
  L.scratch_ts = k.take<float>(R * (int64_t)rowcap);
  L.scratch_te = k.take<float>(R * (int64_t)rowcap);
  L.a_sdf = k.take<float>(rows); L.a_grad = k.take<float>(rows * 3); L.a_feat = k.take<float>(rows * 13);
  L.enc_cache = k.take<char>(L.enc_cache_bytes);
  L.normal = k.take<float>(N * 3); L.rgb = k.take<float>(N * 3);
  L.alpha = k.take<float>(N); L.w = k.take<float>(N);
  L.comp = k.take<float>(R * 8); L.d_comp = k.take<float>(R * 8); L.terms = k.take<float>(16);
  L.d_sdf_all = k.take<float>(rows); L.d_grad_all = k.take<float>(rows * 3);
  L.d_feat_all = k.take<float>(rows * 13);
  L.d_normal = k.take<float>(N * 3); L.d_rgb = k.take<float>(N * 3);
  L.tex_ws = k.take<char>(L.tex_ws_bytes > 4 ? L.tex_ws_bytes : 4);
  L.sdf_ws = k.take<char>(L.sdf_ws_bytes > 4 ? L.sdf_ws_bytes : 4);
  L.tex_map = k.take<int32_t>(dsu_texture_partial_map(nullptr));
  L.tex_mask = k.take<uint32_t>(N * 2);
  L.g_geo = k.take<float>(N_GEO + N_TEX + 1);
  L.g_tex = L.g_geo ? L.g_geo + N_GEO : nullptr;
  L.d_inv = L.g_geo ? L.g_geo + N_GEO + N_TEX : nullptr;
  L.w0_eff = k.take<float>(64 * 23); L.w1_eff = k.take<float>(13 * 64); L.inv_s = k.take<float>(4);
  L.adam_m = k.take<float>(N_SMALL); L.adam_v = k.take<float>(N_SMALL);
  L.total = (k.off + 255) / 256 * 256;
  return DSU_OK;
}

bool cfg_ok(const dsu_nsr_driver_cfg& c) {
  return c.cap_points > 0 && c.cap_rays > 0 && c.cap_rays <= DSU_RAY_LOSS_MAX_RAYS && c.n_random > 0 &&
         c.radius > 0.0f && c.render_step_size > 0.0f && c.image_channels >= 1 && c.image_channels <= 4 &&
         (c.sort_bits == 0 || (c.sort_bits >= 4 && c.sort_bits <= 7)) && c.grid.n_levels <= DSU_MAX_LEVELS;
}

// Side streams handed from one driver to the next (dsu_set_nsr_side_stream_pooling, OFF by default: six of six
// back-to-back reconstructions fast at high priority with it, the bench line with drawings in flight not measured yet).  A driver lives for one drawing; with the pool a process creates as many side streams as it has
// drawings in flight instead of one per drawing — the slow mode of profiles/round6_side_stream_priority.txt starts
// with the fourth stream of non-default priority a process creates.  Keyed by device and priority level; a stream
// enters the pool drained (dsu_nsr_driver_destroy synchronises it first).
struct PooledSide { int device, level; hipStream_t stream; };
std::mutex side_pool_mu;
std::vector<PooledSide> side_pool;

hipStream_t side_pool_take(int device, int level) {
  std::lock_guard<std::mutex> g(side_pool_mu);
  for (size_t i = 0; i < side_pool.size(); ++i)
    if (side_pool[i].device == device && side_pool[i].level == level) {
      hipStream_t s = side_pool[i].stream;
      side_pool.erase(side_pool.begin() + (long)i);
      return s;
    }
  return nullptr;
}

void side_pool_give(int device, int level, hipStream_t s) {
  std::lock_guard<std::mutex> g(side_pool_mu);
  side_pool.push_back({device, level, s});
}

}  // namespace

struct dsu_nsr_driver {
  dsu_nsr_driver_cfg cfg;
  Layout L;
  // Sample sets are produced one step ahead of their use and consumed one step behind the host:
  // while the device runs step t-1 the host (in call t) already holds step t's set and issues
  // step t+1's — three sets in flight, indexed by step % 3.
  hipStream_t side = nullptr;
  hipEvent_t ready[3] = {nullptr, nullptr, nullptr};   // side: set packed, stats copied to the host
  hipEvent_t gate = nullptr;                           // main: MLP part of the latest backward done
  hipEvent_t fwd_done = nullptr;                       // main: this step's geometry forward done
  bool fold = true;                 // DSU_NSR_FOLD (variant builds)
  bool tex_masks = true;            // DSU_NSR_TEX_MASKS (variant builds)
  bool side_high_priority = true;   // DSU_NSR_SIDE_PRIO
  int side_level = 1, side_device = 0;   // (what the side stream was made with: the pool's key)
  int pack_gate = 0;                // DSU_NSR_PACK_GATE: 0 with the march, 1 behind this step's geometry
                                    // forward (own event), 2 behind the MLP part of this step's backward
  int32_t* host_stats = nullptr;                       // pinned, 3 x int32[2]
  bool pf_valid[3] = {false, false, false};
  int64_t pf_step[3] = {-1, -1, -1};
  int32_t pf_rays[3] = {0, 0, 0};
  // what else the marched set depends on (a set issued for other settings is not reused)
  int32_t pf_randomized[3] = {0, 0, 0}, pf_occ_res[3] = {0, 0, 0};
  const void* pf_occ[3] = {nullptr, nullptr, nullptr};
  int64_t last_step = INT64_MIN;              // the sample-loss accumulators are pre-zeroed for last_step + 1
  bool initialised = false;                   // effective weights / inv_s / zeroed accumulators
  float aabb[6];
  int32_t rowcap;
  // optional HIP-event timing of the two geometry families (bench.py's roofline object)
  bool timing = false;                        // this step's launches are timed
  int32_t timing_stride = 0;                  // 0: off; n: every n-th step (an event pair costs ~7 us
                                              // of main-queue time: 28 us per step when all are timed)
  std::vector<hipEvent_t> ev[2];              // family 0: geometry forward, 1: geometry backward
  double work[2] = {0.0, 0.0};                // algorithmic bytes of the timed launches
  double flops[2] = {0.0, 0.0};               // their algorithmic MLP flops (what actually bounds them)
};

#define DSU_TRY(expr)            \
  do {                           \
    const int rc__ = (expr);     \
    if (rc__ != DSU_OK) return rc__; \
  } while (0)
#define DSU_HIP(expr)            \
  do {                           \
    if ((expr) != hipSuccess) return DSU_ELAUNCH; \
  } while (0)

namespace {

// draws + ray batch + ray/box + single-pass march of the sample set of `step` into prefetch set
// `p`, on stream `s` ...
int enqueue_march(dsu_nsr_driver* d, int p, int64_t step, int32_t n_rays,
                  const dsu_nsr_step_args& a, bool injected, hipStream_t s) {
  const dsu_nsr_driver_cfg& c = d->cfg;
  Prefetch& f = d->L.pf[p];
  const int64_t* index = f.index; const int64_t* px = f.px; const int64_t* py = f.py;
  const float* jitter = f.jitter;
  const bool have_rays = a.inj_rays || (a.inj_index && a.inj_x && a.inj_y);
  const bool need_draw = !injected || !have_rays || !a.inj_pts_random || !a.inj_perturb ||
                         (a.randomized && !a.inj_jitter);
  if (need_draw)
    DSU_TRY(dsu_nsr_draws(c.seed, step, n_rays, c.V, c.H, c.W, f.index, f.px, f.py, f.jitter,
                          c.n_random, f.pts_random, f.perturb, s));
  if (injected) {
    if (a.inj_index) index = a.inj_index;
    if (a.inj_x) px = a.inj_x;
    if (a.inj_y) py = a.inj_y;
    if (a.inj_jitter) jitter = a.inj_jitter;
  }
  if (injected && a.inj_rays) {
    // tests: a whole ray batch (what preprocess_data returns) instead of dataset draws — the rows
    // are copied into the prefetch set, the rays split into origins / directions
    if (!a.inj_rgb || !a.inj_normal || !a.inj_mask || !a.inj_cosines || !a.inj_view_weights)
      return DSU_EINVAL;
    const size_t n = (size_t)n_rays;
    DSU_HIP(hipMemcpyAsync(f.rays, a.inj_rays, n * 6 * sizeof(float), hipMemcpyDeviceToDevice, s));
    DSU_HIP(hipMemcpyAsync(f.rgb, a.inj_rgb, n * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    DSU_HIP(hipMemcpyAsync(f.normal, a.inj_normal, n * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    DSU_HIP(hipMemcpyAsync(f.mask, a.inj_mask, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    DSU_HIP(hipMemcpyAsync(f.cosines, a.inj_cosines, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    DSU_HIP(hipMemcpyAsync(f.vw, a.inj_view_weights, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    DSU_HIP(hipMemcpy2DAsync(f.rays_o, 3 * sizeof(float), a.inj_rays, 6 * sizeof(float),
                             3 * sizeof(float), n, hipMemcpyDeviceToDevice, s));
    DSU_HIP(hipMemcpy2DAsync(f.rays_d, 3 * sizeof(float), a.inj_rays + 3, 6 * sizeof(float),
                             3 * sizeof(float), n, hipMemcpyDeviceToDevice, s));
  } else {
    // rays (n,6) and, from the same launch, contiguous origins / directions for the marcher and the
    // compositing kernels (they used to be two strided copies per step)
    DSU_TRY(dsu_ortho_ray_batch_split(index, px, py, n_rays, c.c2w, c.origins, c.directions, c.images,
                                      c.image_channels, c.normals, c.masks, c.view_weights, c.H, c.W,
                                      f.rays, f.rgb, f.normal, f.mask, f.cosines, f.vw, f.rays_o,
                                      f.rays_d, s));
  }
  DSU_TRY(dsu_ray_aabb(f.rays_o, f.rays_d, n_rays, d->aabb, a.randomized ? jitter : nullptr,
                       c.render_step_size, f.tmin, f.tmax, s));
  DSU_TRY(dsu_ray_march_scratch(f.rays_o, f.rays_d, f.tmin, f.tmax, n_rays, d->aabb, a.occ_binary,
                                a.occ_binary ? a.occ_res : 0, c.render_step_size, d->rowcap,
                                f.counts, d->L.scratch_ts, d->L.scratch_te, s));
  return DSU_OK;
}

// ... and offsets scan + packing + random tail + Morton sort; the two stats words (total, max
// count) go to pinned memory.
int enqueue_pack(dsu_nsr_driver* d, int p, int32_t n_rays, const dsu_nsr_step_args& a,
                 bool injected, hipStream_t s) {
  const dsu_nsr_driver_cfg& c = d->cfg;
  Prefetch& f = d->L.pf[p];
  const float* pts_random = (injected && a.inj_pts_random) ? a.inj_pts_random : f.pts_random;
  const float* perturb = (injected && a.inj_perturb) ? a.inj_perturb : f.perturb;
  DSU_TRY(dsu_ray_offsets(f.counts, n_rays, f.offsets, f.stats, s));
  DSU_TRY(dsu_ray_compact_points_cap(d->L.scratch_ts, d->L.scratch_te, d->rowcap, f.offsets,
                                     f.counts, n_rays, f.rays_o, f.rays_d, f.t_starts, f.t_ends,
                                     f.points, c.cap_points, s));
  DSU_TRY(dsu_points_tail(f.points, (int64_t)c.cap_points + 2 * c.n_random, f.stats, pts_random,
                          perturb, c.n_random, 1e-2f, s));
  if (c.sort_bits)
    DSU_TRY(dsu_spatial_sort_dev(f.points, (int64_t)c.cap_points + 2 * c.n_random, f.stats,
                                 2 * c.n_random, c.radius, c.sort_bits, f.perm, f.sorted,
                                 f.sort_ws, d->L.sort_ws_bytes, s));
  DSU_HIP(hipMemcpyAsync(d->host_stats + 2 * p, f.stats, 2 * sizeof(int32_t),
                         hipMemcpyDeviceToHost, s));
  return DSU_OK;
}

int enqueue_samples(dsu_nsr_driver* d, int p, int64_t step, int32_t n_rays,
                    const dsu_nsr_step_args& a, bool injected, hipStream_t s) {
  DSU_TRY(enqueue_march(d, p, step, n_rays, a, injected, s));
  return enqueue_pack(d, p, n_rays, a, injected, s);
}

// start/stop events around a timed family (no-ops unless timing is on)
int mark(dsu_nsr_driver* d, int family, hipStream_t s) {
  if (!d->timing) return DSU_OK;
  hipEvent_t e;
  DSU_HIP(hipEventCreate(&e));
  DSU_HIP(hipEventRecord(e, s));
  d->ev[family].push_back(e);
  return DSU_OK;
}

int launch_small_update(dsu_nsr_driver* d, const dsu_nsr_step_args* a, int update, hipStream_t s) {
  const dsu_nsr_driver_cfg& c = d->cfg;
  SmallArgs sa;
  sa.w0_v = c.w0_v; sa.w0_g = c.w0_g; sa.b0 = c.b0; sa.w1_v = c.w1_v; sa.w1_g = c.w1_g; sa.b1 = c.b1;
  for (int k = 0; k < 6; ++k) sa.tex[k] = c.tex[k];
  sa.variance = c.variance;
  sa.g_geo = d->L.g_geo; sa.g_tex = d->L.g_tex; sa.d_inv = d->L.d_inv;
  sa.m = d->L.adam_m; sa.v = d->L.adam_v;
  sa.w0_eff = d->L.w0_eff; sa.w1_eff = d->L.w1_eff; sa.inv_s = d->L.inv_s;
  sa.zero_terms = d->L.terms + 8 * (int)((a->step + 1) & 1) + 4;
  sa.terms = d->L.terms + 8 * (int)(a->step & 1);
  sa.terms_out = update ? a->terms_out : nullptr;
  sa.beta1 = c.beta1; sa.beta2 = c.beta2; sa.eps = c.adam_eps; sa.wd = c.weight_decay;
  sa.update = update;
  sa.lr_geo = sa.lr_tex = sa.lr_var = 0.0f;
  sa.bc1 = sa.bc2_sqrt = 1.0f;
  if (update) {
    sa.lr_geo = a->lr_geometry; sa.lr_tex = a->lr_texture; sa.lr_var = a->lr_variance;
    sa.bc1 = (float)(1.0 - pow((double)c.beta1, (double)a->adam_step));
    sa.bc2_sqrt = (float)sqrt(1.0 - pow((double)c.beta2, (double)a->adam_step));
  }
  // ... and the hash table (active levels; level / decay bookkeeping is the caller's)
  dsu_table_adamw_args ta{};
  int table_blocks = 0;
  if (update && a->table_p && !d->fold) {
    DSU_TRY(dsu_table_adamw(a->table_p, a->table_grad, a->table_m, a->table_v,
                            const_cast<void*>(a->table_img), a->table_n, a->table_lr, c.beta1, c.beta2,
                            a->table_eps, a->table_wd, a->table_bc1, a->table_bc2_sqrt, s));
  } else if (update && a->table_p) {
    if (a->table_n < 0 || (a->table_n & 3) || !a->table_grad || !a->table_m || !a->table_v ||
        !a->table_img || !(a->table_bc1 > 0.0f) || !(a->table_bc2_sqrt > 0.0f))
      return DSU_EINVAL;
    ta = dsu_table_adamw_args{(float4*)a->table_p, (float4*)a->table_grad, (float4*)a->table_m,
                              (float4*)a->table_v, (__half2*)const_cast<void*>(a->table_img),
                              a->table_n / 4, a->table_lr, c.beta1, c.beta2, a->table_eps,
                              a->table_wd, a->table_bc1, a->table_bc2_sqrt};
    table_blocks = (int)((ta.n4 + 1023) / 1024 < 512 ? (ta.n4 + 1023) / 1024 : 512);
  }
  small_update_kernel<<<dim3(1 + table_blocks), dim3(1024), 0, s>>>(sa, ta);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // namespace

extern "C" {

int dsu_nsr_draws(uint64_t seed, int64_t step, int32_t n_rays, int32_t V, int32_t H, int32_t W,
                  int64_t* index, int64_t* x, int64_t* y, float* jitter, int32_t n_random,
                  float* pts_random, float* perturb, void* stream) {
  if (n_rays < 0 || n_random < 0 || V < 1 || H < 1 || W < 1 ||
      (n_rays && (!index || !x || !y || !jitter)) || (n_random && (!pts_random || !perturb)))
    return DSU_EINVAL;
  const int n = n_rays > n_random ? n_rays : n_random;
  if (n == 0) return DSU_OK;
  nsr_draw_kernel<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(
      seed, step, n_rays, V, H, W, index, x, y, jitter, n_random, pts_random, perturb);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int64_t dsu_nsr_driver_workspace_bytes(const dsu_nsr_driver_cfg* cfg) {
  if (!cfg || !cfg_ok(*cfg)) return DSU_EINVAL;
  Layout L;
  if (carve(*cfg, nullptr, L) != DSU_OK) return DSU_EINVAL;
  return L.total;
}

int dsu_nsr_driver_create(const dsu_nsr_driver_cfg* cfg, dsu_nsr_driver** out) {
  if (!cfg || !out || !cfg_ok(*cfg) || !cfg->workspace) return DSU_EINVAL;
  const void* need[] = {cfg->c2w, cfg->origins, cfg->directions, cfg->images, cfg->normals,
                        cfg->masks, cfg->view_weights, cfg->w0_v, cfg->w0_g, cfg->b0, cfg->w1_v,
                        cfg->w1_g, cfg->b1, cfg->variance, cfg->tex[0], cfg->tex[1], cfg->tex[2],
                        cfg->tex[3], cfg->tex[4], cfg->tex[5]};
  for (const void* p : need)
    if (!p) return DSU_EINVAL;
  dsu_nsr_driver* d = new dsu_nsr_driver();
  d->cfg = *cfg;
  if (carve(*cfg, (char*)cfg->workspace, d->L) != DSU_OK || d->L.total > cfg->workspace_bytes) {
    delete d;
    return DSU_EINVAL;
  }
  for (int k = 0; k < 3; ++k) {
    d->aabb[k] = -cfg->radius;
    d->aabb[3 + k] = cfg->radius;
  }
  d->rowcap = march_row_capacity(cfg->radius, cfg->render_step_size);
  {
    std::vector<int32_t> map((size_t)dsu_texture_partial_map(nullptr));
    dsu_texture_partial_map(map.data());
    if (hipMemcpy(d->L.tex_map, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice) !=
        hipSuccess) {
      delete d;
      return DSU_ELAUNCH;
    }
  }
  int lo = 0, hi = 0;
  const int side_prio = dsu_ab_int("DSU_NSR_SIDE_PRIO", dsu_nsr_side_priority_value);   // 1 high, 2 normal, 0 low
  d->side_high_priority = side_prio == 1;
  // 0 since round 4: with 16-ray marching waves and the shorter step the packing right behind the
  // march measured 1.148 against 1.161 ms per step for "behind the geometry forward" (five
  // interleaved pairs, same box) — and the main queue loses one event record per step
  d->pack_gate = dsu_ab_int("DSU_NSR_PACK_GATE", 0);
  d->fold = dsu_ab_int("DSU_NSR_FOLD", 1) != 0;
  d->tex_masks = dsu_ab_int("DSU_NSR_TEX_MASKS", 1) != 0;   // (variant builds: 0 = exact f32 recompute, for A/B)
  d->side_level = side_prio;
  bool ok = hipGetDevice(&d->side_device) == hipSuccess;
  if (ok && dsu_nsr_side_pool_value) d->side = side_pool_take(d->side_device, side_prio);
  ok = ok && (d->side != nullptr ||
              (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
               hipStreamCreateWithPriority(&d->side, hipStreamNonBlocking,
                                           side_prio == 1 ? hi : (side_prio == 2 ? 0 : lo)) == hipSuccess));
  for (int p = 0; p < 3; ++p)
    ok = ok && hipEventCreateWithFlags(&d->ready[p], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&d->gate, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&d->fwd_done, hipEventDisableTiming) == hipSuccess &&
       hipHostMalloc((void**)&d->host_stats, 6 * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
  if (!ok) {
    dsu_nsr_driver_destroy(d);
    return DSU_ELAUNCH;
  }
  *out = d;
  return DSU_OK;
}

void dsu_nsr_driver_destroy(dsu_nsr_driver* d) {
  if (!d) return;
  if (d->side) {
    (void)hipStreamSynchronize(d->side);
    if (dsu_nsr_side_pool_value) side_pool_give(d->side_device, d->side_level, d->side);
    else (void)hipStreamDestroy(d->side);
  }
  for (int p = 0; p < 3; ++p)
    if (d->ready[p]) (void)hipEventDestroy(d->ready[p]);
  if (d->gate) (void)hipEventDestroy(d->gate);
  if (d->fwd_done) (void)hipEventDestroy(d->fwd_done);
  if (d->host_stats) (void)hipHostFree(d->host_stats);
  for (int f = 0; f < 2; ++f)
    for (hipEvent_t e : d->ev[f]) (void)hipEventDestroy(e);
  delete d;
}

const float* dsu_nsr_driver_terms(const dsu_nsr_driver* d) { return d ? d->L.terms : nullptr; }

const float* dsu_nsr_driver_adam_moments(const dsu_nsr_driver* d, int32_t second) {
  return d ? (second ? d->L.adam_v : d->L.adam_m) : nullptr;
}

int dsu_nsr_driver_step(dsu_nsr_driver* d, dsu_nsr_step_args* a, void* main_stream) {
  if (!d || !a || a->n_rays <= 0 || a->n_rays > d->cfg.cap_rays || !a->table_img || !a->table_grad ||
      a->active_levels == 0 || a->active_levels > d->cfg.grid.n_levels || a->adam_step < 1)
    return DSU_EINVAL;
  const dsu_nsr_driver_cfg& c = d->cfg;
  Layout& L = d->L;
  hipStream_t s = (hipStream_t)main_stream;
  d->timing = d->timing_stride > 0 && a->step % d->timing_stride == 0;
  const int p = (int)(a->step % 3);
  Prefetch& f = L.pf[p];
  float* terms = L.terms + 8 * (int)(a->step & 1);   // two sets: the optimizer kernel pre-zeroes the next one
  // any error return below leaves no "valid" prefetched set behind (its `ready` event may never
  // have been recorded: the next call would read stale counts)
  struct Guard {
    dsu_nsr_driver* d;
    bool ok = false;
    ~Guard() {
      if (!ok) for (int k = 0; k < 3; ++k) d->pf_valid[k] = false;
    }
  } guard{d};
  if (d->initialised && a->step != d->last_step + 1)
    // the previous step pre-zeroed the accumulators of `last_step + 1`, not of this step (the
    // caller moved global_step: resume, tests): zero this step's set here
    DSU_HIP(hipMemsetAsync(terms + 4, 0, 3 * sizeof(float), s));
  if (!d->initialised) {
    // zeroed accumulators and optimizer moments
    DSU_HIP(hipMemsetAsync(L.g_geo, 0, (N_GEO + N_TEX + 1) * sizeof(float), s));
    DSU_HIP(hipMemsetAsync(L.adam_m, 0, N_SMALL * sizeof(float), s));
    DSU_HIP(hipMemsetAsync(L.adam_v, 0, N_SMALL * sizeof(float), s));
    DSU_HIP(hipMemsetAsync(L.terms, 0, 16 * sizeof(float), s));
  }
  if (!d->initialised || a->refresh_effective) {
    DSU_TRY(launch_small_update(d, a, 0, s));       // effective weights and inv_s
    d->initialised = true;
  }
  // ---- this step's samples: prefetched by the previous call, or produced now
  const bool injected = a->inj_index || a->inj_x || a->inj_y || a->inj_jitter || a->inj_pts_random ||
                        a->inj_perturb || a->inj_rays;
  if (d->pf_valid[p] && d->pf_step[p] == a->step && d->pf_rays[p] == a->n_rays && !injected &&
      d->pf_randomized[p] == a->randomized && d->pf_occ[p] == a->occ_binary &&
      d->pf_occ_res[p] == a->occ_res) {
    // the host waits: no device-side cross-queue barrier on the main stream is needed afterwards
    DSU_HIP(hipEventSynchronize(d->ready[p]));
  } else {
    // nothing usable in flight (first step, a step behind an occupancy refresh, a ray count
    // changed from outside, injected draws): let the side stream drain, then produce the set on
    // the main stream and wait for its two stats words
    DSU_HIP(hipStreamSynchronize(d->side));
    for (int k = 0; k < 3; ++k) d->pf_valid[k] = false;
    DSU_TRY(enqueue_samples(d, p, a->step, a->n_rays, *a, injected, s));
    DSU_HIP(hipStreamSynchronize(s));
  }
  d->pf_valid[p] = false;
  const int32_t total = d->host_stats[2 * p], cmax = d->host_stats[2 * p + 1];
  a->out_n_samples = total;
  a->out_max_count = cmax;
  if (total > c.cap_points || cmax > d->rowcap) return DSU_EUNSUP;
  const int64_t n_s = total, n_r = c.n_random, n_all = n_s + 2 * n_r;
  // dynamic ray count of the next step (neus_ortho.py:88-92)
  int32_t next_rays = a->n_rays;
  if (c.dynamic_ray_sampling && n_s > 0) {
    const int tr = (int)((double)a->n_rays * ((double)c.train_num_samples / (double)n_s));
    const int nr2 = (int)((double)a->n_rays * 0.9 + (double)tr * 0.1);
    next_rays = nr2 < c.cap_rays ? nr2 : c.cap_rays;
    if (next_rays < 1) next_rays = 1;
  }
  a->out_next_n_rays = next_rays;
  if (a->prefetch_next) {
    // The next step's samples depend on the occupancy grid and the draws, not on the parameters.
    // Their serial march (128 waves of ~100 registers for ~0.35 ms) cannot share a SIMD with the
    // one-wave-per-SIMD kernels of a step (texture backward: 509 registers, MLP part of the
    // geometry backward: 458) — started behind the geometry forward it delayed the texture
    // backward's last workgroups by the march's remaining time (245 instead of 170 us, kernel
    // trace).  It is therefore queued one step earlier, behind the MLP part of the backward the
    // device is executing NOW (`gate`, recorded by the previous call): it then runs beside the
    // table-gradient scatter, the optimizer kernels and the next geometry forward, which all
    // leave registers free.  Set q was last read by step t-2, whose kernels precede `gate` in
    // the main stream's order: no event of its own (an event record costs the main queue a
    // 6-11 us bubble: barrier packet + the next dispatch waiting for it).
    const int q = (int)((a->step + 1) % 3);
    DSU_HIP(hipStreamWaitEvent(d->side, d->gate, 0));
    dsu_nsr_step_args na = *a;
    na.inj_index = na.inj_x = na.inj_y = nullptr;
    na.inj_jitter = na.inj_pts_random = na.inj_perturb = nullptr;
    na.inj_rays = nullptr;
    DSU_TRY(enqueue_march(d, q, a->step + 1, next_rays, na, false, d->side));
    d->pf_step[q] = a->step + 1;
    d->pf_rays[q] = next_rays;
    d->pf_randomized[q] = a->randomized;
    d->pf_occ[q] = a->occ_binary;
    d->pf_occ_res[q] = a->occ_res;
    if (d->pack_gate == 0) {
      DSU_TRY(enqueue_pack(d, q, next_rays, na, false, d->side));
      DSU_HIP(hipEventRecord(d->ready[q], d->side));
      d->pf_valid[q] = true;                  // only once `ready[q]` is on the side stream
    }
  }

  const float* pts = c.sort_bits ? f.sorted : f.points;
  const int32_t* perm = c.sort_bits ? f.perm : nullptr;
  dsu_sdf_mlp mlp{L.w0_eff, c.b0, L.w1_eff, c.b1};
  // ---- forward
  const double alg_bytes = (double)n_all * (7.0 * a->active_levels * 8 * 4 + 12 + 72);
  // geometry MLP per point: 7 evaluations of 64 x (3 + 2 L) MACs, 13 outputs for the centre and 1
  // for each offset; the backward pass does the layer-0 product three times (recompute, dIn, gW0)
  // and the layer-1 product three times (recompute, dH, gW1)
  const double l0 = 7.0 * 64 * (3 + 2 * a->active_levels), l1 = 64.0 * (13 + 6);
  const double fwd_flops = (double)n_all * 2.0 * (l0 + l1), bwd_flops = (double)n_all * 2.0 * 3.0 * (l0 + l1);
  DSU_TRY(mark(d, 0, s));
  DSU_TRY(dsu_sdf_fd_fwd_sorted(&c.grid, a->table_img, &mlp, pts, perm, n_all, c.radius, a->eps,
                                a->active_levels, L.a_sdf, L.a_grad, L.a_feat, nullptr, L.enc_cache, s));
  DSU_TRY(mark(d, 0, s));
  if (d->timing) { d->work[0] += alg_bytes; d->flops[0] += fwd_flops; }
  if (a->prefetch_next && d->pack_gate == 1) {
    // The packing / sorting launches of the next set (a dozen short, chip-wide kernels) slowed the
    // gather-bound geometry forward by a quarter when they ran beside it (kernel trace: 221 vs
    // 175 us); behind it they share the chip with the small shading / loss kernels instead.
    const int q = (int)((a->step + 1) % 3);
    dsu_nsr_step_args na = *a;
    na.inj_index = na.inj_x = na.inj_y = nullptr;
    na.inj_jitter = na.inj_pts_random = na.inj_perturb = nullptr;
    na.inj_rays = nullptr;
    DSU_HIP(hipEventRecord(d->fwd_done, s));
    DSU_HIP(hipStreamWaitEvent(d->side, d->fwd_done, 0));
    DSU_TRY(enqueue_pack(d, q, next_rays, na, false, d->side));
    DSU_HIP(hipEventRecord(d->ready[q], d->side));
    d->pf_valid[q] = true;
  }
  if (n_s > 0) {
    dsu_tex_mlp tex{c.tex[0], c.tex[1], c.tex[2], c.tex[3], c.tex[4], c.tex[5]};
    // (with the ReLU pattern of hidden layer 1 for the backward: its recompute runs as bf16 x 3)
    DSU_TRY(dsu_texture_fwd_shaded_m(&tex, L.a_feat, L.a_grad, n_s, L.normal, L.rgb,
                                     d->tex_masks ? L.tex_mask : nullptr, s));
    DSU_TRY(dsu_neus_composite_fwd(L.a_sdf, L.normal, L.rgb, f.rays_d, f.t_starts, f.t_ends,
                                   f.offsets, f.counts, a->n_rays, L.inv_s, a->cos_anneal_ratio,
                                   L.alpha, L.w, L.comp, s));
  } else {
    DSU_HIP(hipMemsetAsync(L.comp, 0, (size_t)a->n_rays * 8 * sizeof(float), s));
  }
  DSU_TRY(dsu_ray_losses(L.comp, f.rgb, f.normal, f.mask, f.cosines, f.vw, a->n_rays, &c.ray_loss,
                         terms, L.d_comp, s));
  // ---- backward
  dsu_partial_reduce tex_red{};
  bool have_tex_red = false;
  if (n_s > 0) {
    DSU_TRY(dsu_neus_composite_bwd(L.a_sdf, L.normal, L.rgb, f.rays_d, f.t_starts, f.t_ends,
                                   f.offsets, f.counts, a->n_rays, L.inv_s, a->cos_anneal_ratio,
                                   L.alpha, L.w, L.d_comp, nullptr, L.d_sdf_all, L.d_normal, L.d_rgb,
                                   L.d_inv, s));
    dsu_tex_mlp tex{c.tex[0], c.tex[1], c.tex[2], c.tex[3], c.tex[4], c.tex[5]};
    // parameter gradients of the texture MLP: per-workgroup partials now, summed into g_tex by
    // extra workgroups of the geometry backward's scatter launch (no launch of their own;
    // DSU_NSR_FOLD=0 in variant builds: the separate reduction and table-AdamW launches, for A/B)
    if (!d->fold) {
      float* gt = L.g_tex;
      DSU_TRY(dsu_texture_bwd_shaded(&tex, L.a_feat, L.a_grad, L.rgb, L.d_rgb, L.d_normal, n_s, 2 * n_r,
                                     L.d_grad_all, L.d_feat_all, gt, gt + 1024, gt + 1088, gt + 5184,
                                     gt + 5248, gt + 5440, L.tex_ws, L.tex_ws_bytes, s));
    } else {
    DSU_TRY(dsu_texture_bwd_shaded_partials_m(&tex, L.a_feat, L.a_grad, L.rgb, L.d_rgb, L.d_normal, n_s,
                                              2 * n_r, L.d_grad_all, L.d_feat_all,
                                              d->tex_masks ? L.tex_mask : nullptr, L.tex_ws,
                                              L.tex_ws_bytes, &tex_red, s));
    tex_red.map = L.tex_map;
    tex_red.base = L.g_tex;
    have_tex_red = true;
    }
  } else {
    DSU_HIP(hipMemsetAsync(L.d_feat_all, 0, (size_t)(2 * n_r) * 13 * sizeof(float), s));
  }
  DSU_TRY(dsu_sample_losses(L.a_sdf, L.a_grad, n_s, n_r, c.lambda_eikonal, c.lambda_sparsity,
                            c.sparsity_scale, c.lambda_smooth, 1 | 2, L.d_sdf_all, L.d_grad_all,
                            terms + 4, s));
  float* gg = L.g_geo;
  DSU_TRY(mark(d, 1, s));
  DSU_TRY(dsu_sdf_fd_bwd_sorted_fold(&c.grid, a->table_img, &mlp, pts, perm, n_all, c.radius, a->eps,
                                     a->active_levels, L.d_sdf_all, L.d_grad_all, L.d_feat_all, nullptr,
                                     a->table_grad, gg, gg + 64 * 23, gg + 64 * 23 + 64,
                                     gg + 64 * 23 + 64 + 13 * 64, L.sdf_ws, L.sdf_ws_bytes,
                                     L.enc_cache, d->gate, have_tex_red ? &tex_red : nullptr, s));
  DSU_TRY(mark(d, 1, s));
  if (d->timing) { d->work[1] += alg_bytes; d->flops[1] += bwd_flops; }
  if (a->prefetch_next && d->pack_gate == 2) {
    const int q = (int)((a->step + 1) % 3);
    dsu_nsr_step_args na = *a;
    na.inj_index = na.inj_x = na.inj_y = nullptr;
    na.inj_jitter = na.inj_pts_random = na.inj_perturb = nullptr;
    na.inj_rays = nullptr;
    DSU_HIP(hipStreamWaitEvent(d->side, d->gate, 0));          // recorded inside the call above
    DSU_TRY(enqueue_pack(d, q, next_rays, na, false, d->side));
    DSU_HIP(hipEventRecord(d->ready[q], d->side));
    d->pf_valid[q] = true;
  }
  // ---- optimizer: the hash table and the small tensors in one launch
  DSU_TRY(launch_small_update(d, a, 1, s));
  d->last_step = a->step;
  guard.ok = true;
  return DSU_OK;
}

int dsu_nsr_driver_occ_refresh(dsu_nsr_driver* d, const dsu_occgrid_refresh_args* args,
                               void* main_stream) {
  if (!d || !args) return DSU_EINVAL;
  if (!d->initialised) return DSU_EUNSUP;             // effective weights exist after the first step
  const dsu_nsr_driver_cfg& c = d->cfg;
  // nothing of the side stream may still read the grid that is about to be rewritten
  DSU_HIP(hipStreamSynchronize(d->side));
  for (int k = 0; k < 3; ++k) d->pf_valid[k] = false;
  dsu_occgrid_refresh_args a = *args;
  const dsu_sdf_mlp mlp{d->L.w0_eff, c.b0, d->L.w1_eff, c.b1};
  a.grid = &c.grid;
  a.mlp = &mlp;
  a.inv_s = d->L.inv_s;
  a.aabb = d->aabb;
  a.seed = c.seed;
  a.radius = c.radius;
  a.render_step_size = c.render_step_size;
  return dsu_occgrid_refresh(&a, main_stream);
}

int dsu_nsr_driver_timing(dsu_nsr_driver* d, int32_t enable) {
  if (!d) return DSU_EINVAL;
  for (int f = 0; f < 2; ++f) {
    for (hipEvent_t e : d->ev[f]) (void)hipEventDestroy(e);
    d->ev[f].clear();
    d->work[f] = 0.0;
    d->flops[f] = 0.0;
  }
  d->timing_stride = enable > 0 ? enable : 0;
  d->timing = false;
  return DSU_OK;
}

int dsu_nsr_driver_timing_read(dsu_nsr_driver* d, int32_t family, int64_t* launches,
                               double* total_ms, double* alg_bytes) {
  if (!d || family < 0 || family > 1 || !launches || !total_ms || !alg_bytes) return DSU_EINVAL;
  const std::vector<hipEvent_t>& ev = d->ev[family];
  double ms = 0.0;
  const size_t pairs = ev.size() / 2;
  for (size_t i = 0; i < pairs; ++i) {
    DSU_HIP(hipEventSynchronize(ev[2 * i + 1]));
    float t = 0.0f;
    DSU_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
    ms += t;
  }
  *launches = (int64_t)pairs;
  *total_ms = ms;
  *alg_bytes = d->work[family];
  return DSU_OK;
}

int dsu_nsr_driver_timing_flops(dsu_nsr_driver* d, int32_t family, double* mlp_flops) {
  if (!d || family < 0 || family > 1 || !mlp_flops) return DSU_EINVAL;
  *mlp_flops = d->flops[family];
  return DSU_OK;
}

int dsu_nsr_driver_sync(dsu_nsr_driver* d) {
  if (!d) return DSU_EINVAL;
  DSU_HIP(hipStreamSynchronize(d->side));
  for (int k = 0; k < 3; ++k) d->pf_valid[k] = false;
  return DSU_OK;
}

}  // extern "C"
