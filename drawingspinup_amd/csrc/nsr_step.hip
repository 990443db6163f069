// Per-step glue of the Instant-NSR optimisation loop as single launches (gfx950).
//
// The reference evaluates these pieces as dozens of small framework kernels per step
// (2_charactor_reconstructor/instant_nsr/systems/neus_ortho.py:94-151 for the loss terms,
// systems/criterions.py:4-27 for binary_cross_entropy / ranking_loss); each one is tiny, so the
// step is bound by launch count, not by bytes.  Here:
//   ray_losses_kernel     the three ray-level ranking losses AND their gradient w.r.t. the
//                         composite (R,8) in one launch (one workgroup per term: error ->
//                         LDS bitonic sort -> the reference's selection -> chain rule)
//   sample_losses_kernel  eikonal / sparsity / 3-D normal smoothness and their gradients
//   ray_offsets_kernel    exclusive scan of the per-ray sample counts + (total, max)
#include "common.h"
#include "adamw_dev.h"
#include <math.h>

namespace {

constexpr int RL_THREADS = 1024;

__device__ __forceinline__ uint32_t flipf(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unflipf(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sum over the workgroup; every thread gets the result.  red: >= 17 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < nw; ++w) t += red[w];
  return t;
}

// exclusive scan of one int per thread over the workgroup; *total = sum.  red: >= 17 ints.
__device__ __forceinline__ int block_excl_scan(int v, int* red, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) red[wave] = inc;
  __syncthreads();
  int before = 0, all = 0;
  for (int w = 0; w < nw; ++w) {
    const int t = red[w];
    if (w < wave) before += t;
    all += t;
  }
  *total = all;
  return before + inc - v;
}

__device__ void bitonic_sort(uint64_t* keys, int P) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const uint64_t a = keys[i], b = keys[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          keys[i] = b;
          keys[l] = a;
        }
      }
      __syncthreads();
    }
  }
}

struct RankLds {
  uint64_t* keys;   // [P]
  int* pos;         // [P]
  float* gerr;      // [P]
  float* redf;      // [32]
  int* redi;        // [32]
};

// ranking_loss(error[valid], ratio, weights[valid], type) of the reference, INCLUDING its
// selection rule (criterions.py:17-18 indexes the SORTED errors with the ORIGINAL positions of the
// k smallest): term = sum_{j<k} sorted[pos(idx_j)] * w[idx_j]  (/ k for 'mean'), where idx_j is
// the element with the j-th smallest error and pos() its position inside error[valid].
// Leaves in lds.gerr[i] the sum of weights with which error[i] enters the term.
// Ties are ordered by element index (the reference's device sort leaves their order open).
template <class ErrF>
__device__ float rank_select(ErrF errf, const float* __restrict__ vw, int R, int P, double ratio,
                             bool mean, const RankLds& lds, int64_t* k_out) {
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    float e = 0.0f;
    bool valid = false;
    if (i < R) errf(i, e, valid);
    lds.keys[i] = ((uint64_t)flipf(valid ? e : INFINITY) << 32) | (uint32_t)i;
    lds.pos[i] = valid ? 1 : 0;
    lds.gerr[i] = 0.0f;
  }
  __syncthreads();
  const int E = P / blockDim.x;          // contiguous chunk per thread
  int c = 0;
  for (int q = 0; q < E; ++q) c += lds.pos[threadIdx.x * E + q];
  int nvalid;
  int run = block_excl_scan(c, lds.redi, &nvalid);
  for (int q = 0; q < E; ++q) {
    const int i = threadIdx.x * E + q;
    run += lds.pos[i];
    lds.pos[i] = run - 1;
  }
  __syncthreads();
  bitonic_sort(lds.keys, P);
  const int64_t kk = (int64_t)floor(ratio * (double)nvalid);
  float total = 0.0f;
  for (int64_t j = threadIdx.x; j < kk; j += blockDim.x) {
    const uint32_t idx = (uint32_t)lds.keys[j];
    int pj = lds.pos[idx];
    pj = pj < 0 ? 0 : pj;
    const uint64_t kt = lds.keys[pj];
    const float val = unflipf((uint32_t)(kt >> 32));
    const float w = vw ? vw[idx] : 1.0f;
    total += val * w;
    atomicAdd(&lds.gerr[(uint32_t)kt], w);
  }
  total = block_sum(total, lds.redf);
  *k_out = kk;
  return mean ? total / (float)kk : total;
}

__device__ __forceinline__ float signf0(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// blockIdx.x: 0 = rgb (mse, then l1 when enabled), 1 = normal, 2 = mask
__global__ __launch_bounds__(RL_THREADS) void ray_losses_kernel(
    const float* __restrict__ comp, const float* __restrict__ rgb, const float* __restrict__ normal,
    const float* __restrict__ maskf, const float* __restrict__ cosines,
    const float* __restrict__ vw, int R, int P, dsu_ray_loss_cfg cfg, float* __restrict__ terms,
    float* __restrict__ d_comp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  RankLds lds;
  lds.keys = reinterpret_cast<uint64_t*>(smem);
  lds.pos = reinterpret_cast<int*>(smem + (size_t)P * 8);
  lds.gerr = reinterpret_cast<float*>(smem + (size_t)P * 12);
  lds.redf = reinterpret_cast<float*>(smem + (size_t)P * 16);
  lds.redi = reinterpret_cast<int*>(smem + (size_t)P * 16 + 128);
  const int term = blockIdx.x;

  // mask = (batch mask > 0) & (cosines' < -0.1), cosines' = cosines > -0.1 ? 0 : cosines
  auto fg = [&](int i) {
    const float cb = cosines[i] > -0.1f ? 0.0f : cosines[i];
    return maskf[i] > 0.0f && cb < -0.1f;
  };

  if (term == 0) {
    auto mse = [&](int i, float& e, bool& valid) {
      const float d0 = comp[i * 8 + 2] - rgb[i * 3], d1 = comp[i * 8 + 3] - rgb[i * 3 + 1],
                  d2 = comp[i * 8 + 4] - rgb[i * 3 + 2];
      e = (d0 * d0 + d1 * d1) + d2 * d2;
      valid = fg(i);
    };
    int64_t kk;
    float t = rank_select(mse, nullptr, R, P, cfg.rgb_p_ratio, true, lds, &kk) * cfg.lambda_rgb_mse;
    if (threadIdx.x == 0) terms[0] = t;
    __syncthreads();
    const float c = kk > 0 ? cfg.lambda_rgb_mse / (float)kk : 0.0f;
    for (int i = threadIdx.x; i < R; i += blockDim.x) {
      const float g = lds.gerr[i] * c;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        d_comp[i * 8 + 2 + ch] = 2.0f * (comp[i * 8 + 2 + ch] - rgb[i * 3 + ch]) * g;
    }
    float t1 = 0.0f;
    if (cfg.lambda_rgb_l1 != 0.0f) {
      __syncthreads();
      auto l1 = [&](int i, float& e, bool& valid) {
        const float d0 = comp[i * 8 + 2] - rgb[i * 3], d1 = comp[i * 8 + 3] - rgb[i * 3 + 1],
                    d2 = comp[i * 8 + 4] - rgb[i * 3 + 2];
        e = (fabsf(d0) + fabsf(d1)) + fabsf(d2);
        valid = fg(i);
      };
      t1 = rank_select(l1, nullptr, R, P, cfg.rgb_p_ratio, true, lds, &kk) * cfg.lambda_rgb_l1;
      __syncthreads();
      const float c1 = kk > 0 ? cfg.lambda_rgb_l1 / (float)kk : 0.0f;
      for (int i = threadIdx.x; i < R; i += blockDim.x) {
        const float g = lds.gerr[i] * c1;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
          d_comp[i * 8 + 2 + ch] += signf0(comp[i * 8 + 2 + ch] - rgb[i * 3 + ch]) * g;
      }
    }
    if (threadIdx.x == 0) terms[1] = t1;
  } else if (term == 1) {
    // normal_errors = 1 - cosine_similarity(F.normalize(comp_normal), normal); geo-aware:
    // * exp(|cosines'|) / sum_real exp(|cosines'|), ranking 'sum'; else ranking 'mean'
    float S = 1.0f;
    if (cfg.geo_aware) {
      float s = 0.0f;
      for (int i = threadIdx.x; i < R; i += blockDim.x) {
        const float cb = cosines[i] > -0.1f ? 0.0f : cosines[i];
        s += expf(fabsf(cb));
      }
      S = block_sum(s, lds.redf);
      __syncthreads();
    }
    auto cosim = [&](int i, float& cs, float u[3], float gh[3], float& nn, float& a, float n[3]) {
      const float v0 = comp[i * 8 + 5], v1 = comp[i * 8 + 6], v2 = comp[i * 8 + 7];
      const float nv = sqrtf((v0 * v0 + v1 * v1) + v2 * v2);
      a = fmaxf(nv, 1e-12f);
      n[0] = v0 / a; n[1] = v1 / a; n[2] = v2 / a;
      nn = fmaxf(sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]), 1e-8f);
      const float g0 = normal[i * 3], g1 = normal[i * 3 + 1], g2 = normal[i * 3 + 2];
      const float gn = fmaxf(sqrtf((g0 * g0 + g1 * g1) + g2 * g2), 1e-8f);
      u[0] = n[0] / nn; u[1] = n[1] / nn; u[2] = n[2] / nn;
      gh[0] = g0 / gn; gh[1] = g1 / gn; gh[2] = g2 / gn;
      cs = (u[0] * gh[0] + u[1] * gh[1]) + u[2] * gh[2];
    };
    auto nerr = [&](int i, float& e, bool& valid) {
      float cs, u[3], gh[3], nn, a, n[3];
      cosim(i, cs, u, gh, nn, a, n);
      e = 1.0f - cs;
      if (cfg.geo_aware) {
        const float cb = cosines[i] > -0.1f ? 0.0f : cosines[i];
        e = e * expf(fabsf(cb)) / S;
      }
      valid = fg(i);
    };
    int64_t kk;
    const bool mean = !cfg.geo_aware;
    float t = rank_select(nerr, vw, R, P, cfg.normal_p_ratio, mean, lds, &kk) * cfg.lambda_normal;
    if (threadIdx.x == 0) terms[2] = t;
    __syncthreads();
    const float c = mean ? (kk > 0 ? cfg.lambda_normal / (float)kk : 0.0f) : cfg.lambda_normal;
    for (int i = threadIdx.x; i < R; i += blockDim.x) {
      float cs, u[3], gh[3], nn, a, n[3];
      cosim(i, cs, u, gh, nn, a, n);
      float de = lds.gerr[i] * c;                       // d term / d err_i
      if (cfg.geo_aware) {
        const float cb = cosines[i] > -0.1f ? 0.0f : cosines[i];
        de = de / S * expf(fabsf(cb));
      }
      const float dcs = -de;
      const float nv = sqrtf((comp[i * 8 + 5] * comp[i * 8 + 5] + comp[i * 8 + 6] * comp[i * 8 + 6]) +
                             comp[i * 8 + 7] * comp[i * 8 + 7]);
      float dv[3] = {0.0f, 0.0f, 0.0f};
      if (nv > 1e-12f && dcs != 0.0f) {
        // d cos / d n = (ghat - cos u) / |n| ; d n / d v = (I - n n^T) / |v|
        float tn[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) tn[ch] = (gh[ch] - cs * u[ch]) / nn * dcs;
        const float nt = (n[0] * tn[0] + n[1] * tn[1]) + n[2] * tn[2];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) dv[ch] = (tn[ch] - n[ch] * nt) / a;
      }
      d_comp[i * 8 + 5] = dv[0];
      d_comp[i * 8 + 6] = dv[1];
      d_comp[i * 8 + 7] = dv[2];
    }
  } else {
    auto bce = [&](int i, float& e, bool& valid) {
      const float oc = fminf(fmaxf(comp[i * 8], 1e-3f), 1.0f - 1e-3f);
      const float m = maskf[i];
      e = -(m * logf(oc) + (1.0f - m) * logf(1.0f - oc));
      valid = true;
    };
    int64_t kk;
    float t = rank_select(bce, vw, R, P, cfg.mask_p_ratio, true, lds, &kk) * cfg.lambda_mask;
    if (threadIdx.x == 0) terms[3] = t;
    __syncthreads();
    const float c = kk > 0 ? cfg.lambda_mask / (float)kk : 0.0f;
    for (int i = threadIdx.x; i < R; i += blockDim.x) {
      const float o = comp[i * 8];
      const float oc = fminf(fmaxf(o, 1e-3f), 1.0f - 1e-3f);
      const float m = maskf[i];
      const bool pass = o >= 1e-3f && o <= 1.0f - 1e-3f;
      const float g = lds.gerr[i] * c;
      d_comp[i * 8] = pass ? g * (-(m / oc) + (1.0f - m) / (1.0f - oc)) : 0.0f;
      d_comp[i * 8 + 1] = 0.0f;                 // depth enters no loss
    }
  }
}

// i in [0, n_s): samples (eikonal);  i in [n_s, n_s + n_r): random points r = i - n_s (sparsity on
// their sdf, smoothness between their gradient and the gradient at the perturbed copy n_s+n_r+r)
__global__ __launch_bounds__(256) void sample_losses_kernel(
    const float* __restrict__ sdf, const float* __restrict__ grad, int64_t n_s, int64_t n_r,
    float lambda_eik, float lambda_sp, float sp_scale, float lambda_sm, int accumulate_prefix,
    float* __restrict__ d_sdf, float* __restrict__ d_grad, float* __restrict__ terms) {
  float s_eik = 0.0f, s_sp = 0.0f, s_sm = 0.0f;
  const int64_t n = n_s + n_r;
  const float inv_ns = n_s > 0 ? 1.0f / (float)n_s : 0.0f;
  const float inv_nr = n_r > 0 ? 1.0f / (float)n_r : 0.0f;
  const float inv_3nr = n_r > 0 ? 1.0f / (float)(3 * n_r) : 0.0f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n_s) {
      const float g0 = grad[i * 3], g1 = grad[i * 3 + 1], g2 = grad[i * 3 + 2];
      const float nrm = sqrtf((g0 * g0 + g1 * g1) + g2 * g2);
      const float dl = nrm - 1.0f;
      s_eik += dl * dl;
      const float cf = nrm > 0.0f ? lambda_eik * 2.0f * dl * inv_ns / nrm : 0.0f;
      if (accumulate_prefix) {
        d_grad[i * 3] += cf * g0;
        d_grad[i * 3 + 1] += cf * g1;
        d_grad[i * 3 + 2] += cf * g2;
      } else {
        d_grad[i * 3] = cf * g0;
        d_grad[i * 3 + 1] = cf * g1;
        d_grad[i * 3 + 2] = cf * g2;
        d_sdf[i] = 0.0f;
      }
    } else {
      const int64_t a = i, b = i + n_r;
      const float x = sdf[a];
      const float ex = expf(-sp_scale * fabsf(x));
      s_sp += ex;
      d_sdf[a] = lambda_sp * inv_nr * ex * (-sp_scale) * signf0(x);
      d_sdf[b] = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float df = grad[a * 3 + c] - grad[b * 3 + c];
        s_sm += fabsf(df);
        const float sg = lambda_sm * inv_3nr * signf0(df);
        d_grad[a * 3 + c] = sg;
        d_grad[b * 3 + c] = -sg;
      }
    }
  }
  // one atomic triple per workgroup (thousands of same-address atomics serialise)
  __shared__ float red[3][4];
  s_eik = wave_sum(s_eik);
  s_sp = wave_sum(s_sp);
  s_sm = wave_sum(s_sm);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = s_eik;
    red[1][wave] = s_sp;
    red[2][wave] = s_sm;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float t = (red[threadIdx.x][0] + red[threadIdx.x][1]) +
                    (red[threadIdx.x][2] + red[threadIdx.x][3]);
    const float sc = threadIdx.x == 0 ? inv_ns * lambda_eik
                                      : (threadIdx.x == 1 ? inv_nr * lambda_sp : inv_3nr * lambda_sm);
    if (t != 0.0f) atomicAdd(&terms[threadIdx.x], t * sc);
  }
}

// offsets = exclusive scan(counts); stats = {sum, max}.  One workgroup; every thread owns 8
// consecutive rays per pass (one pass for the <= 8192 rays of a training batch: the first version
// ran one barrier-heavy block scan per 1024 rays, 92 us on the side stream, and whichever
// one-wave-per-SIMD kernel of the main stream shared its CU waited for it).
__global__ __launch_bounds__(1024) void ray_offsets_kernel(const int32_t* __restrict__ counts,
                                                           int64_t n, int32_t* __restrict__ offsets,
                                                           int32_t* __restrict__ stats) {
  __shared__ int red[32];
  constexpr int PER = 8;
  int carry = 0, mx = 0;
  for (int64_t base = 0; base < n; base += (int64_t)blockDim.x * PER) {
    const int64_t i0 = base + (int64_t)threadIdx.x * PER;
    int c[PER], tsum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      c[k] = i0 + k < n ? counts[i0 + k] : 0;
      mx = max(mx, c[k]);
      tsum += c[k];
    }
    int total;
    int ex = carry + block_excl_scan(tsum, red, &total);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (i0 + k < n) offsets[i0 + k] = ex;
      ex += c[k];
    }
    carry += total;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    int m = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = max(m, red[w]);
    stats[0] = carry;
    stats[1] = m;
  }
}

// OrthoNeuSSystem.preprocess_data (systems/neus_ortho.py:26-82, batch_image_sampling) for n random
// (view, y, x) triples: c2w gather, get_ortho_rays (models/ray_utils.py:36-58), colour / normal /
// mask / view-weight gathers, cosine between ray and normal, normalised direction.
__global__ __launch_bounds__(256) void ortho_ray_batch_kernel(
    const int64_t* __restrict__ index, const int64_t* __restrict__ px, const int64_t* __restrict__ py,
    int64_t n, const float* __restrict__ c2w /*(V,3,4)*/, const float* __restrict__ origins,
    const float* __restrict__ directions, const float* __restrict__ images, int img_c,
    const float* __restrict__ normals, const float* __restrict__ masks,
    const float* __restrict__ vweights, int H, int W, float* __restrict__ rays /*(n,6)*/,
    float* __restrict__ rgb, float* __restrict__ normal, float* __restrict__ mask,
    float* __restrict__ cosines, float* __restrict__ vw, float* __restrict__ rays_o /*(n,3) or null*/,
    float* __restrict__ rays_d) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = index[i];
  const int64_t pix = (v * H + py[i]) * W + px[i];
  const float* M = c2w + v * 12;
  const float o0 = origins[pix * 3], o1 = origins[pix * 3 + 1], o2 = origins[pix * 3 + 2];
  const float d0 = directions[pix * 3], d1 = directions[pix * 3 + 1], d2 = directions[pix * 3 + 2];
  float ro[3], rd[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rd[r] = (M[r * 4] * d0 + M[r * 4 + 1] * d1) + M[r * 4 + 2] * d2;
    ro[r] = M[r * 4 + 3] + ((M[r * 4] * o0 + M[r * 4 + 1] * o1) + M[r * 4 + 2] * o2);
  }
  const float nx = normals[pix * 3], ny = normals[pix * 3 + 1], nz = normals[pix * 3 + 2];
  // F.cosine_similarity(rays_d, normal, eps=1e-6): each vector divided by max(|.|, eps)
  const float dn = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
  const float nn = sqrtf((nx * nx + ny * ny) + nz * nz);
  const float da = fmaxf(dn, 1e-6f), na = fmaxf(nn, 1e-6f);
  cosines[i] = ((rd[0] / da) * (nx / na) + (rd[1] / da) * (ny / na)) + (rd[2] / da) * (nz / na);
  const float dz = fmaxf(dn, 1e-12f);                    // F.normalize(rays_d, p=2, eps=1e-12)
  rays[i * 6 + 0] = ro[0]; rays[i * 6 + 1] = ro[1]; rays[i * 6 + 2] = ro[2];
  rays[i * 6 + 3] = rd[0] / dz; rays[i * 6 + 4] = rd[1] / dz; rays[i * 6 + 5] = rd[2] / dz;
  if (rays_o) {
    // the same values as two contiguous (n,3) arrays (what the marcher and the compositing kernels
    // read): the native step used to copy them out of `rays` with two strided copies per step
    rays_o[i * 3] = ro[0]; rays_o[i * 3 + 1] = ro[1]; rays_o[i * 3 + 2] = ro[2];
    rays_d[i * 3] = rd[0] / dz; rays_d[i * 3 + 1] = rd[1] / dz; rays_d[i * 3 + 2] = rd[2] / dz;
  }
  for (int c = 0; c < img_c; ++c) rgb[i * img_c + c] = images[pix * img_c + c];
  normal[i * 3] = nx; normal[i * 3 + 1] = ny; normal[i * 3 + 2] = nz;
  mask[i] = masks[pix];
  vw[i] = vweights[pix];
}


// torch.optim.AdamW on the hash-table parameters (neus_ortho.py / configs: AdamW, betas (0.9,
// 0.99), eps 1e-15, default weight decay 0.01) for the ACTIVE levels only, fused with what the
// step needs around it: the f16 image the kernels read is rewritten and the gradient zeroed in
// the same pass (torch: 215 MB of optimizer traffic + a 31 MB fill + a 46 MB f32->f16 pass per step
// over all 3.8 M entries, 80 % of which belong to levels the progressive schedule has not
// switched on yet and only see `p *= 1 - lr * wd`: that factor is applied lazily by
// table_decay_kernel when a level is switched on / at the end).
__global__ __launch_bounds__(256) void table_adamw_kernel(dsu_table_adamw_args a) {
  dsu_table_adamw_range(a, blockIdx.x * (int64_t)blockDim.x + threadIdx.x,
                        (int64_t)gridDim.x * blockDim.x);
}

// torch.optim.AdamW for the model's small tensors (SDF MLP, texture MLP, variance: 13 tensors of
// 1..4096 elements in three parameter groups) in ONE launch: torch's fused optimizer needs a
// step-counter launch + an update launch per group (6 launches, ~45 us of a 1.45 ms step).
struct AdamwMultiArgs {
  dsu_adamw_tensor t[DSU_ADAMW_MAX_TENSORS];
  float beta1, beta2, eps, wd;
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamwMultiArgs a) {
  const dsu_adamw_tensor t = a.t[blockIdx.y];
  const float step_size = t.lr / t.bias_correction1;
  const float omb1 = (float)(1.0 - (double)a.beta1), omb2 = (float)(1.0 - (double)a.beta2);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < t.n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float g = t.g[i];
    float x = t.p[i];
    x -= t.lr * a.wd * x;
    const float mk = t.m[i] + (g - t.m[i]) * omb1;
    const float vk = a.beta2 * t.v[i] + omb2 * g * g;
    x -= step_size * mk / (sqrtf(vk) / t.bias_correction2_sqrt + a.eps);
    t.p[i] = x; t.m[i] = mk; t.v[i] = vk;
  }
}

__global__ __launch_bounds__(256) void table_decay_kernel(float4* __restrict__ p,
                                                          __half2* __restrict__ img, int64_t n4,
                                                          float factor) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 P = p[i];
    P.x *= factor; P.y *= factor; P.z *= factor; P.w *= factor;
    p[i] = P;
    img[2 * i] = __floats2half2_rn(P.x, P.y);
    img[2 * i + 1] = __floats2half2_rn(P.z, P.w);
  }
}

}  // namespace

extern "C" {

int dsu_table_adamw(float* p, float* g, float* m, float* v, void* img_f16, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                    float bias_correction2_sqrt, void* stream) {
  if (n < 0 || (n & 3) || (n && (!p || !g || !m || !v || !img_f16))) return DSU_EINVAL;
  if (!(bias_correction1 > 0.0f) || !(bias_correction2_sqrt > 0.0f)) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  table_adamw_kernel<<<dsu_capped_blocks(n / 4, 256, 2048), 256, 0, (hipStream_t)stream>>>(
      dsu_table_adamw_args{(float4*)p, (float4*)g, (float4*)m, (float4*)v, (__half2*)img_f16, n / 4,
                           lr, beta1, beta2, eps, weight_decay, bias_correction1,
                           bias_correction2_sqrt});
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_adamw_multi(const dsu_adamw_tensor* tensors, int32_t count, float beta1, float beta2,
                    float eps, float weight_decay, void* stream) {
  if (count < 0 || count > DSU_ADAMW_MAX_TENSORS || (count && !tensors)) return DSU_EINVAL;
  if (count == 0) return DSU_OK;
  AdamwMultiArgs a;
  int64_t nmax = 0;
  for (int i = 0; i < count; ++i) {
    const dsu_adamw_tensor& t = tensors[i];
    if (t.n < 0 || (t.n && (!t.p || !t.g || !t.m || !t.v))) return DSU_EINVAL;
    if (!(t.bias_correction1 > 0.0f) || !(t.bias_correction2_sqrt > 0.0f)) return DSU_EINVAL;
    a.t[i] = t;
    if (t.n > nmax) nmax = t.n;
  }
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  if (nmax == 0) return DSU_OK;
  adamw_multi_kernel<<<dim3(dsu_capped_blocks(nmax, 256, 64), count), 256, 0, (hipStream_t)stream>>>(a);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_table_decay(float* p, void* img_f16, int64_t n, float factor, void* stream) {
  if (n < 0 || (n & 3) || (n && (!p || !img_f16))) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  table_decay_kernel<<<dsu_capped_blocks(n / 4, 256, 2048), 256, 0, (hipStream_t)stream>>>(
      (float4*)p, (__half2*)img_f16, n / 4, factor);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ortho_ray_batch(const int64_t* index, const int64_t* x, const int64_t* y, int64_t n,
                        const float* c2w, const float* origins, const float* directions,
                        const float* images, int32_t image_channels, const float* normals,
                        const float* masks, const float* view_weights, int32_t H, int32_t W,
                        float* rays, float* rgb, float* normal, float* mask, float* cosines,
                        float* vw, void* stream) {
  return dsu_ortho_ray_batch_split(index, x, y, n, c2w, origins, directions, images, image_channels,
                                   normals, masks, view_weights, H, W, rays, rgb, normal, mask,
                                   cosines, vw, nullptr, nullptr, stream);
}

int dsu_ortho_ray_batch_split(const int64_t* index, const int64_t* x, const int64_t* y, int64_t n,
                              const float* c2w, const float* origins, const float* directions,
                              const float* images, int32_t image_channels, const float* normals,
                              const float* masks, const float* view_weights, int32_t H, int32_t W,
                              float* rays, float* rgb, float* normal, float* mask, float* cosines,
                              float* vw, float* rays_o, float* rays_d, void* stream) {
  if (n < 0 || H <= 0 || W <= 0 || image_channels <= 0) return DSU_EINVAL;
  if ((rays_o == nullptr) != (rays_d == nullptr)) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  if (!index || !x || !y || !c2w || !origins || !directions || !images || !normals || !masks ||
      !view_weights || !rays || !rgb || !normal || !mask || !cosines || !vw)
    return DSU_EINVAL;
  ortho_ray_batch_kernel<<<dsu_blocks_for(n, 256), 256, 0, (hipStream_t)stream>>>(
      index, x, y, n, c2w, origins, directions, images, image_channels, normals, masks, view_weights,
      H, W, rays, rgb, normal, mask, cosines, vw, rays_o, rays_d);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ray_losses(const float* comp, const float* rgb, const float* normal, const float* mask,
                   const float* cosines, const float* view_weights, int32_t n_rays,
                   const dsu_ray_loss_cfg* cfg, float* terms, float* d_comp, void* stream) {
  if (!cfg || !terms || n_rays < 0) return DSU_EINVAL;
  if (n_rays && (!comp || !rgb || !normal || !mask || !cosines || !view_weights || !d_comp))
    return DSU_EINVAL;
  if (n_rays > DSU_RAY_LOSS_MAX_RAYS) return DSU_EUNSUP;
  int P = RL_THREADS;
  while (P < n_rays) P <<= 1;
  const size_t shm = (size_t)P * 16 + 256;
  DSU_ENSURE_DYN_LDS(ray_losses_kernel, shm);
  ray_losses_kernel<<<3, RL_THREADS, shm, (hipStream_t)stream>>>(
      comp, rgb, normal, mask, cosines, view_weights, n_rays, P, *cfg, terms, d_comp);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_sample_losses(const float* sdf_all, const float* grad_all, int64_t n_samples,
                      int64_t n_random, float lambda_eikonal, float lambda_sparsity,
                      float sparsity_scale, float lambda_smooth, int32_t accumulate_prefix,
                      float* d_sdf_all, float* d_grad_all, float* terms, void* stream) {
  if (n_samples < 0 || n_random < 0 || !terms) return DSU_EINVAL;
  const int64_t n = n_samples + n_random;
  if (n && (!sdf_all || !grad_all || !d_sdf_all || !d_grad_all)) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // bit 1 of accumulate_prefix: the caller has zeroed `terms` already (the step driver does it in
  // the previous step's optimizer kernel: one fill launch less per step)
  if (!(accumulate_prefix & 2) && hipMemsetAsync(terms, 0, 3 * sizeof(float), s) != hipSuccess)
    return DSU_ELAUNCH;
  if (n == 0) return DSU_OK;
  sample_losses_kernel<<<dsu_capped_blocks(n, 256, 512), 256, 0, s>>>(
      sdf_all, grad_all, n_samples, n_random, lambda_eikonal, lambda_sparsity, sparsity_scale,
      lambda_smooth, accumulate_prefix & 1, d_sdf_all, d_grad_all, terms);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_ray_offsets(const int32_t* counts, int64_t n_rays, int32_t* offsets, int32_t* stats,
                    void* stream) {
  if (n_rays < 0 || !stats || (n_rays && (!counts || !offsets))) return DSU_EINVAL;
  ray_offsets_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(counts, n_rays, offsets, stats);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
