// GroupNorm(+SiLU) on NHWC f16, LayerNorm, GEGLU for the diffusion UNet (K4 of SURVEY.md
// §2.3: the GroupNorm/SiLU/LayerNorm/GEGLU kernels the reference runs through diffusers'
// ResnetBlock2D (norm1/norm2 + nonlinearity), TransformerMV2DModel.norm
// (mvdiffusion/models/transformer_mv2d.py:304), BasicMVTransformerBlock.norm1/2/3/norm_joint_mid
// (:532-625) and FeedForward(GEGLU) (:483)).  All HBM-bound: every kernel reads its input once
// with 16-byte accesses and keeps statistics in f32.
#include "common.h"

#include <stdlib.h>

namespace {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- GroupNorm statistics: per (image, channel) partial sums over a slab of pixels, folded
// to per-(image, group) sum / sum-of-squares with one atomic pair per workgroup and group.
// grid = (pixel slabs, B); block = 256 threads; thread t owns 8-channel group (t % C8) of
// pixels t / C8, t / C8 + 256 / C8 ...   (C8 = C/8 <= 256)
__global__ __launch_bounds__(256) void gn_stats_kernel(const f16* __restrict__ x, int HW, int C,
                                                       int G, int pix_per_block,
                                                       float* __restrict__ stats /*(B,G,2)*/) {
  extern __shared__ float sh[];   // [2][C]
  const int n = blockIdx.y;
  const int C8 = C >> 3;
  const int nsplit = (C8 + 255) / 256;               // channel range split when C/8 > 256
  const int lanes_per_pix = C8 / nsplit;             // host guarantees divisibility
  const int pix_par = 256 / lanes_per_pix;           // pixels processed concurrently
  const int pl = threadIdx.x / lanes_per_pix;
  for (int i = threadIdx.x; i < 2 * C; i += 256) sh[i] = 0.0f;
  __syncthreads();
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  for (int sp = 0; sp < nsplit; ++sp) {
    const int cg = sp * lanes_per_pix + threadIdx.x % lanes_per_pix;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.0f;
    if (pl < pix_par) {
      for (int p = p0 + pl; p < p1; p += pix_par) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(x + ((size_t)n * HW + p) * C + cg * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          s[e] += f;
          q[e] += f * f;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        atomicAdd(&sh[cg * 8 + e], s[e]);
        atomicAdd(&sh[C + cg * 8 + e], q[e]);
      }
    }
  }
  __syncthreads();
  const int cpg = C / G;
  for (int g = threadIdx.x; g < G; g += 256) {
    float ss = 0.0f, qq = 0.0f;
    for (int c = 0; c < cpg; ++c) {
      ss += sh[g * cpg + c];
      qq += sh[C + g * cpg + c];
    }
    unsafeAtomicAdd(&stats[((size_t)n * G + g) * 2], ss);
    unsafeAtomicAdd(&stats[((size_t)n * G + g) * 2 + 1], qq);
  }
}

__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }

__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x,
                                                       const float* __restrict__ stats,
                                                       const f16* __restrict__ gamma,
                                                       const f16* __restrict__ beta, int HW,
                                                       int C, int G, float eps, int do_silu,
                                                       int64_t total8, f16* __restrict__ out) {
  const int cpg = C / G;
  const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
  const int C8 = C >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % C8) * 8;
    const int n = (int)(i / ((int64_t)C8 * HW));
    const f16x8 v = *reinterpret_cast<const f16x8*>(x + i * 8);
    const f16x8 gm = *reinterpret_cast<const f16x8*>(gamma + c0);
    const f16x8 bt = *reinterpret_cast<const f16x8*>(beta + c0);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (c0 + e) / cpg;
      const float mean = stats[((size_t)n * G + g) * 2] * inv_cnt;
      const float var = fmaxf(stats[((size_t)n * G + g) * 2 + 1] * inv_cnt - mean * mean, 0.0f);
      float y = ((float)v[e] - mean) * rsqrtf(var + eps) * (float)gm[e] + (float)bt[e];
      if (do_silu) y = silu(y);
      o[e] = (f16)y;
    }
    *reinterpret_cast<f16x8*>(out + i * 8) = o;
  }
}

// ---- small tensors (every GroupNorm of the UNet): ONE launch, one workgroup per (image, group).
// The memset + statistics + apply sequence above costs three dependent launches (25-60 us of
// mostly launch latency on 0.1-6 MB tensors); here the group's HW x cpg elements (<= 64 KB, L2
// resident) are read twice by the same workgroup.
__global__ __launch_bounds__(256) void gn_fused_small_kernel(const f16* __restrict__ x,
                                                             const f16* __restrict__ gamma,
                                                             const f16* __restrict__ beta, int HW,
                                                             int C, int G, float eps, int do_silu,
                                                             f16* __restrict__ out) {
  // channels per group are even for every layer (10..80), so the group's rows are walked as
  // half2 pairs; up to GN_KEEP pairs per thread stay in registers between the two passes
  constexpr int GN_KEEP = 16;
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  __shared__ float red[2][4];
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G, cp2 = cpg >> 1;
  const int cnt2 = HW * cp2;
  const f16* xb = x + (size_t)n * HW * C + g * cpg;
  f16* ob = out + (size_t)n * HW * C + g * cpg;
  const bool keep = cnt2 <= GN_KEEP * 256;
  f16x2 kv[GN_KEEP];
  float s = 0.0f, q = 0.0f;
  if (keep) {
#pragma unroll
    for (int k = 0; k < GN_KEEP; ++k) {
      const int e = threadIdx.x + 256 * k;
      f16x2 v;
      v[0] = (f16)0.0f; v[1] = (f16)0.0f;
      if (e < cnt2) {
        const int p = e / cp2, c = e - p * cp2;
        v = *reinterpret_cast<const f16x2*>(xb + (size_t)p * C + 2 * c);
      }
      kv[k] = v;
      const float f0 = (float)v[0], f1 = (float)v[1];
      s += f0 + f1;
      q += f0 * f0 + f1 * f1;
    }
  } else {
    for (int e = threadIdx.x; e < cnt2; e += 256) {
      const int p = e / cp2, c = e - p * cp2;
      const f16x2 v = *reinterpret_cast<const f16x2*>(xb + (size_t)p * C + 2 * c);
      const float f0 = (float)v[0], f1 = (float)v[1];
      s += f0 + f1;
      q += f0 * f0 + f1 * f1;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
  const float mean = s * inv_cnt;
  const float var = fmaxf(q * inv_cnt - mean * mean, 0.0f);
  const float rstd = rsqrtf(var + eps);
  auto apply = [&](int e, f16x2 v) {
    const int p = e / cp2, c = e - p * cp2;
    const int ch = g * cpg + 2 * c;
    const f16x2 gm = *reinterpret_cast<const f16x2*>(gamma + ch);
    const f16x2 bt = *reinterpret_cast<const f16x2*>(beta + ch);
    float y0 = ((float)v[0] - mean) * rstd * (float)gm[0] + (float)bt[0];
    float y1 = ((float)v[1] - mean) * rstd * (float)gm[1] + (float)bt[1];
    if (do_silu) { y0 = silu(y0); y1 = silu(y1); }
    f16x2 o;
    o[0] = (f16)y0; o[1] = (f16)y1;
    *reinterpret_cast<f16x2*>(ob + (size_t)p * C + 2 * c) = o;
  };
  if (keep) {
#pragma unroll
    for (int k = 0; k < GN_KEEP; ++k) {
      const int e = threadIdx.x + 256 * k;
      if (e < cnt2) apply(e, kv[k]);
    }
  } else {
    for (int e = threadIdx.x; e < cnt2; e += 256) {
      const int p = e / cp2, c = e - p * cp2;
      apply(e, *reinterpret_cast<const f16x2*>(xb + (size_t)p * C + 2 * c));
    }
  }
}

// ---- the same with 16-byte accesses: one 1024-thread workgroup per (image, SUPER-group), a
// super-group being the smallest run of whole groups that is also a whole number of 8-channel
// chunks (10 channels per group -> 4 groups = 40 channels = 5 chunks).  The kernel above reads a
// group's 10..80 channels of a pixel as half2 pairs, 20..160 bytes out of every 640..5120-byte
// pixel row: 13 cache lines per wave instruction, each line fetched again by the 3-6 other
// workgroups whose groups share it (28.7 us for the 7.9 MB of a (12, 32x32, 320) tensor = 0.55 TB/s).
// Here a thread keeps its <= GS_KEEP chunks in registers between the statistics and the apply pass.
constexpr int GS_KEEP = 16;
constexpr int GS_THREADS = 1024;
__global__ __launch_bounds__(GS_THREADS) void gn_super_kernel(const f16* __restrict__ x,
                                                              const f16* __restrict__ gamma,
                                                              const f16* __restrict__ beta, int HW,
                                                              int C, int cpg, int SG, float eps,
                                                              int do_silu, f16* __restrict__ out) {
  __shared__ float red[GS_THREADS / 64][8];
  __shared__ float stat[8];                       // mean[4] | rstd[4]
  const int nsg = C / SG, CH = SG >> 3;
  const int n = blockIdx.x / nsg, c_base = (blockIdx.x % nsg) * SG;
  const int total = HW * CH;
  const f16* xb = x + (size_t)n * HW * C + c_base;
  f16* ob = out + (size_t)n * HW * C + c_base;
  f16x8 kv[GS_KEEP];
  int off[GS_KEEP];                                // element offset of the chunk, -1 = none
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < GS_KEEP; ++k) {
    const int e = threadIdx.x + GS_THREADS * k;
    off[k] = -1;
    if (e < total) {
      const int p = e / CH, c = e - p * CH;
      off[k] = p * C + 8 * c;
      const f16x8 v = *reinterpret_cast<const f16x8*>(xb + off[k]);
      kv[k] = v;
      // cpg >= 8: a chunk touches at most two groups, g0 for its first `split` channels
      const int g0 = (8 * c) / cpg, split = (g0 + 1) * cpg - 8 * c;
      float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        if (j < split) { s0 += f; q0 += f * f; } else { s1 += f; q1 += f * f; }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s[g] += g == g0 ? s0 : (g == g0 + 1 ? s1 : 0.0f);
        q[g] += g == g0 ? q0 : (g == g0 + 1 ? q1 : 0.0f);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s[g] += __shfl_xor(s[g], o);
      q[g] += __shfl_xor(q[g], o);
    }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      red[threadIdx.x >> 6][g] = s[g];
      red[threadIdx.x >> 6][4 + g] = q[g];
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float ss = 0.f, qq = 0.f;
    for (int w = 0; w < GS_THREADS / 64; ++w) { ss += red[w][threadIdx.x]; qq += red[w][4 + threadIdx.x]; }
    const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
    const float mean = ss * inv_cnt;
    const float var = fmaxf(qq * inv_cnt - mean * mean, 0.0f);
    stat[threadIdx.x] = mean;
    stat[4 + threadIdx.x] = rsqrtf(var + eps);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < GS_KEEP; ++k) {
    if (off[k] < 0) continue;
    const int c8 = off[k] % C;                     // 8 * chunk index inside the super-group
    const int g0 = c8 / cpg, split = (g0 + 1) * cpg - c8;
    const float m0 = stat[g0], r0 = stat[4 + g0];
    const float m1 = stat[g0 + 1 < 4 ? g0 + 1 : 3], r1 = stat[4 + (g0 + 1 < 4 ? g0 + 1 : 3)];
    const f16x8 gm = *reinterpret_cast<const f16x8*>(gamma + c_base + c8);
    const f16x8 bt = *reinterpret_cast<const f16x8*>(beta + c_base + c8);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mean = j < split ? m0 : m1, rstd = j < split ? r0 : r1;
      float y = ((float)kv[k][j] - mean) * rstd * (float)gm[j] + (float)bt[j];
      if (do_silu) y = silu(y);
      o[j] = (f16)y;
    }
    *reinterpret_cast<f16x8*>(ob + off[k]) = o;
  }
}

// ---- LayerNorm over the last dimension: one wave per row, C % 8 == 0, C <= 64*8*4
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x,
                                                        const f16* __restrict__ gamma,
                                                        const f16* __restrict__ beta,
                                                        int64_t rows, int C, float eps,
                                                        f16* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int C8 = C >> 3;
  f16x8 v[4];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = lane + 64 * i;
    if (g < C8) {
      v[i] = *reinterpret_cast<const f16x8*>(x + row * C + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[i][e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = lane + 64 * i;
    if (g < C8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[i][e] - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = lane + 64 * i;
    if (g < C8) {
      const f16x8 gm = *reinterpret_cast<const f16x8*>(gamma + g * 8);
      const f16x8 bt = *reinterpret_cast<const f16x8*>(beta + g * 8);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = (f16)(((float)v[i][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
      *reinterpret_cast<f16x8*>(out + row * C + g * 8) = o;
    }
  }
}

// ---- GEGLU: out[r][j] = h[r][j] * gelu(h[r][D + j])   (diffusers GEGLU: exact erf GELU)
__global__ __launch_bounds__(256) void geglu_kernel(const f16* __restrict__ h, int64_t rows,
                                                    int D, f16* __restrict__ out) {
  const int D8 = D >> 3;
  const int64_t total = rows * D8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D8;
    const int j = (int)(i % D8) * 8;
    const f16x8 a = *reinterpret_cast<const f16x8*>(h + r * 2 * D + j);
    const f16x8 g = *reinterpret_cast<const f16x8*>(h + r * 2 * D + D + j);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gv = (float)g[e];
      const float gelu = 0.5f * gv * (1.0f + erff(gv * 0.70710678118654752f));
      o[e] = (f16)((float)a[e] * gelu);
    }
    *reinterpret_cast<f16x8*>(out + r * D + j) = o;
  }
}

}  // namespace

extern "C" {

int dsu_groupnorm_nhwc_f16(const void* x, const void* gamma, const void* beta, int32_t B,
                           int32_t HW, int32_t C, int32_t G, float eps, int32_t silu,
                           float* stats_ws, void* out, void* stream) {
  if (!x || !gamma || !beta || !stats_ws || !out) return DSU_EINVAL;
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0) return DSU_EINVAL;
  if (C % 8 != 0 || C / 8 > 1024 || (C / 8) % ((C / 8 + 255) / 256) != 0) return DSU_EUNSUP;
  hipStream_t s = (hipStream_t)stream;
  {
    // super-group form: groups of >= 8 channels whose super-group has <= 4 groups and fits the
    // registers of one 1024-thread workgroup (DSU_GN_SUPER=0: A/B switch)
    static const bool use_super = dsu_ab_int("DSU_GN_SUPER", 1) != 0;
    const int cpg = C / G;
    int SG = cpg;
    while (SG % 8 != 0) SG += cpg;
    // (measured per UNet shape, tools/gn_time.py: 19-47 us instead of 26-58 at 32x32, 15.6 instead of
    // 20.1 at 16x16x1280, a tie at 16x16x640, 2-3 us SLOWER on the 8x8 / 4x4 levels — few, fat workgroups)
    if (use_super && HW >= 256 && cpg >= 8 && SG / cpg <= 4 && C % SG == 0 &&
        (int64_t)HW * (SG / 8) <= (int64_t)GS_KEEP * GS_THREADS && (int64_t)HW * C < ((int64_t)1 << 31)) {
      gn_super_kernel<<<B * (C / SG), GS_THREADS, 0, s>>>((const f16*)x, (const f16*)gamma,
                                                          (const f16*)beta, HW, C, cpg, SG, eps, silu,
                                                          (f16*)out);
      DSU_CHECK_LAUNCH();
      return DSU_OK;
    }
  }
  if ((int64_t)HW * (C / G) <= 32768 && (int64_t)B * G >= 64 && (C / G) % 2 == 0) {
    gn_fused_small_kernel<<<B * G, 256, 0, s>>>((const f16*)x, (const f16*)gamma,
                                                (const f16*)beta, HW, C, G, eps, silu, (f16*)out);
    DSU_CHECK_LAUNCH();
    return DSU_OK;
  }
  if (hipMemsetAsync(stats_ws, 0, (size_t)B * G * 2 * sizeof(float), s) != hipSuccess)
    return DSU_ELAUNCH;
  const int pix_per_block = 64;
  dim3 grid((HW + pix_per_block - 1) / pix_per_block, B);
  gn_stats_kernel<<<grid, 256, 2 * C * sizeof(float), s>>>((const f16*)x, HW, C, G,
                                                           pix_per_block, stats_ws);
  const int64_t total8 = (int64_t)B * HW * (C / 8);
  gn_apply_kernel<<<dsu_capped_blocks(total8, 256, 4096), 256, 0, s>>>(
      (const f16*)x, stats_ws, (const f16*)gamma, (const f16*)beta, HW, C, G, eps, silu, total8,
      (f16*)out);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_layernorm_f16(const void* x, const void* gamma, const void* beta, int64_t rows,
                      int32_t C, float eps, void* out, void* stream) {
  if (!x || !gamma || !beta || !out || rows < 0 || C <= 0) return DSU_EINVAL;
  if (C % 8 != 0 || C > 2048) return DSU_EUNSUP;
  if (rows == 0) return DSU_OK;
  layernorm_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      (const f16*)x, (const f16*)gamma, (const f16*)beta, rows, C, eps, (f16*)out);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_geglu_f16(const void* h, int64_t rows, int32_t D, void* out, void* stream) {
  if (!h || !out || rows < 0 || D <= 0) return DSU_EINVAL;
  if (D % 8 != 0) return DSU_EUNSUP;
  if (rows == 0) return DSU_OK;
  geglu_kernel<<<dsu_capped_blocks(rows * (D / 8), 256, 4096), 256, 0, (hipStream_t)stream>>>(
      (const f16*)h, rows, D, (f16*)out);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
