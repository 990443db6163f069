// Training-side kernels of the per-character style translator:
// the backward passes and train-mode normalisations that
// 3_style_translator/training/trainers.py:140-192 runs through autograd + cuDNN/torchvision
// in the reference (GeneratorJ / GeneratorJ_RIC, DiscriminatorN_IN, PerceptualVGG19 on
// 40 x C x 32 x 32 patches, configs/config_stage1.yaml:38-62).
//
//   * conv_wgrad_kernel      dW = dout x im2col(in)^T on the f32 MFMA, for nn.Conv2d
//                            (k = 1/3/4/7, stride 1/2) and for the fixed-offset 3x3 deformable
//                            convolution (models.py:302-351; the offsets come from
//                            generate_coordinates and are not learned, so the op is linear in
//                            its input and only dX / dW exist).
//   * deform_gather_kernel   dX of the deformable convolution = S^T (W^T dout): the W^T dout
//                            product is a 1x1 convolution (dsu_conv2d_fwd), S^T is applied as a
//                            gather over a transposed (CSR) sampling table — no atomics, fixed
//                            summation order.
//   * norm_fwd/bwd_kernel    BatchNorm2d in training mode (batch statistics, running-stat
//                            update) and InstanceNorm2d (DiscriminatorN_IN, models.py:436-439),
//                            with the following LeakyReLU / ReLU fused.
//   * pooling / resampling / activation / loss kernels around them.
//
// The data-gradient of a plain convolution is the forward kernel (style_conv.hip) run on dout
// with the spatially flipped, channel-transposed weights; it needs no kernel of its own.
#include "common.h"
#include "style_dev.h"

namespace {
using namespace dsu_style;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// fixed-offset sampling table: one record per (pixel, tap)
// ------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) TapRec {
  int i00, i01, i10, i11;      // plane offsets of the four corners (clamped)
  float w00, w01, w10, w11;    // bilinear weights, zero for corners outside the image
};

__global__ void deform_tap_table_kernel(const float* __restrict__ offset, int H, int W,
                                        TapRec* __restrict__ table) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int npix = H * W;
  if (idx >= npix * 9) return;
  const int pix = idx / 9, t = idx - pix * 9;
  const int oy = pix / W, ox = pix - oy * W;
  const float dh = offset[(size_t)(2 * t) * npix + pix];
  const float dw = offset[(size_t)(2 * t + 1) * npix + pix];
  // same expression as the forward kernel (style_conv.hip), pad = 1
  const float h = (float)(oy - 1 + t / 3) + dh;
  const float w = (float)(ox - 1 + t % 3) + dw;
  const Tap tp = make_tap(h, w, H, W);
  TapRec r;
  r.i00 = tp.r0 + tp.c0; r.i01 = tp.r0 + tp.c1; r.i10 = tp.r1 + tp.c0; r.i11 = tp.r1 + tp.c1;
  r.w00 = tp.w00; r.w01 = tp.w01; r.w10 = tp.w10; r.w11 = tp.w11;
  table[idx] = r;
}

// ------------------------------------------------------------------------------------------
// weight gradient
//   GEMM view: dW[o][n] = sum_g dout[o][g] * col[g][n],  g = (image, output pixel),
//   n = (input channel, tap) flattened as in the weight tensor.
//   A operand (M = o) = dout, B operand (N = n) = im2col values, K = pixels.
//   Workgroup: all O (<=128) x TN columns, a slice of the pixel range; partial sums per slice
//   are written out and added by wgrad_reduce_kernel (fixed order: deterministic).
// ------------------------------------------------------------------------------------------
constexpr int WG_PB = 128;   // pixels per K chunk
constexpr int WG_TN = 64;    // n columns per workgroup (MODE 1: 7 channels x 9 taps = 63 used)

struct WgradArgs {
  const float* in;       // (B,C,H,W)
  const float* dout;     // (B,O,OH,OW)
  const TapRec* taps;    // MODE 1: (OH*OW, 9)
  float* partial;        // (slices, O, ntot)
  int B, C, H, W, O, OH, OW, stride, pad;
  int ntot;              // C * KS * KS
  int nchunks;           // ceil(B*OH*OW / WG_PB)
  int chunks_per_slice;
};

template <int MODE, int KS, int MB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int KK = KS * KS;
  constexpr int MT = 32 * MB;              // output-channel rows held in LDS
  constexpr int SD = MT + 1, SC = WG_TN + 1;
  constexpr int ND = MT / 2;               // dout values staged per thread
  constexpr int NACC = (MB + 1) / 2;
  constexpr int TNR = MODE == 1 ? 63 : WG_TN;   // real columns per tile
  extern __shared__ float smem[];
  float* sD = smem;                        // [WG_PB][SD]  dout, pixel-major
  float* sC = smem + WG_PB * SD;           // [WG_PB][SC]  im2col values, pixel-major

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int n0 = blockIdx.x * TNR;
  const int npix = a.OH * a.OW;
  const int64_t total = (int64_t)a.B * npix;
  const int q_begin = blockIdx.y * a.chunks_per_slice;
  const int q_end = min(q_begin + a.chunks_per_slice, a.nchunks);
  const int p = tid & (WG_PB - 1), half = tid >> 7;
  const size_t plane = (size_t)a.H * a.W;

  // MODE 0: the (channel, tap) of each of this thread's 32 columns, packed once
  uint32_t pk[MODE == 0 ? 32 : 1];
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int n = n0 + half * 32 + j;
      const int c = n / KK, t = n - c * KK;
      const int ty = t / KS, tx = t - ty * KS;
      pk[j] = n < a.ntot ? (((uint32_t)c << 8) | ((uint32_t)ty << 4) | (uint32_t)tx) : 0xFFFFFFFFu;
    }
  }

  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  // raw values of the next chunk, loaded before the MFMA loop and consumed after it
  constexpr int NRAW = MODE == 1 ? 36 * 4 : 32;
  float rv[NRAW];
  float rwt[MODE == 1 ? 36 : 1];     // MODE 1: bilinear weights (4 per tap, 9 taps)
  float rd[ND];
  // All staging loads are issued unconditionally from clamped (always mapped) addresses and
  // invalid values are zeroed when they are written to LDS: a `cond ? load : 0` per element
  // compiled to one exec-mask branch region per load (330 of them in the first version).
  uint32_t okm = 0;                  // MODE 0: validity of the 32 staged columns
  uint32_t cvm = 0;                  // MODE 1: validity of the 4 staged channels
  bool dvalid = false;

  auto stage_load = [&](int q) {
    const int64_t g = (int64_t)q * WG_PB + p;
    const bool valid = g < total;
    const int b = valid ? (int)(g / npix) : 0;
    const int pix = valid ? (int)(g - (int64_t)b * npix) : 0;
    // ---- dout[b][o][pix], o in [half*ND, half*ND + ND)
    const float* dp = a.dout + ((size_t)b * a.O) * npix + pix;
    dvalid = valid;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int o = half * ND + j;
      rd[j] = dp[(size_t)min(o, a.O - 1) * npix];
    }
    if constexpr (MODE == 1) {
      const TapRec* tr = a.taps + (size_t)pix * 9;
      int ti[36];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int4 ii = *reinterpret_cast<const int4*>(&tr[t].i00);
        const float4 ww = *reinterpret_cast<const float4*>(&tr[t].w00);
        ti[4 * t + 0] = ii.x; ti[4 * t + 1] = ii.y; ti[4 * t + 2] = ii.z; ti[4 * t + 3] = ii.w;
        rwt[4 * t + 0] = ww.x; rwt[4 * t + 1] = ww.y; rwt[4 * t + 2] = ww.z; rwt[4 * t + 3] = ww.w;
      }
      // half 0: tile channels 0..3, half 1: tile channels 4..6
      cvm = 0;
#pragma unroll
      for (int cl = 0; cl < 4; ++cl) {
        const int c = blockIdx.x * 7 + half * 4 + cl;
        const bool cv = valid & (c < a.C) & ((half * 4 + cl) < 7);
        cvm |= cv ? (1u << cl) : 0u;
        const float* pl = a.in + ((size_t)b * a.C + (cv ? c : 0)) * plane;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
          for (int k = 0; k < 4; ++k) rv[(cl * 9 + t) * 4 + k] = pl[ti[4 * t + k]];
        }
      }
    } else {
      const int oy = pix / a.OW, ox = pix - oy * a.OW;
      const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
      const float* ib = a.in + (size_t)b * a.C * plane;
      okm = 0;
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const uint32_t u = pk[j];
        const uint32_t c = u >> 8;
        const int iy = iy0 + (int)((u >> 4) & 15u), ix = ix0 + (int)(u & 15u);
        // branch-free validity (a short-circuit && chain became nested exec-mask branches)
        const bool ok = valid & (u != 0xFFFFFFFFu) & ((uint32_t)iy < (uint32_t)a.H) &
                        ((uint32_t)ix < (uint32_t)a.W);
        okm |= ok ? (1u << j) : 0u;
        // 32-bit element offsets: the host rejects tensors of 2^31 elements or more
        const uint32_t off = c * (uint32_t)plane + (uint32_t)iy * (uint32_t)a.W + (uint32_t)ix;
        rv[j] = ib[ok ? off : 0u];
      }
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int j = 0; j < ND; ++j)
      sD[p * SD + half * ND + j] = (dvalid & (half * ND + j < a.O)) ? rd[j] : 0.0f;
    if constexpr (MODE == 1) {
#pragma unroll
      for (int cl = 0; cl < 4; ++cl) {
        if (half * 4 + cl < 7) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            // same association as the forward kernel's sample
            const float s = rwt[4 * t + 0] * rv[(cl * 9 + t) * 4 + 0] +
                            rwt[4 * t + 1] * rv[(cl * 9 + t) * 4 + 1] +
                            rwt[4 * t + 2] * rv[(cl * 9 + t) * 4 + 2] +
                            rwt[4 * t + 3] * rv[(cl * 9 + t) * 4 + 3];
            sC[p * SC + (half * 4 + cl) * 9 + t] = ((cvm >> cl) & 1u) ? s : 0.0f;
          }
        }
      }
      if (half == 1) sC[p * SC + 63] = 0.0f;   // padding column
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) sC[p * SC + half * 32 + j] = ((okm >> j) & 1u) ? rv[j] : 0.0f;
    }
  };

  const int nblk = wave & 1;
  // accumulator i of this wave covers output-channel block mrow[i]; always a valid block, so the
  // MFMAs below are unconditional.  (A `if (m < MB)` around them made the accumulators flow
  // through phi copies: ~100 v_accvgpr_mov + an s_nop 15 per MFMA in the first version.)  With
  // one block (MB == 1) waves 2 and 3 repeat block 0 and do not write.
  int mrow[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) mrow[i] = MB == 1 ? 0 : (wave >> 1) + 2 * i;
  const bool writer = MB > 1 || (wave >> 1) == 0;
  if (q_begin < q_end) {
    stage_load(q_begin);
    stage_store();
    __syncthreads();
    for (int q = q_begin; q < q_end; ++q) {
      if (q + 1 < q_end) stage_load(q + 1);
#pragma unroll 4
      for (int kp = 0; kp < WG_PB / 2; ++kp) {
        const int pkx = 2 * kp + kh;
        const float bv = sC[pkx * SC + nblk * 32 + l31];
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          const float av = sD[pkx * SD + mrow[i] * 32 + l31];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
        }
      }
      __syncthreads();
      if (q + 1 < q_end) stage_store();
      __syncthreads();
    }
  }

  // ---- epilogue: accumulator row = output channel, column (lane & 31) = n
  const int nl = nblk * 32 + l31;
  const int n = n0 + nl;
  if (writer && nl < TNR && n < a.ntot) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = mrow[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (o < a.O)
          a.partial[((size_t)blockIdx.y * a.O + o) * a.ntot + n] = acc[i][r];
      }
    }
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int slices, int64_t count,
                                    float* __restrict__ dw, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = accumulate ? dw[i] : 0.0f;
  for (int k = 0; k < slices; ++k) s += partial[(size_t)k * count + i];
  dw[i] = s;
}

// ------------------------------------------------------------------------------------------
// dX of the deformable convolution: gather through the transposed sampling table
//   dcol (B, C*9, npix) = W^T dout;  dX[b][c][q] = sum_e w[e] * dcol[b][c*9 + t_e][p_e]
//   with src[e] = t_e * npix + p_e, rows of the CSR = input pixels q.
// ------------------------------------------------------------------------------------------
constexpr int GATHER_PL = 8;   // planes per thread: the (src, weight) pair of an entry is read once for 8 gathers

__global__ __launch_bounds__(256) void deform_gather_kernel(
    const float* __restrict__ dcol, const int* __restrict__ rowptr, const int* __restrict__ src,
    const float* __restrict__ wgt, int64_t planes, int npix, float* __restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t groups = (planes + GATHER_PL - 1) / GATHER_PL;
  if (idx >= groups * npix) return;
  const int64_t pg = idx / npix;
  const int q = (int)(idx - pg * npix);
  const int64_t p0 = pg * GATHER_PL;
  const size_t pstride = (size_t)9 * npix;
  const float* base = dcol + (size_t)p0 * pstride;
  const int np = planes - p0 < GATHER_PL ? (int)(planes - p0) : GATHER_PL;
  const int e0 = rowptr[q], e1 = rowptr[q + 1];
  float s[GATHER_PL];
#pragma unroll
  for (int j = 0; j < GATHER_PL; ++j) s[j] = 0.0f;
  if (np == GATHER_PL) {
    for (int e = e0; e < e1; ++e) {
      const float w = wgt[e];
      const float* b = base + src[e];
#pragma unroll
      for (int j = 0; j < GATHER_PL; ++j) s[j] += w * b[(size_t)j * pstride];
    }
  } else {
    for (int e = e0; e < e1; ++e) {
      const float w = wgt[e];
      const float* b = base + src[e];
#pragma unroll
      for (int j = 0; j < GATHER_PL; ++j)
        if (j < np) s[j] += w * b[(size_t)j * pstride];
    }
  }
#pragma unroll
  for (int j = 0; j < GATHER_PL; ++j)
    if (j < np) dx[(size_t)(p0 + j) * npix + q] = s[j];
}

// ------------------------------------------------------------------------------------------
// block reductions
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over a 256-thread block; every thread receives the result
__device__ __forceinline__ float block_sum(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();               // red may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------
// BatchNorm2d (training) / InstanceNorm2d with fused activation
//   group g covers n_outer slabs of `inner` contiguous elements:
//     element(k, i) = x[g * group_stride + k * outer_stride + i]
//   BatchNorm: groups = C, group_stride = HW, n_outer = B, outer_stride = C*HW
//   InstanceNorm: groups = B*C, group_stride = HW, n_outer = 1
// ------------------------------------------------------------------------------------------
struct NormArgs {
  const float* x;
  const float* y_in;      // bwd: forward output (activation derivative)
  const float* dy;        // bwd
  float* out;             // fwd: y, bwd: dx
  const float* gamma;     // per channel, may be null
  const float* beta;
  float* mean;            // per group (fwd: written, bwd: read)
  float* invstd;
  float* running_mean;    // BatchNorm only, may be null
  float* running_var;
  float* dgamma;          // bwd, may be null
  float* dbeta;
  int64_t group_stride, outer_stride;
  int groups, n_outer, inner, channels;
  float eps, momentum;
  int stat_updates;       // how many times the running statistics take this batch
  int act;                // 0 none, 1 ReLU, 2 LeakyReLU(0.2)
};

// Thread-strided walk over a group's n = n_outer * inner elements without a division per
// element: i = threadIdx.x, += 256; (k, r) = (i / inner, i % inner) kept incrementally.
struct GroupWalk {
  int64_t off;   // k * outer_stride + r
  int r;
  __device__ __forceinline__ GroupWalk(int inner, int64_t outer_stride) {
    const int k = (int)threadIdx.x / inner;
    r = (int)threadIdx.x - k * inner;
    off = (int64_t)k * outer_stride + r;
  }
  __device__ __forceinline__ void next(int inner, int64_t outer_stride) {
    r += 256;
    off += 256;
    while (r >= inner) {
      r -= inner;
      off += outer_stride - inner;
    }
  }
};

__global__ __launch_bounds__(256) void norm_fwd_kernel(NormArgs a) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  const float* xg = a.x + (size_t)g * a.group_stride;
  float* yg = a.out + (size_t)g * a.group_stride;
  const int64_t n = (int64_t)a.n_outer * a.inner;
  float s = 0.0f;
  {
    GroupWalk w(a.inner, a.outer_stride);
    for (int64_t i = threadIdx.x; i < n; i += 256, w.next(a.inner, a.outer_stride)) s += xg[w.off];
  }
  const float mean = block_sum(s, red) / (float)n;
  float v = 0.0f;
  {
    GroupWalk w(a.inner, a.outer_stride);
    for (int64_t i = threadIdx.x; i < n; i += 256, w.next(a.inner, a.outer_stride)) {
      const float d = xg[w.off] - mean;
      v += d * d;
    }
  }
  const float var = block_sum(v, red) / (float)n;
  const float invstd = 1.0f / sqrtf(var + a.eps);
  const int c = g % a.channels;
  const float ga = a.gamma ? a.gamma[c] : 1.0f, be = a.beta ? a.beta[c] : 0.0f;
  {
    GroupWalk w(a.inner, a.outer_stride);
    for (int64_t i = threadIdx.x; i < n; i += 256, w.next(a.inner, a.outer_stride)) {
      const float xh = (xg[w.off] - mean) * invstd;
      yg[w.off] = apply_act(xh * ga + be, a.act);
    }
  }
  if (threadIdx.x == 0) {
    a.mean[g] = mean;
    a.invstd[g] = invstd;
    if (a.running_mean) {
      // nn.BatchNorm2d: running <- (1 - momentum) * running + momentum * batch (unbiased var)
      const float unb = n > 1 ? var * ((float)n / (float)(n - 1)) : var;
      float rm = a.running_mean[c], rvv = a.running_var[c];
      for (int u = 0; u < a.stat_updates; ++u) {
        rm = (1.0f - a.momentum) * rm + a.momentum * mean;
        rvv = (1.0f - a.momentum) * rvv + a.momentum * unb;
      }
      a.running_mean[c] = rm;
      a.running_var[c] = rvv;
    }
  }
}

__device__ __forceinline__ float act_grad_from_output(float y, int act) {
  switch (act) {
    case 1: return y > 0.0f ? 1.0f : 0.0f;
    case 2: return y > 0.0f ? 1.0f : 0.2f;
    case 3: return 1.0f - y * y;
    default: return 1.0f;
  }
}

__global__ __launch_bounds__(256) void norm_bwd_kernel(NormArgs a) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  const size_t gb = (size_t)g * a.group_stride;
  const float* xg = a.x + gb;
  const float* yg = a.y_in + gb;
  const float* dg = a.dy + gb;
  float* og = a.out + gb;
  const int64_t n = (int64_t)a.n_outer * a.inner;
  const float mean = a.mean[g], invstd = a.invstd[g];
  float s1 = 0.0f, s2 = 0.0f;
  {
    GroupWalk w(a.inner, a.outer_stride);
    for (int64_t i = threadIdx.x; i < n; i += 256, w.next(a.inner, a.outer_stride)) {
      const int64_t off = w.off;
      const float gr = dg[off] * act_grad_from_output(yg[off], a.act);
      s1 += gr;
      s2 += gr * ((xg[off] - mean) * invstd);
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  const int c = g % a.channels;
  const float ga = a.gamma ? a.gamma[c] : 1.0f;
  const float m1 = s1 / (float)n, m2 = s2 / (float)n;
  const float k0 = ga * invstd;
  {
    GroupWalk w(a.inner, a.outer_stride);
    for (int64_t i = threadIdx.x; i < n; i += 256, w.next(a.inner, a.outer_stride)) {
      const int64_t off = w.off;
      const float gr = dg[off] * act_grad_from_output(yg[off], a.act);
      const float xh = (xg[off] - mean) * invstd;
      og[off] = k0 * (gr - m1 - xh * m2);
    }
  }
  if (threadIdx.x == 0) {
    if (a.dgamma) a.dgamma[g] = s2;
    if (a.dbeta) a.dbeta[g] = s1;
  }
}

// per-channel sum over (B, HW): bias gradient of a convolution
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, int B, int C,
                                                           int HW, float* __restrict__ out) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const int64_t n = (int64_t)B * HW;
  float s = 0.0f;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t b = i / HW, r = i - b * HW;
    s += x[((size_t)b * C + c) * HW + r];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[c] = s;
}

// ------------------------------------------------------------------------------------------
// element-wise / resampling
// ------------------------------------------------------------------------------------------
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                               int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = apply_act(x[i], act);
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                               float* __restrict__ dx, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = dy[i] * act_grad_from_output(y[i], act);
}

// nn.MaxPool2d(2, 2): planes of H x W -> (H/2) x (W/2)
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                    int64_t planes, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const int64_t n = planes * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    const float* p = x + (size_t)pl * H * W + (size_t)(2 * oy) * W + 2 * ox;
    y[i] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
  }
}

// gradient goes to the first maximum in row-major window order (ATen max_pool2d_with_indices)
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    float* __restrict__ dx, int64_t planes, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const int64_t n = planes * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    const size_t base = (size_t)pl * H * W + (size_t)(2 * oy) * W + 2 * ox;
    const float v[4] = {x[base], x[base + 1], x[base + W], x[base + W + 1]};
    int best = 0;
    float bv = v[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (v[k] > bv) { bv = v[k]; best = k; }
    const float g = dy[i];
    dx[base] = best == 0 ? g : 0.0f;
    dx[base + 1] = best == 1 ? g : 0.0f;
    dx[base + W] = best == 2 ? g : 0.0f;
    dx[base + W + 1] = best == 3 ? g : 0.0f;
  }
}

// nn.Upsample(scale_factor=2) (nearest): planes of H x W -> 2H x 2W
__global__ void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                     int64_t planes, int H, int W) {
  const int OH = 2 * H, OW = 2 * W;
  const int64_t n = planes * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    y[i] = x[(size_t)pl * H * W + (size_t)(oy >> 1) * W + (ox >> 1)];
  }
}

__global__ void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                     int64_t planes, int H, int W) {
  const int OW = 2 * W;
  const int64_t n = planes * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    const int64_t t = i / W;
    const int yy = (int)(t % H);
    const int64_t pl = t / H;
    const float* p = dy + (size_t)pl * 4 * H * W + (size_t)(2 * yy) * OW + 2 * xx;
    dx[i] = (p[0] + p[1]) + (p[OW] + p[OW + 1]);
  }
}

// ------------------------------------------------------------------------------------------
// L1Loss / MSELoss (reduction = mean) against a tensor or a constant; writes 256 partial sums
// (added by the caller in a fixed order) and, optionally, the gradient scaled by grad_scale.
// ------------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 256;

__global__ __launch_bounds__(256) void pair_loss_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ t, float tconst,
                                                        int64_t n, int kind, float grad_scale,
                                                        float* __restrict__ grad,
                                                        float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)LOSS_BLOCKS * 256) {
    const float d = x[i] - (t ? t[i] : tconst);
    if (kind == 0) {
      s += fabsf(d);
      if (grad) grad[i] = d > 0.0f ? grad_scale : (d < 0.0f ? -grad_scale : 0.0f);
    } else {
      s += d * d;
      if (grad) grad[i] = 2.0f * d * grad_scale;
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <int MODE, int KS>
int launch_wgrad(const WgradArgs& a, int slices, hipStream_t s) {
  const int tnr = MODE == 1 ? 63 : WG_TN;
  const int ntiles = MODE == 1 ? (a.C + 6) / 7 : (a.ntot + tnr - 1) / tnr;
  dim3 grid(ntiles, slices);
  const int mb = a.O > 64 ? 4 : (a.O > 32 ? 2 : 1);
  const size_t shm = (size_t)WG_PB * ((32 * mb + 1) + (WG_TN + 1)) * sizeof(float);
  if (mb == 4) {
    DSU_ENSURE_DYN_LDS((conv_wgrad_kernel<MODE, KS, 4>), shm);
    conv_wgrad_kernel<MODE, KS, 4><<<grid, 256, shm, s>>>(a);
  } else if (mb == 2) {
    DSU_ENSURE_DYN_LDS((conv_wgrad_kernel<MODE, KS, 2>), shm);
    conv_wgrad_kernel<MODE, KS, 2><<<grid, 256, shm, s>>>(a);
  } else {
    DSU_ENSURE_DYN_LDS((conv_wgrad_kernel<MODE, KS, 1>), shm);
    conv_wgrad_kernel<MODE, KS, 1><<<grid, 256, shm, s>>>(a);
  }
  if (hipGetLastError() != hipSuccess) return DSU_ELAUNCH;
  return DSU_OK;
}

int wgrad_slices(int ntiles, int nchunks) {
  // the kernel runs one workgroup per CU (99 KB of LDS): aim for at most two full rounds over
  // the 256 CUs (rounding the slice count UP gave 518-532 workgroups = a third, nearly empty
  // round on the two largest layers), at least 2 chunks per slice
  int s = 512 / ntiles;
  if (s > (nchunks + 1) / 2) s = (nchunks + 1) / 2;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return s;
}

}  // namespace

extern "C" {

int64_t dsu_deform_tap_table_bytes(int32_t H, int32_t W) {
  if (H <= 0 || W <= 0) return 0;
  return (int64_t)H * W * 9 * (int64_t)sizeof(TapRec);
}

int dsu_deform_tap_table(const float* offset, int32_t H, int32_t W, void* table, void* stream) {
  if (!offset || !table || H <= 0 || W <= 0) return DSU_EINVAL;
  deform_tap_table_kernel<<<dsu_blocks_for((int64_t)H * W * 9, 256), 256, 0,
                            (hipStream_t)stream>>>(offset, H, W, (TapRec*)table);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

static int wgrad_plan(int32_t mode, int32_t B, int32_t C, int32_t O, int32_t OH, int32_t OW,
                      int32_t k, int* ntiles, int* nchunks, int* slices) {
  const int64_t total = (int64_t)B * OH * OW;
  if (total <= 0 || total > (int64_t)1 << 30) return DSU_EINVAL;
  *nchunks = (int)((total + WG_PB - 1) / WG_PB);
  *ntiles = mode == 1 ? (C + 6) / 7 : (C * k * k + WG_TN - 1) / WG_TN;
  *slices = wgrad_slices(*ntiles, *nchunks);
  (void)O;
  return DSU_OK;
}

int64_t dsu_conv2d_wgrad_workspace_bytes(int32_t deform, int32_t B, int32_t C, int32_t O,
                                         int32_t OH, int32_t OW, int32_t k) {
  if (B <= 0 || C <= 0 || O <= 0 || OH <= 0 || OW <= 0 || k <= 0) return 0;
  int ntiles, nchunks, slices;
  if (wgrad_plan(deform ? 1 : 0, B, C, O, OH, OW, k, &ntiles, &nchunks, &slices) != DSU_OK)
    return 0;
  return (int64_t)slices * O * C * k * k * (int64_t)sizeof(float);
}

int dsu_conv2d_wgrad(const float* input, const float* dout, const void* tap_table, int32_t B,
                     int32_t C, int32_t H, int32_t W, int32_t O, int32_t k, int32_t stride,
                     int32_t pad, float* workspace, float* dweight, int32_t accumulate,
                     void* stream) {
  if (!input || !dout || !workspace || !dweight) return DSU_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || O > 128 || pad < 0 || stride <= 0)
    return DSU_EINVAL;
  if ((int64_t)B * C * H * W >= ((int64_t)1 << 31)) return DSU_EINVAL;
  const int mode = tap_table ? 1 : 0;
  if (mode == 1 && (k != 3 || stride != 1 || pad != 1)) return DSU_EUNSUP;
  if (C >= (1 << 24)) return DSU_EINVAL;
  WgradArgs a{};
  a.in = input; a.dout = dout; a.taps = (const TapRec*)tap_table; a.partial = workspace;
  a.B = B; a.C = C; a.H = H; a.W = W; a.O = O; a.stride = stride; a.pad = pad;
  a.OH = (H + 2 * pad - k) / stride + 1;
  a.OW = (W + 2 * pad - k) / stride + 1;
  if (a.OH <= 0 || a.OW <= 0) return DSU_EINVAL;
  a.ntot = C * k * k;
  int ntiles, slices;
  int rc = wgrad_plan(mode, B, C, O, a.OH, a.OW, k, &ntiles, &a.nchunks, &slices);
  if (rc != DSU_OK) return rc;
  a.chunks_per_slice = (a.nchunks + slices - 1) / slices;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1) rc = launch_wgrad<1, 3>(a, slices, s);
  else if (k == 1) rc = launch_wgrad<0, 1>(a, slices, s);
  else if (k == 3) rc = launch_wgrad<0, 3>(a, slices, s);
  else if (k == 4) rc = launch_wgrad<0, 4>(a, slices, s);
  else if (k == 7) rc = launch_wgrad<0, 7>(a, slices, s);
  else return DSU_EUNSUP;
  if (rc != DSU_OK) return rc;
  const int64_t count = (int64_t)O * a.ntot;
  wgrad_reduce_kernel<<<dsu_blocks_for(count, 256), 256, 0, s>>>(workspace, slices, count, dweight,
                                                                  accumulate);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_deform_conv3x3_dgrad_gather(const float* dcol, const int32_t* rowptr, const int32_t* src,
                                    const float* wgt, int64_t planes, int32_t npix, float* dx,
                                    void* stream) {
  if (!dcol || !rowptr || !src || !wgt || !dx || planes <= 0 || npix <= 0) return DSU_EINVAL;
  const int64_t groups = (planes + GATHER_PL - 1) / GATHER_PL;
  deform_gather_kernel<<<dsu_blocks_for(groups * npix, 256), 256, 0, (hipStream_t)stream>>>(
      dcol, rowptr, src, wgt, planes, npix, dx);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

static int norm_args_ok(const dsu_norm_cfg* c) {
  return c && c->batch > 0 && c->channels > 0 && c->hw > 0 && c->act >= 0 && c->act <= 2 &&
         (c->instance == 0 || c->instance == 1);
}

static void norm_fill(NormArgs* a, const dsu_norm_cfg* c) {
  a->inner = c->hw;
  a->channels = c->channels;
  a->group_stride = c->hw;
  if (c->instance) {
    a->groups = c->batch * c->channels;
    a->n_outer = 1;
    a->outer_stride = 0;
  } else {
    a->groups = c->channels;
    a->n_outer = c->batch;
    a->outer_stride = (int64_t)c->channels * c->hw;
  }
  a->eps = c->eps;
  a->momentum = c->momentum;
  a->stat_updates = c->stat_updates;
  a->act = c->act;
}

int dsu_norm_train_fwd(const dsu_norm_cfg* cfg, const float* x, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float* y,
                       float* save_mean, float* save_invstd, void* stream) {
  if (!norm_args_ok(cfg) || !x || !y || !save_mean || !save_invstd) return DSU_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return DSU_EINVAL;
  if (cfg->instance && running_mean) return DSU_EINVAL;
  NormArgs a{};
  norm_fill(&a, cfg);
  a.x = x; a.out = y; a.gamma = gamma; a.beta = beta; a.mean = save_mean; a.invstd = save_invstd;
  a.running_mean = running_mean; a.running_var = running_var;
  norm_fwd_kernel<<<a.groups, 256, 0, (hipStream_t)stream>>>(a);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_norm_train_bwd(const dsu_norm_cfg* cfg, const float* x, const float* y, const float* dy,
                       const float* gamma, const float* save_mean, const float* save_invstd,
                       float* dx, float* dgamma, float* dbeta, void* stream) {
  if (!norm_args_ok(cfg) || !x || !y || !dy || !dx || !save_mean || !save_invstd)
    return DSU_EINVAL;
  if (cfg->instance && (dgamma || dbeta)) return DSU_EINVAL;
  NormArgs a{};
  norm_fill(&a, cfg);
  a.x = x; a.y_in = y; a.dy = dy; a.out = dx; a.gamma = gamma;
  a.mean = const_cast<float*>(save_mean); a.invstd = const_cast<float*>(save_invstd);
  a.dgamma = dgamma; a.dbeta = dbeta;
  norm_bwd_kernel<<<a.groups, 256, 0, (hipStream_t)stream>>>(a);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_channel_sum(const float* x, int32_t B, int32_t C, int32_t HW, float* out, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || HW <= 0) return DSU_EINVAL;
  channel_sum_kernel<<<C, 256, 0, (hipStream_t)stream>>>(x, B, C, HW, out);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_act_fwd(const float* x, float* y, int64_t n, int32_t act, void* stream) {
  if (!x || !y || n <= 0 || act < 0 || act > 3) return DSU_EINVAL;
  act_fwd_kernel<<<dsu_capped_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(x, y, n, act);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int32_t act, void* stream) {
  if (!dy || !y || !dx || n <= 0 || act < 0 || act > 3) return DSU_EINVAL;
  act_bwd_kernel<<<dsu_capped_blocks(n, 256), 256, 0, (hipStream_t)stream>>>(dy, y, dx, n, act);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_maxpool2_fwd(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream) {
  if (!x || !y || planes <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return DSU_EINVAL;
  maxpool2_fwd_kernel<<<dsu_capped_blocks(planes * (H / 2) * (W / 2), 256), 256, 0,
                        (hipStream_t)stream>>>(x, y, planes, H, W);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_maxpool2_bwd(const float* x, const float* dy, float* dx, int64_t planes, int32_t H,
                     int32_t W, void* stream) {
  if (!x || !dy || !dx || planes <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return DSU_EINVAL;
  maxpool2_bwd_kernel<<<dsu_capped_blocks(planes * (H / 2) * (W / 2), 256), 256, 0,
                        (hipStream_t)stream>>>(x, dy, dx, planes, H, W);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_upsample2_fwd(const float* x, float* y, int64_t planes, int32_t H, int32_t W,
                      void* stream) {
  if (!x || !y || planes <= 0 || H <= 0 || W <= 0) return DSU_EINVAL;
  upsample2_fwd_kernel<<<dsu_capped_blocks(planes * 4 * H * W, 256), 256, 0,
                         (hipStream_t)stream>>>(x, y, planes, H, W);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_upsample2_bwd(const float* dy, float* dx, int64_t planes, int32_t H, int32_t W,
                      void* stream) {
  if (!dy || !dx || planes <= 0 || H <= 0 || W <= 0) return DSU_EINVAL;
  upsample2_bwd_kernel<<<dsu_capped_blocks(planes * H * W, 256), 256, 0, (hipStream_t)stream>>>(
      dy, dx, planes, H, W);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_pair_loss(const float* x, const float* target, float target_const, int64_t n,
                  int32_t kind, float grad_scale, float* grad, float* partial256, void* stream) {
  if (!x || !partial256 || n <= 0 || (kind != 0 && kind != 1)) return DSU_EINVAL;
  pair_loss_kernel<<<LOSS_BLOCKS, 256, 0, (hipStream_t)stream>>>(x, target, target_const, n, kind,
                                                                 grad_scale, grad, partial256);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
