// f32 implicit-GEMM convolution and fixed-offset deformable convolution on the gfx950
// f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate).
//
// Replaces, for the per-frame style-translator nets
// (3_style_translator/training/models.py):
//   * torchvision.ops.deform_conv2d (torchvision 0.15.1, un-vendored; call sites
//     models.py:302,308,314,322,325,331,338,344,348,351) — bilinear im2col + GEMM,
//   * nn.Conv2d / BatchNorm2d(eval) / LeakyReLU / ReLU / Tanh of GeneratorJ
//     (models.py:41-129) — the cuDNN path in the reference.
//
// GEMM view:  out[o][pix] = sum_{c,tap} W[o][c][tap] * col[c][tap][pix]
//   A operand (M = output channels) = weights, B operand (N = pixels) = im2col values, so
//   that the 32x32 accumulator tile has a PIXEL per lane and stores are row-contiguous.
// Per workgroup (256 threads = 4 waves): 128 consecutive output pixels x BN channels,
// K walked in chunks of KC channels x KK taps staged through LDS (double buffered).
#include "common.h"
#include "style_dev.h"

namespace {
using namespace dsu_style;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;  // pixels per workgroup

struct ConvArgs {
  const float* in;
  const float* w;
  const float* bias;
  const float* offset;       // deform only: (18,H,W)
  int64_t offset_bstride;    // elements between images' offset maps (0 = shared)
  const float* ep_scale;
  const float* ep_shift;
  const float* residual;
  float* out;
  int B, C, H, W, O, OH, OW, pad;
  int act;
  int in_relu;   // apply ReLU to the input values as they are read (resnet pre-activation)
};

__device__ __forceinline__ float sample_tap(const float* __restrict__ plane, const Tap& t,
                                            bool in_relu) {
  float v00 = plane[t.r0 + t.c0], v01 = plane[t.r0 + t.c1];
  float v10 = plane[t.r1 + t.c0], v11 = plane[t.r1 + t.c1];
  if (in_relu) {
    v00 = fmaxf(v00, 0.0f); v01 = fmaxf(v01, 0.0f);
    v10 = fmaxf(v10, 0.0f); v11 = fmaxf(v11, 0.0f);
  }
  return t.w00 * v00 + t.w01 * v01 + t.w10 * v10 + t.w11 * v11;
}

// MODE 0: plain conv (KS x KS taps, stride STRIDE); MODE 1: 3x3 deformable, stride 1.
template <int MODE, int KS, int STRIDE, int KC, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int KK = KS * KS;
  constexpr int KE = KC * KK;            // real K elements per chunk
  constexpr int KP = (KE + 1) & ~1;      // padded to the MFMA's k=2
  constexpr int NT = BN / 32;            // 32-channel MFMA row blocks per wave
  __shared__ __attribute__((aligned(16))) float sB[2][KP][BM];   // im2col values
  __shared__ __attribute__((aligned(16))) float sA[2][KP][BN];   // weights, k-major

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z;
  const int o_base = blockIdx.y * BN;
  const int npix = a.OH * a.OW;
  const int pix0 = blockIdx.x * BM;

  // ---- staging roles
  const int sp = tid & (BM - 1);   // pixel this thread stages
  const int sg = tid >> 7;         // which half of the chunk's K elements (0/1)
  const int spix = pix0 + sp;
  const bool spv = spix < npix;
  const int oy = spv ? spix / a.OW : 0, ox = spv ? spix % a.OW : 0;
  const float* in_b = a.in + (size_t)b * a.C * a.H * a.W;
  const size_t plane = (size_t)a.H * a.W;

  Tap taps[MODE == 1 ? 9 : 1];
  if (MODE == 1) {
    const float* off = a.offset + (size_t)b * a.offset_bstride;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float dh = 0.0f, dw = 0.0f;
      if (spv) {
        dh = off[(size_t)(2 * t) * npix + spix];
        dw = off[(size_t)(2 * t + 1) * npix + spix];
      }
      const float h = (float)(oy - a.pad + t / 3) + dh;
      const float w = (float)(ox - a.pad + t % 3) + dw;
      taps[t] = make_tap(h, w, a.H, a.W);
    }
  }

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

  const int nchunks = (a.C + KC - 1) / KC;

  // Staging is split in two so that global loads overlap the MFMAs of the current chunk: the
  // values of chunk ch+1 are LOADED into registers before the MFMA loop of chunk ch and only
  // CONSUMED (bilinear blend, ReLU, LDS store) after it.  The first version loaded and stored in
  // one step before the MFMAs: every chunk waited for its own gathers (PMC: 54 % of wave cycles
  // in s_waitcnt/barrier waits, 12-20 % MFMA busy).
  constexpr int EH = (KP + 1) / 2;                       // MODE 0: K elements per staging half
  constexpr int CH = KC / 2 > 0 ? KC / 2 : 1;            // MODE 1: channels per staging half
  constexpr int NV = MODE == 1 ? CH * 9 * 4 : EH;        // raw values held per thread
  constexpr int NW = (KP * BN + 255) / 256;              // weight values held per thread
  float rv[NV], rw[NW];
  bool cvs[MODE == 1 ? CH : 1];

  // MODE 0: which tap / channel-in-chunk each of this thread's EH staged elements is, its
  // offset from the chunk's first channel plane and whether it falls inside the image — all
  // independent of the chunk, so computed once (the 7x7 layer spent ~9 VALU instructions per
  // MFMA re-deriving them for every one of its 166 chunks)
  int tap_rel[MODE == 0 ? EH : 1], tap_cc[MODE == 0 ? EH : 1];
  uint32_t tap_ok = 0;
  static_assert(EH <= 32, "validity bits are kept in one 32-bit mask");
  if (MODE == 0) {
#pragma unroll
    for (int q = 0; q < EH; ++q) {
      const int e = sg * EH + q;
      const int cc = e / KK, t = e % KK;
      const int iy = oy * STRIDE - a.pad + t / KS, ix = ox * STRIDE - a.pad + t % KS;
      const bool ok = e < KE && spv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      tap_cc[q] = cc;
      tap_rel[q] = ok ? cc * (int)plane + iy * a.W + ix : 0;
      tap_ok |= ok ? (1u << q) : 0u;
    }
  }

  auto stage_load = [&](int chunk) {
    const int c0 = chunk * KC;
    if (MODE == 1) {
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        const int c = c0 + sg * CH + cc;
        const bool cv = spv && c < a.C;
        cvs[cc] = cv;
        const float* pl = in_b + (size_t)(cv ? c : 0) * plane;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const Tap& tp = taps[t];
          rv[(cc * 9 + t) * 4 + 0] = pl[tp.r0 + tp.c0];
          rv[(cc * 9 + t) * 4 + 1] = pl[tp.r0 + tp.c1];
          rv[(cc * 9 + t) * 4 + 2] = pl[tp.r1 + tp.c0];
          rv[(cc * 9 + t) * 4 + 3] = pl[tp.r1 + tp.c1];
        }
      }
    } else {
      const float* cb = in_b + (size_t)c0 * plane;
#pragma unroll
      for (int q = 0; q < EH; ++q) {
        float v = 0.0f;
        if (((tap_ok >> q) & 1u) && c0 + tap_cc[q] < a.C) v = cb[tap_rel[q]];
        rv[q] = v;
      }
    }
    // --- weights: W[o][c0*KK .. c0*KK+KE) is contiguous per o
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int idx = tid + 256 * q;
      const int o = idx % BN, e = idx / BN;
      float v = 0.0f;
      const int c = c0 + e / KK;
      if (idx < KP * BN && e < KE && c < a.C && o_base + o < a.O)
        v = a.w[((size_t)(o_base + o) * a.C + c0) * KK + e];
      rw[q] = v;
    }
  };
  auto stage_store = [&](int buf) {
    if (MODE == 1) {
#pragma unroll
      for (int cc = 0; cc < CH; ++cc)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const Tap& tp = taps[t];
          float v00 = rv[(cc * 9 + t) * 4 + 0], v01 = rv[(cc * 9 + t) * 4 + 1];
          float v10 = rv[(cc * 9 + t) * 4 + 2], v11 = rv[(cc * 9 + t) * 4 + 3];
          if (a.in_relu) {
            v00 = fmaxf(v00, 0.0f); v01 = fmaxf(v01, 0.0f);
            v10 = fmaxf(v10, 0.0f); v11 = fmaxf(v11, 0.0f);
          }
          const float v = tp.w00 * v00 + tp.w01 * v01 + tp.w10 * v10 + tp.w11 * v11;
          sB[buf][(sg * CH + cc) * 9 + t][sp] = cvs[cc] ? v : 0.0f;
        }
    } else {
#pragma unroll
      for (int q = 0; q < EH; ++q) {
        const int e = sg * EH + q;
        if (e < KP) sB[buf][e][sp] = a.in_relu ? fmaxf(rv[q], 0.0f) : rv[q];
      }
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int idx = tid + 256 * q;
      if (idx < KP * BN) sA[buf][idx / BN][idx % BN] = rw[q];
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) stage_load(ch + 1);
    // wave `wave` owns pixels [32*wave, 32*wave+32) x all BN channels
    const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kp = 0; kp < KP / 2; ++kp) {
      const float bv = sB[buf][2 * kp + kh][wave * 32 + l31];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float av = sA[buf][2 * kp + kh][n * 32 + l31];
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
      }
    }
    if (ch + 1 < nchunks) stage_store(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane -> pixel (lane&31), register r -> channel row
  const int pix = pix0 + wave * 32 + (lane & 31);
  if (pix < npix) {
    float* out_b = a.out + (size_t)b * a.O * npix;
    const float* res_b = a.residual ? a.residual + (size_t)b * a.O * npix : nullptr;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o_base + n * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (o < a.O) {
          float v = acc[n][r];
          if (a.bias) v += a.bias[o];
          if (a.ep_scale) v = v * a.ep_scale[o] + a.ep_shift[o];
          v = apply_act(v, a.act);
          if (res_b) v += res_b[(size_t)o * npix + pix];
          out_b[(size_t)o * npix + pix] = v;
        }
      }
    }
  }
}

// generate_coordinates (models.py:551-604): theta = atan2(col - cx, row - cy) mod 2pi,
// rounded to 1e-4; tap k samples at p + (cos, sin)(theta + m_k*pi/4); centre tap offset 0.
__global__ void ric_offsets_kernel(int H, int W, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int row = idx / W, col = idx % W;
  const float cy = (float)H / 2.0f - 0.5f, cx = (float)W / 2.0f - 0.5f;
  const float two_pi = 3.14159274101257324f * 2.0f;  // torch.Tensor([math.pi]) * 2.0 in f32
  const float dx = (float)row - cy, dy = (float)col - cx;
  float th = atan2f(dy, dx);
  // python-style modulo for tensors: result takes the sign of the divisor
  th = fmodf(th, two_pi);
  if (th < 0.0f) th += two_pi;
  th = rintf(10000.0f * th) / 10000.0f;
  const float step = two_pi / 8.0f;
  const float bi[9] = {1.f, 1.f, 1.f, 0.f, 0.f, 0.f, -1.f, -1.f, -1.f};
  const float bj[9] = {1.f, 0.f, -1.f, 1.f, 0.f, -1.f, 1.f, 0.f, -1.f};
  const size_t n = (size_t)H * W;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float oh = 0.0f, ow = 0.0f;
    if (k != 4) {
      const float m = (float)(k < 4 ? k : k - 1);
      const float ang = th + step * m;
      oh = cosf(ang) + bi[k];
      ow = sinf(ang) + bj[k];
    }
    out[(size_t)(2 * k) * n + idx] = oh;
    out[(size_t)(2 * k + 1) * n + idx] = ow;
  }
}

template <int MODE, int KS, int STRIDE, int KC>
int launch_conv(const ConvArgs& a, hipStream_t s) {
  const int npix = a.OH * a.OW;
  const int gx = (npix + BM - 1) / BM;
  // Small images (the 8x8 / 16x16 levels of the 32x32 training patches): with 128-channel
  // tiles the launch has fewer workgroups than the chip has CUs.  32-channel tiles give 4x
  // the workgroups; each repeats the im2col staging, but on CUs that would otherwise idle.
  // Every output element accumulates in the same order whatever the tile width.
  if (a.O > 32 && (int64_t)gx * ((a.O + 127) / 128) * a.B < 256) {
    dim3 grid(gx, (a.O + 31) / 32, a.B);
    conv_igemm_kernel<MODE, KS, STRIDE, KC, 32><<<grid, 256, 0, s>>>(a);
  } else if (a.O > 64) {
    dim3 grid(gx, (a.O + 127) / 128, a.B);
    conv_igemm_kernel<MODE, KS, STRIDE, KC, 128><<<grid, 256, 0, s>>>(a);
  } else if (a.O > 32) {
    dim3 grid(gx, 1, a.B);
    conv_igemm_kernel<MODE, KS, STRIDE, KC, 64><<<grid, 256, 0, s>>>(a);
  } else {
    dim3 grid(gx, 1, a.B);
    conv_igemm_kernel<MODE, KS, STRIDE, KC, 32><<<grid, 256, 0, s>>>(a);
  }
  if (hipGetLastError() != hipSuccess) return DSU_ELAUNCH;
  return DSU_OK;
}

}  // namespace

extern "C" {

int dsu_ric_offsets(int32_t H, int32_t W, float* offsets, void* stream) {
  if (H <= 0 || W <= 0 || !offsets) return DSU_EINVAL;
  ric_offsets_kernel<<<dsu_blocks_for((int64_t)H * W, 256), 256, 0, (hipStream_t)stream>>>(
      H, W, offsets);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_deform_conv3x3_fwd(const float* input, const float* offset, int64_t offset_batch_stride,
                           const float* weight, int32_t B, int32_t C, int32_t H, int32_t W,
                           int32_t O, int32_t in_relu, const float* ep_scale,
                           const float* ep_shift, int32_t act, const float* residual, float* out,
                           void* stream) {
  if (!input || !offset || !weight || !out) return DSU_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || act < 0 || act > 3) return DSU_EINVAL;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return DSU_EINVAL;
  ConvArgs a{};
  a.in = input; a.w = weight; a.bias = nullptr; a.offset = offset;
  a.offset_bstride = offset_batch_stride;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.residual = residual; a.out = out;
  a.B = B; a.C = C; a.H = H; a.W = W; a.O = O; a.OH = H; a.OW = W; a.pad = 1; a.act = act; a.in_relu = in_relu;
  return launch_conv<1, 3, 1, 4>(a, (hipStream_t)stream);
}

int dsu_conv2d_fwd(const float* input, const float* weight, const float* bias, int32_t B,
                   int32_t C, int32_t H, int32_t W, int32_t O, int32_t k, int32_t stride,
                   int32_t pad, int32_t in_relu, const float* ep_scale, const float* ep_shift,
                   int32_t act, const float* residual, float* out, void* stream) {
  if (!input || !weight || !out) return DSU_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || act < 0 || act > 3 || pad < 0)
    return DSU_EINVAL;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return DSU_EINVAL;
  ConvArgs a{};
  a.in = input; a.w = weight; a.bias = bias; a.offset = nullptr; a.offset_bstride = 0;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.residual = residual; a.out = out;
  a.B = B; a.C = C; a.H = H; a.W = W; a.O = O; a.pad = pad; a.act = act; a.in_relu = in_relu;
  a.OH = (H + 2 * pad - k) / stride + 1;
  a.OW = (W + 2 * pad - k) / stride + 1;
  if (a.OH <= 0 || a.OW <= 0) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (k == 1 && stride == 1) return launch_conv<0, 1, 1, 32>(a, s);
  if (k == 3 && stride == 1) return launch_conv<0, 3, 1, 4>(a, s);
  if (k == 3 && stride == 2) return launch_conv<0, 3, 2, 4>(a, s);
  if (k == 7 && stride == 1) return launch_conv<0, 7, 1, 1>(a, s);
  if (k == 4 && stride == 1) return launch_conv<0, 4, 1, 2>(a, s);   // DiscriminatorN_IN
  if (k == 4 && stride == 2) return launch_conv<0, 4, 2, 2>(a, s);
  return DSU_EUNSUP;
}

}  // extern "C"
