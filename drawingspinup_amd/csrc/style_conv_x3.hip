// Evaluation-time convolution / fixed-offset deformable convolution of the style-translator nets
// on the gfx950 bf16 MFMA with "bf16 x 3" operands:
//     a b ~ a_hi b_hi + a_hi b_mid + a_mid b_hi,   x = x_hi + x_mid + O(2^-16 |x|)
// (f32 accumulation; relative error ~2^-15 per product, i.e. ~32x finer than the TF32 arithmetic
// PyTorch's cuDNN convolutions use by default on the reference's GPUs).  v_mfma_f32_32x32x16_bf16
// runs at 16x the rate of the f32 MFMA of style_conv.hip, so three of them per k-step still cut
// the matrix time by 16/3; what then bounds the kernel is forming the im2col values, which is why
// the B operand never goes through LDS here:
//
//   GEMM view  out[o][pix] = sum_{cb, tap, k<16} W[o][16 cb + k][tap] * col[16 cb + k][tap][pix]
//   k-step     = one tap x one block of 16 input channels.  The MFMA's B operand wants, in lane
//                (pixel = lane & 31, kh = lane >> 5), the 8 values k = 8 kh .. 8 kh + 7 of that
//                lane's pixel: eight channels at ONE tap, so the lane reuses one set of bilinear
//                offsets/weights (deform) or one validity test (plain) for all eight, computes them
//                in registers, splits them to bf16 hi/mid and feeds the MFMA directly.
//   A operand  = weights, packed once per weight version on the device to (Opad, CB, KK, 16) bf16
//                hi and mid arrays (dsu_conv_x3_pack_weights) so that a lane's 8 k-values are one
//                16-byte read; staged through LDS in groups of G k-steps, double buffered, one
//                barrier per group, shared by the four waves of the workgroup.
//   per wave   32 pixels x BN output channels (NT = BN/32 accumulator tiles), raw input values of
//                the NEXT k-step in flight while the current one is in the MFMAs.
//
// Same call surface as dsu_conv2d_fwd / dsu_deform_conv3x3_fwd (style_conv.hip) except for the
// packed weights; replaces, for evaluation, the same reference operators
// (3_style_translator/training/models.py:41-129 GeneratorJ, :302-351 the RIC generators'
// torchvision.ops.deform_conv2d call sites).  Training keeps the exact-f32 kernels.
#include "common.h"
#include "style_dev.h"

namespace {
using namespace dsu_style;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct XArgs {
  const float* in;
  const uint4* w_hi;         // (Opad, CB, KK, 16) bf16, two uint4 per (o, cb, tap)
  const uint4* w_mid;
  const uint4* w_f32;        // F32 kernels: (Opad, CB, KK, 16) f32, four uint4 per (o, cb, tap)
  const float* bias;
  const float* offset;       // deform only: (18,H,W)
  int64_t offset_bstride;
  const float* ep_scale;
  const float* ep_shift;
  const float* residual;
  float* out;
  int B, C, CB, H, W, O, Opad, OH, OW, pad;
  int act;
  int in_relu;
};

__device__ __forceinline__ void split8(const float* x, bf16x8& hi, bf16x8& mid) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    mid[i] = (__bf16)(x[i] - (float)h);
  }
}

// Bilinear state of one (pixel, tap), kept small because nine of them live in registers next to
// the accumulators: the first corner's offset, the two fractions, and six flag bits (held for all
// taps in two words): bit 0 c1 = c0 + 1, bit 1 r1 = r0 + W, bits 2..5 corner 00/01/10/11 inside.
// The four weights are re-formed per k-step with make_tap's own expressions (same bits).
struct XTap {
  int o00;
  float lh, lw;
};

__device__ __forceinline__ uint32_t tap_flags(float h, float w, int H, int W) {
  const bool inside = (h > -1.0f) && (w > -1.0f) && (h < (float)H) && (w < (float)W);
  const int h0 = (int)floorf(h), w0 = (int)floorf(w), h1 = h0 + 1, w1 = w0 + 1;
  const bool vh0 = inside && h0 >= 0, vh1 = inside && h1 <= H - 1;
  const bool vw0 = w0 >= 0, vw1 = w1 <= W - 1;
  const int dc = min(max(w1, 0), W - 1) - min(max(w0, 0), W - 1);
  const int dr = min(max(h1, 0), H - 1) - min(max(h0, 0), H - 1);
  return (uint32_t)dc | ((uint32_t)dr << 1) | ((vh0 && vw0) ? 4u : 0u) | ((vh0 && vw1) ? 8u : 0u) |
         ((vh1 && vw0) ? 16u : 0u) | ((vh1 && vw1) ? 32u : 0u);
}

// MODE 0: plain conv (KS x KS taps, stride STRIDE); MODE 1: 3x3 deformable, stride 1.
// G = k-steps (taps) per LDS weight group, G | KS*KS.
// F32 = false: bf16 x 3 products on v_mfma_f32_32x32x16_bf16 (one k-step = 3 MFMAs per tile).
// F32 = true : exact f32 products on v_mfma_f32_32x32x2_f32 — the arithmetic of the reference's
//   deform_conv2d (im2col + f32 addmm, TF32 off by default) and of an f32 ONNX / cuDNN-without-TF32
//   convolution.  Same k-step (one tap x 16 channels, the lane's 8 values straight from registers):
//   MFMA j of a k-step contracts channel 16 cb + j (lanes 0-31) with channel 16 cb + 8 + j (lanes
//   32-63), so its B operand is the lane's value j as it is and its A operand the lane's weight
//   W[o = 32 n + (lane & 31)][16 cb + 8 (lane >> 5) + j][tap] — 8 consecutive floats of the packed
//   row, read from LDS as two b128 (rows padded to 20 floats: conflict-free for the b128 lane
//   groups).  8 MFMAs of 64 clocks per tile and k-step; the im2col side (gathers, blend) is the
//   same ~130 VALU + 32 loads as in the bf16 form and hides behind them at two waves per SIMD.
template <int MODE, int KS, int STRIDE, int BN, int G, bool F32>
__global__ __launch_bounds__(256, 2) void conv_x3_kernel(XArgs a) {
  constexpr int KK = KS * KS;
  static_assert(KK % G == 0, "a weight group never straddles two channel blocks");
  constexpr int NG = KK / G;
  constexpr int NT = BN / 32;
  constexpr int NRAW = MODE == 1 ? 32 : 8;
  constexpr int AROW = 20;                                // F32: floats per LDS weight row (16 + 4)
  // uint4 per k-step: bf16 [hi|mid][o][kh], f32 [o][4]
  constexpr int A_ITEMS = F32 ? BN * 4 : 2 * BN * 2;
  constexpr int NQ = (A_ITEMS + 255) / 256;
  __shared__ uint4 sA[F32 ? 1 : 2][F32 ? 1 : G][2][F32 ? 1 : BN][2];   // bf16: [buf][k-step][hi|mid][o][kh]
  __shared__ __attribute__((aligned(16))) float sAf[F32 ? 2 : 1][F32 ? G : 1][F32 ? BN : 1][AROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int b = blockIdx.z;
  const int o_base = blockIdx.y * BN;
  const int npix = a.OH * a.OW;
  const int pix = blockIdx.x * 128 + wave * 32 + l31;
  const bool pv = pix < npix;
  const int oy = pv ? pix / a.OW : 0, ox = pv ? pix % a.OW : 0;
  const float* __restrict__ in_b = a.in + (size_t)b * a.C * a.H * a.W;
  const int plane = a.H * a.W;
  const float relu_lo = a.in_relu ? 0.0f : -INFINITY;
  const int relu_floor = a.in_relu ? 0 : (int)0x80000000;

  XTap taps[MODE == 1 ? 9 : 1];
  uint32_t tflags[2] = {0u, 0u};                          // 6 bits per tap: taps 0-4 | taps 5-8
  if (MODE == 1) {
    const float* off = a.offset + (size_t)b * a.offset_bstride;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float dh = 0.0f, dw = 0.0f;
      if (pv) {
        dh = off[(size_t)(2 * t) * npix + pix];
        dw = off[(size_t)(2 * t + 1) * npix + pix];
      }
      const float h = (float)(oy - a.pad + t / 3) + dh, w = (float)(ox - a.pad + t % 3) + dw;
      const Tap tp = make_tap(h, w, a.H, a.W);
      taps[t].o00 = tp.r0 + tp.c0;
      taps[t].lh = h - floorf(h);
      taps[t].lw = w - floorf(w);
      const uint32_t f = pv ? tap_flags(h, w, a.H, a.W) : 0u;   // no pixel: all corners "outside"
      tflags[t / 5] |= f << (6 * (t % 5));
    }
  }

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

  // ---- weights: global -> registers -> LDS, one k-step's worth at a time (few live registers);
  // the whole group of G k-steps is in the other buffer by the barrier that ends the group
  const size_t w_row = (size_t)a.CB * KK * (F32 ? 4 : 2);   // uint4 per output channel
  auto load_A = [&](uint4 (&ra)[NQ], int cb, int t) {
    const size_t s0 = ((size_t)cb * KK + t) * (F32 ? 4 : 2);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int id = tid + 256 * q;
      // rows past Opad (a tile wider than the layer) re-read the last row: their accumulators are
      // never stored; threads past A_ITEMS (BN = 32) repeat the first items
      if (F32) {
        const int k4 = id & 3, o = (id >> 2) % BN;
        ra[q] = a.w_f32[(size_t)min(o_base + o, a.Opad - 1) * w_row + s0 + k4];
      } else {
        const int k2 = id & 1, o = (id >> 1) % BN, part = (id >> 1) / BN;
        const uint4* w = (part & 1) ? a.w_mid : a.w_hi;
        ra[q] = w[(size_t)min(o_base + o, a.Opad - 1) * w_row + s0 + k2];
      }
    }
  };
  auto store_A = [&](const uint4 (&ra)[NQ], int buf, int i) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int id = tid + 256 * q;
      if (F32) {
        const int k4 = id & 3, o = (id >> 2) % BN;
        *reinterpret_cast<uint4*>(&sAf[buf][i][o][4 * k4]) = ra[q];
      } else {
        const int k2 = id & 1, o = (id >> 1) % BN, part = (id >> 1) / BN;
        sA[buf][i][part & 1][o][k2] = ra[q];
      }
    }
  };

  // ---- raw input values of one k-step: channels 16 cb + 8 kh + j at tap t.  Channels past C are
  // read from the last plane (finite values) and meet zero weights.
  // Buffer loads: descriptor of this image's (C,H,W) block in SGPRs, the channel 16 cb + j as the
  // uniform soffset, ONE 32-bit lane byte offset per corner in voffset (the lane's 8 kh planes +
  // the corner) — a k-step spends 4 vector adds on addressing.  C % 8 == 0 (host-checked); when
  // C % 16 == 8 the last block's kh = 1 lanes have no channels: they re-read the kh = 0 planes
  // (valid memory) and meet zero weights.
  float rv[NRAW];
  bool rok = false;
  const uint32_t plane_b = (uint32_t)plane * 4u;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in_b), 0, (int)((uint32_t)a.C * plane_b), 0x00020000);
  const uint32_t kh_full = (uint32_t)kh * 8u * plane_b;
  uint32_t kh_b = 8 >= a.C ? 0u : kh_full;
  int lb = 0;                                             // channel block of the loads in flight
  auto ld = [&](int j, uint32_t voff) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
        rsrc, (int)voff, (int)((uint32_t)(lb * 16 + j) * plane_b), 0));
  };
  auto load_raw = [&](int ty, int tx, int t) {
    if (MODE == 1) {
      // (opaque to the optimiser: hoisted out of the channel-block loop these per-tap values
      // would occupy the registers the compact form exists to free)
      uint32_t f = tflags[t / 5];
      uint32_t o00 = (uint32_t)taps[t].o00;
      asm volatile("" : "+v"(f), "+v"(o00));
      f >>= 6 * (t % 5);
      o00 = o00 * 4u + kh_b;
      const uint32_t o01 = o00 + ((f & 1u) << 2);
      const uint32_t o10 = o00 + ((f & 2u) ? (uint32_t)a.W * 4u : 0u);
      const uint32_t o11 = o10 + ((f & 1u) << 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        rv[4 * j + 0] = ld(j, o00);
        rv[4 * j + 1] = ld(j, o01);
        rv[4 * j + 2] = ld(j, o10);
        rv[4 * j + 3] = ld(j, o11);
      }
    } else {
      const int iy = oy * STRIDE - a.pad + ty, ix = ox * STRIDE - a.pad + tx;
      rok = pv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const uint32_t o = (rok ? (uint32_t)(iy * a.W + ix) * 4u : 0u) + kh_b;
#pragma unroll
      for (int j = 0; j < 8; ++j) rv[j] = ld(j, o);
    }
  };
  auto make_B = [&](int t, float (&v)[8]) {
    if (MODE == 1) {
      uint32_t f = tflags[t / 5];
      float lh = taps[t].lh, lw = taps[t].lw;
      asm volatile("" : "+v"(f), "+v"(lh), "+v"(lw));
      f >>= 6 * (t % 5);
      const float hh = 1.0f - lh, hw = 1.0f - lw;
      const float w00 = (f & 4u) ? hh * hw : 0.0f, w01 = (f & 8u) ? hh * lw : 0.0f;
      const float w10 = (f & 16u) ? lh * hw : 0.0f, w11 = (f & 32u) ? lh * lw : 0.0f;
      // ReLU on the raw corner values as ONE integer max each (negative floats are negative
      // ints; relu_floor = INT_MIN leaves the value alone) — fmaxf costs a canonicalise + a max
#pragma unroll
      for (int q = 0; q < 32; ++q)
        rv[q] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, rv[q]), relu_floor));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = w00 * rv[4 * j] + w01 * rv[4 * j + 1] + w10 * rv[4 * j + 2] + w11 * rv[4 * j + 3];
    } else {
      // one v_med3 per value: padding taps clamp to [0, 0], the others to [relu_lo, +inf)
      const float lo = rok ? relu_lo : 0.0f, hi = rok ? INFINITY : 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_fmed3f(rv[j], lo, hi);
    }
  };

  // one k-step: this tap's B operand from the raw values in flight, the next k-step's raw loads
  // and weight piece issued, NT x 3 MFMAs, the weight piece parked in the other buffer
  auto k_step = [&](int cb, int tg, int i, int ty, int tx, int nty, int ntx, int buf) {
    const int t = tg * G + i;
    const bool last_t = t + 1 == KK;
    bf16x8 bh, bm;
    float v[8];
    uint4 ra[NQ];
    make_B(t, v);
    if (!F32) split8(v, bh, bm);
    if (last_t) {
      lb = min(cb + 1, a.CB - 1);
      kh_b = lb * 16 + 8 >= a.C ? 0u : kh_full;
    }
    load_raw(nty, ntx, last_t ? 0 : t + 1);
    // (unconditional: after the last group this re-reads valid weights into the buffer nobody
    // reads any more — a branch here made the compiler park the piece in scratch)
    load_A(ra, tg + 1 < NG ? cb : min(cb + 1, a.CB - 1), (tg + 1 < NG ? (tg + 1) * G : 0) + i);
    if (F32) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float* ap = &sAf[buf][i][n * 32 + l31][8 * kh];
        const float4 a0 = *reinterpret_cast<const float4*>(ap);
        const float4 a1 = *reinterpret_cast<const float4*>(ap + 4);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, v[0], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, v[1], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, v[2], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, v[3], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, v[4], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, v[5], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, v[6], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, v[7], acc[n], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, sA[buf][i][0][n * 32 + l31][kh]);
        const bf16x8 am = __builtin_bit_cast(bf16x8, sA[buf][i][1][n * 32 + l31][kh]);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[n], 0, 0, 0);
      }
    }
    store_A(ra, buf ^ 1, i);
  };

#pragma unroll
  for (int i = 0; i < G; ++i) {
    uint4 ra[NQ];
    load_A(ra, 0, i);
    store_A(ra, 0, i);
  }
  load_raw(0, 0, 0);
  __syncthreads();
  int buf = 0;
  for (int cb = 0; cb < a.CB; ++cb) {
    if (MODE == 1) {
      // tap index must be a compile-time constant (register arrays): all nine k-steps unrolled
#pragma unroll
      for (int tg = 0; tg < NG; ++tg) {
#pragma unroll
        for (int i = 0; i < G; ++i) k_step(cb, tg, i, 0, 0, 0, 0, buf);
        __syncthreads();
        buf ^= 1;
      }
    } else {
      // G = KS: a group is one kernel row; rows stay a runtime loop (the 7x7 layers' 49 k-steps
      // unrolled would not fit the instruction cache)
      static_assert(MODE == 1 || G == KS, "plain conv: one kernel row per weight group");
#pragma unroll 1
      for (int tg = 0; tg < NG; ++tg) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const bool wrap = i + 1 == G;
          k_step(cb, tg, i, tg, i, wrap ? (tg + 1 < NG ? tg + 1 : 0) : tg, wrap ? 0 : i + 1, buf);
        }
        __syncthreads();
        buf ^= 1;
      }
    }
  }

  // ---- epilogue: lane -> pixel (lane&31), register r -> channel row
  if (pv) {
    float* out_b = a.out + (size_t)b * a.O * npix;
    const float* res_b = a.residual ? a.residual + (size_t)b * a.O * npix : nullptr;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o_base + n * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
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

__global__ void pack_weights_kernel(const float* __restrict__ w, int O, int C, int KK, int CB,
                                    int64_t total, uint16_t* __restrict__ hi,
                                    uint16_t* __restrict__ mid) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx & 15);
  int64_t r = idx >> 4;
  const int t = (int)(r % KK); r /= KK;
  const int cb = (int)(r % CB);
  const int o = (int)(r / CB);
  const int c = cb * 16 + k;
  const float v = (o < O && c < C) ? w[((size_t)o * C + c) * KK + t] : 0.0f;
  const __bf16 h = (__bf16)v;
  const __bf16 m = (__bf16)(v - (float)h);
  hi[idx] = __builtin_bit_cast(uint16_t, h);
  mid[idx] = __builtin_bit_cast(uint16_t, m);
}

// f32 form of the packed layout (the F32 kernels' A operand): (Opad, CB, KK, 16) floats, zero padded
__global__ void pack_weights_f32_kernel(const float* __restrict__ w, int O, int C, int KK, int CB,
                                        int64_t total, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx & 15);
  int64_t r = idx >> 4;
  const int t = (int)(r % KK); r /= KK;
  const int cb = (int)(r % CB);
  const int o = (int)(r / CB);
  const int c = cb * 16 + k;
  out[idx] = (o < O && c < C) ? w[((size_t)o * C + c) * KK + t] : 0.0f;
}

inline int opad_of(int O) { return (O + 31) & ~31; }
inline int cb_of(int C) { return (C + 15) / 16; }

template <int MODE, int KS, int STRIDE, int G, bool F32 = false>
int launch_x3(const XArgs& a, hipStream_t s) {
  const int npix = a.OH * a.OW;
  const int gx = (npix + 127) / 128;
  // as in style_conv.hip: small images get 32-channel tiles so that the launch covers the chip
  if (a.Opad > 32 && (int64_t)gx * ((a.Opad + 127) / 128) * a.B < 256) {
    dim3 grid(gx, a.Opad / 32, a.B);
    conv_x3_kernel<MODE, KS, STRIDE, 32, G, F32><<<grid, 256, 0, s>>>(a);
  } else if (a.Opad > 64) {
    dim3 grid(gx, (a.Opad + 127) / 128, a.B);
    conv_x3_kernel<MODE, KS, STRIDE, 128, G, F32><<<grid, 256, 0, s>>>(a);
  } else if (a.Opad > 32) {
    dim3 grid(gx, 1, a.B);
    conv_x3_kernel<MODE, KS, STRIDE, 64, G, F32><<<grid, 256, 0, s>>>(a);
  } else {
    dim3 grid(gx, 1, a.B);
    conv_x3_kernel<MODE, KS, STRIDE, 32, G, F32><<<grid, 256, 0, s>>>(a);
  }
  if (hipGetLastError() != hipSuccess) return DSU_ELAUNCH;
  return DSU_OK;
}

}  // namespace

extern "C" {

int64_t dsu_conv_x3_packed_elems(int32_t O, int32_t C, int32_t k) {
  if (O <= 0 || C <= 0 || k <= 0) return 0;
  return (int64_t)opad_of(O) * cb_of(C) * k * k * 16;
}

int dsu_conv_x3_pack_weights(const float* weight, int32_t O, int32_t C, int32_t k, uint16_t* w_hi,
                             uint16_t* w_mid, void* stream) {
  if (!weight || !w_hi || !w_mid || O <= 0 || C <= 0 || k <= 0) return DSU_EINVAL;
  if ((((uintptr_t)w_hi) | ((uintptr_t)w_mid)) & 15) return DSU_EINVAL;
  const int64_t total = dsu_conv_x3_packed_elems(O, C, k);
  pack_weights_kernel<<<dsu_blocks_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      weight, O, C, k * k, cb_of(C), total, w_hi, w_mid);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_deform_conv3x3_fwd_x3(const float* input, const float* offset, int64_t offset_batch_stride,
                              const uint16_t* w_hi, const uint16_t* w_mid, int32_t B, int32_t C,
                              int32_t H, int32_t W, int32_t O, int32_t in_relu,
                              const float* ep_scale, const float* ep_shift, int32_t act,
                              const float* residual, float* out, void* stream) {
  if (!input || !offset || !w_hi || !w_mid || !out) return DSU_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || act < 0 || act > 3) return DSU_EINVAL;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return DSU_EINVAL;
  if ((int64_t)C * H * W >= (1ll << 30) || (C & 7)) return DSU_EUNSUP;
  XArgs a{};
  a.in = input; a.w_hi = (const uint4*)w_hi; a.w_mid = (const uint4*)w_mid; a.bias = nullptr;
  a.offset = offset; a.offset_bstride = offset_batch_stride;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.residual = residual; a.out = out;
  a.B = B; a.C = C; a.CB = cb_of(C); a.H = H; a.W = W; a.O = O; a.Opad = opad_of(O);
  a.OH = H; a.OW = W; a.pad = 1; a.act = act; a.in_relu = in_relu;
  return launch_x3<1, 3, 1, 3>(a, (hipStream_t)stream);
}

int dsu_conv_f32p_pack_weights(const float* weight, int32_t O, int32_t C, int32_t k, float* w_packed,
                               void* stream) {
  if (!weight || !w_packed || O <= 0 || C <= 0 || k <= 0) return DSU_EINVAL;
  if (((uintptr_t)w_packed) & 15) return DSU_EINVAL;
  const int64_t total = dsu_conv_x3_packed_elems(O, C, k);
  pack_weights_f32_kernel<<<dsu_blocks_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      weight, O, C, k * k, cb_of(C), total, w_packed);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_deform_conv3x3_fwd_f32p(const float* input, const float* offset, int64_t offset_batch_stride,
                                const float* w_packed, int32_t B, int32_t C, int32_t H, int32_t W,
                                int32_t O, int32_t in_relu, const float* ep_scale,
                                const float* ep_shift, int32_t act, const float* residual, float* out,
                                void* stream) {
  if (!input || !offset || !w_packed || !out) return DSU_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || act < 0 || act > 3) return DSU_EINVAL;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return DSU_EINVAL;
  if ((int64_t)C * H * W >= (1ll << 30) || (C & 7)) return DSU_EUNSUP;
  XArgs a{};
  a.in = input; a.w_f32 = (const uint4*)w_packed; a.bias = nullptr;
  a.offset = offset; a.offset_bstride = offset_batch_stride;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.residual = residual; a.out = out;
  a.B = B; a.C = C; a.CB = cb_of(C); a.H = H; a.W = W; a.O = O; a.Opad = opad_of(O);
  a.OH = H; a.OW = W; a.pad = 1; a.act = act; a.in_relu = in_relu;
  return launch_x3<1, 3, 1, 3, true>(a, (hipStream_t)stream);
}

int dsu_conv2d_fwd_f32p(const float* input, const float* w_packed, const float* bias, int32_t B,
                        int32_t C, int32_t H, int32_t W, int32_t O, int32_t k, int32_t stride,
                        int32_t pad, int32_t in_relu, const float* ep_scale, const float* ep_shift,
                        int32_t act, const float* residual, float* out, void* stream) {
  if (!input || !w_packed || !out) return DSU_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || act < 0 || act > 3 || pad < 0)
    return DSU_EINVAL;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return DSU_EINVAL;
  if ((int64_t)C * H * W >= (1ll << 30) || (C & 7)) return DSU_EUNSUP;
  XArgs a{};
  a.in = input; a.w_f32 = (const uint4*)w_packed; a.bias = bias;
  a.offset = nullptr; a.offset_bstride = 0;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.residual = residual; a.out = out;
  a.B = B; a.C = C; a.CB = cb_of(C); a.H = H; a.W = W; a.O = O; a.Opad = opad_of(O);
  a.pad = pad; a.act = act; a.in_relu = in_relu;
  a.OH = (H + 2 * pad - k) / stride + 1;
  a.OW = (W + 2 * pad - k) / stride + 1;
  if (a.OH <= 0 || a.OW <= 0) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (k == 1 && stride == 1) return launch_x3<0, 1, 1, 1, true>(a, s);
  if (k == 3 && stride == 1) return launch_x3<0, 3, 1, 3, true>(a, s);
  if (k == 3 && stride == 2) return launch_x3<0, 3, 2, 3, true>(a, s);
  return DSU_EUNSUP;             // 7x7: the LDS weight group of a kernel row does not fit in f32
}

int dsu_conv2d_fwd_x3(const float* input, const uint16_t* w_hi, const uint16_t* w_mid,
                      const float* bias, int32_t B, int32_t C, int32_t H, int32_t W, int32_t O,
                      int32_t k, int32_t stride, int32_t pad, int32_t in_relu,
                      const float* ep_scale, const float* ep_shift, int32_t act,
                      const float* residual, float* out, void* stream) {
  if (!input || !w_hi || !w_mid || !out) return DSU_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || act < 0 || act > 3 || pad < 0)
    return DSU_EINVAL;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return DSU_EINVAL;
  if ((int64_t)C * H * W >= (1ll << 30) || (C & 7)) return DSU_EUNSUP;
  XArgs a{};
  a.in = input; a.w_hi = (const uint4*)w_hi; a.w_mid = (const uint4*)w_mid; a.bias = bias;
  a.offset = nullptr; a.offset_bstride = 0;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.residual = residual; a.out = out;
  a.B = B; a.C = C; a.CB = cb_of(C); a.H = H; a.W = W; a.O = O; a.Opad = opad_of(O);
  a.pad = pad; a.act = act; a.in_relu = in_relu;
  a.OH = (H + 2 * pad - k) / stride + 1;
  a.OW = (W + 2 * pad - k) / stride + 1;
  if (a.OH <= 0 || a.OW <= 0) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (k == 1 && stride == 1) return launch_x3<0, 1, 1, 1>(a, s);
  if (k == 3 && stride == 1) return launch_x3<0, 3, 1, 3>(a, s);
  if (k == 3 && stride == 2) return launch_x3<0, 3, 2, 3>(a, s);
  if (k == 7 && stride == 1) return launch_x3<0, 7, 1, 7>(a, s);
  return DSU_EUNSUP;
}

}  // extern "C"
