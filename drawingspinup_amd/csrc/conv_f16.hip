// f16 implicit-GEMM convolution (NHWC activations, f32 accumulate) on gfx950 MFMA
// (v_mfma_f32_32x32x16_f16) for the diffusion UNet / VAE conv blocks.
//
// Replaces the cuDNN convolutions the reference reaches through diffusers' ResnetBlock2D /
// Downsample2D / Upsample2D / conv_in / conv_out
// (2_charactor_reconstructor/mvdiffusion/models/unet_mv2d_blocks.py:528,649,688,798,839;
//  unet_mv2d_condition.py:290,623) — K3 of SURVEY.md §2.3.
//
// GEMM view (K ordered tap-major, channel-minor so that NHWC rows are contiguous in K):
//   out[pix][o] = sum_{tap,c} W[o][tap][c] * in[pix @ tap][c]
// A operand (M) = weights [O][K] (pre-arranged once, K contiguous), B operand (N) = gathered
// input rows; accumulator tile: lane = one pixel, 4 consecutive output channels per register
// quad -> 8-byte NHWC stores.
// Workgroup = 256 threads (2x2 waves), tile 128 (O) x 128 (pixels), K chunk 64, LDS double
// buffered with register prefetch (global loads of chunk t+1 are in flight while chunk t's
// MFMAs run; they are written to LDS after the MFMAs).
// Fusions: nearest x2 upsample folded into the gather (Upsample2D), stride 2 (Downsample2D),
// bias, per-(image, channel) vector add (ResnetBlock2D's time-embedding projection), residual.
#include "common.h"

namespace {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 128;   // output channels per workgroup
constexpr int TN = 128;   // pixels per workgroup
constexpr int BK = 64;    // K chunk (halfs)
constexpr int ROW = BK + 8;  // LDS row stride in halfs: 144 B = 16 * 9 -> conflict-free b128

struct CArgs {
  const f16* in;        // (B, H, W, C)
  const f16* w;         // (O, KK, C)
  const f16* bias;      // (O) or null
  const f16* addvec;    // (B, O) or null
  const f16* residual;  // (B, OH, OW, O) or null
  f16* out;             // (B, OH, OW, O)
  int B, H, W, C, O, OH, OW, KS, stride, pad, up2;
  int Ktot;             // KS*KS*C
};

template <int DUMMY>
__global__ __launch_bounds__(256) void conv_f16_kernel(CArgs a) {
  __shared__ __attribute__((aligned(16))) f16 sA[2][TM * ROW];
  __shared__ __attribute__((aligned(16))) f16 sB[2][TN * ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;      // 2x2 waves, 64x64 each
  const int o0 = blockIdx.y * TM;
  const int64_t npix = (int64_t)a.B * a.OH * a.OW;
  const int64_t p0 = (int64_t)blockIdx.x * TN;
  const int C8 = a.C >> 3;                      // 8-channel groups per tap
  const int IH = a.up2 ? a.H * 2 : a.H, IW = a.up2 ? a.W * 2 : a.W;  // logical input size

  // staging assignment: 4 A groups + 4 B groups (16 bytes each) per thread per chunk
  int a_row[4], b_row[4], grp[4];
  int b_n[4], b_y[4], b_x[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    a_row[i] = idx >> 3;
    b_row[i] = idx >> 3;
    grp[i] = idx & 7;
    const int64_t p = p0 + b_row[i];
    b_ok[i] = p < npix;
    const int64_t pp = b_ok[i] ? p : 0;
    b_n[i] = (int)(pp / (a.OH * a.OW));
    const int rem = (int)(pp % (a.OH * a.OW));
    b_y[i] = rem / a.OW;
    b_x[i] = rem % a.OW;
  }

  f16x8 ra[4], rb[4];
  auto load_regs = [&](int chunk) {
    const int k0 = chunk * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + grp[i] * 8;
      f16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (f16)0.0f;
      ra[i] = z;
      rb[i] = z;
      if (k < a.Ktot) {
        const int o = o0 + a_row[i];
        if (o < a.O) ra[i] = *reinterpret_cast<const f16x8*>(a.w + (size_t)o * a.Ktot + k);
        if (b_ok[i]) {
          const int kg = k >> 3;
          const int tap = kg / C8, c = (kg - tap * C8) << 3;
          const int ty = tap / a.KS, tx = tap - ty * a.KS;
          int iy = b_y[i] * a.stride - a.pad + ty, ix = b_x[i] * a.stride - a.pad + tx;
          if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) {
            if (a.up2) { iy >>= 1; ix >>= 1; }
            rb[i] = *reinterpret_cast<const f16x8*>(
                a.in + (((size_t)b_n[i] * a.H + iy) * a.W + ix) * a.C + c);
          }
        }
      }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f16x8*>(&sA[buf][a_row[i] * ROW + grp[i] * 8]) = ra[i];
      *reinterpret_cast<f16x8*>(&sB[buf][b_row[i] * ROW + grp[i] * 8]) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nchunks = (a.Ktot + BK - 1) / BK;
  load_regs(0);
  store_lds(0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) load_regs(ch + 1);       // global loads in flight during the MFMAs
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const f16x8*>(
            &sA[buf][(wm * 64 + i * 32 + l31) * ROW + ks * 16 + hh * 8]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bf[j] = *reinterpret_cast<const f16x8*>(
            &sB[buf][(wn * 64 + j * 32 + l31) * ROW + ks * 16 + hh * 8]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (ch + 1 < nchunks) store_lds(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane -> pixel (wn*64 + j*32 + l31); register quad r4 -> channels
  //   o = o0 + wm*64 + i*32 + 8*r4 + 4*hh + {0..3}
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t p = p0 + wn * 64 + j * 32 + l31;
    if (p >= npix) continue;
    const int n = (int)(p / (a.OH * a.OW));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int o = o0 + wm * 64 + i * 32 + 8 * r4 + 4 * hh;
        if (o >= a.O) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * r4 + e];
        if (o + 3 < a.O && (a.O & 3) == 0) {
          if (a.bias) {
            const f16x4 bv = *reinterpret_cast<const f16x4*>(a.bias + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
          }
          if (a.addvec) {
            const f16x4 av = *reinterpret_cast<const f16x4*>(a.addvec + (size_t)n * a.O + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)av[e];
          }
          if (a.residual) {
            const f16x4 rv = *reinterpret_cast<const f16x4*>(a.residual + (size_t)p * a.O + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
          }
          f16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = (f16)v[e];
          *reinterpret_cast<f16x4*>(a.out + (size_t)p * a.O + o) = ov;
        } else {
          for (int e = 0; e < 4 && o + e < a.O; ++e) {
            float t = v[e];
            if (a.bias) t += (float)a.bias[o + e];
            if (a.addvec) t += (float)a.addvec[(size_t)n * a.O + o + e];
            if (a.residual) t += (float)a.residual[(size_t)p * a.O + o + e];
            a.out[(size_t)p * a.O + o + e] = (f16)t;
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" {

int dsu_conv2d_nhwc_f16_fwd(const void* input, const void* weight_okc, const void* bias,
                            int32_t B, int32_t H, int32_t W, int32_t C, int32_t O, int32_t k,
                            int32_t stride, int32_t pad, int32_t upsample2x, const void* addvec,
                            const void* residual, void* out, void* stream) {
  if (!input || !weight_okc || !out) return DSU_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || O <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return DSU_EINVAL;
  if (C % 8 != 0) return DSU_EUNSUP;           // 16-byte channel groups
  CArgs a;
  a.in = (const f16*)input; a.w = (const f16*)weight_okc; a.bias = (const f16*)bias;
  a.addvec = (const f16*)addvec; a.residual = (const f16*)residual; a.out = (f16*)out;
  a.B = B; a.H = H; a.W = W; a.C = C; a.O = O; a.KS = k; a.stride = stride; a.pad = pad;
  a.up2 = upsample2x ? 1 : 0;
  const int IH = a.up2 ? 2 * H : H, IW = a.up2 ? 2 * W : W;
  a.OH = (IH + 2 * pad - k) / stride + 1;
  a.OW = (IW + 2 * pad - k) / stride + 1;
  if (a.OH <= 0 || a.OW <= 0) return DSU_EINVAL;
  a.Ktot = k * k * C;
  const int64_t npix = (int64_t)B * a.OH * a.OW;
  dim3 grid((unsigned)((npix + TN - 1) / TN), (unsigned)((O + TM - 1) / TM));
  conv_f16_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
