// Radiance ("texture") MLP of the NeuS model, forward and backward, on the f32 matrix pipe
// (gfx950, v_mfma_f32_32x32x2_f32).
//
// Replaces VolumeRadiance.forward (2_charactor_reconstructor/instant_nsr/models/texture.py:9-30)
// = VanillaMLP 16 -> 64 -> 64 -> 3 with ReLU (models/network_utils.py:94-138, no weight norm for
// this network) followed by sigmoid, and its autograd backward.  The reference runs it as three
// library GEMMs + elementwise kernels forward and ~10 GEMMs / reductions backward over N ~ 2.6e5
// samples with 16..64 columns: tall-skinny shapes the library handles at a few percent of peak.
//
// One wave = 64 samples.  Activations live transposed in MFMA accumulators
//   H^T[64 hidden x 32 samples] per sample half:  lane = sample, register quad = 4 hidden units,
// and are fed straight back as the B operand of the next layer (the k-pair of one register is
// (unit, unit+4) across the two lane halves; the weight operand is permuted to match, exactly as
// in hashgrid_mfma.hip).  Only the parameter-gradient GEMMs (contraction over samples = lanes)
// go through LDS, 32 samples at a time.  Exact f32 throughout.
#include "common.h"

// -DDSU_TEX_PROF (variant build only): per-phase shader-clock totals of the backward kernel, summed
// over the waves into dsu_tex_prof[] and read back with dsu_debug_tex_prof() (tools/texture_phase_clocks.py)
#ifdef DSU_TEX_PROF
__device__ unsigned long long dsu_tex_prof[16];
#define TEX_PROF_DECL unsigned long long pt__[16] = {0}; unsigned long long pc__ = __builtin_readcyclecounter();
#define TEX_PROF(i) { const unsigned long long n__ = __builtin_readcyclecounter(); pt__[i] += n__ - pc__; pc__ = n__; }
#define TEX_PROF_END if ((threadIdx.x & 63) == 0) { for (int i__ = 0; i__ < 16; ++i__) atomicAdd(&dsu_tex_prof[i__], pt__[i__]); }
#else
#define TEX_PROF_DECL
#define TEX_PROF(i)
#define TEX_PROF_END
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = hi + mid + O(2^-16 |x|), hi and mid in bf16: operands of the "bf16 x 3" products
// a b ~ a_hi b_hi + a_hi b_mid + a_mid b_hi on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA's rate),
// used for the two backward GEMMs whose B operand is an accumulator as it stands (see the kernel)
__device__ __forceinline__ void bf16_split8(const float* x, bf16x8& hi, bf16x8& mid) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hh = (__bf16)x[i];
    hi[i] = hh;
    mid[i] = (__bf16)(x[i] - (float)hh);
  }
}

constexpr int TIN = 16, THID = 64, TOUT = 3;
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
// LDS layout (floats)
constexpr int W1_ROW = 65;                       // padded row: conflict-free by row AND by column
constexpr int W0_ROW = 17;
constexpr int L_W1 = 0;
constexpr int L_W0 = L_W1 + THID * W1_ROW;       // 4160
constexpr int L_W2P = L_W0 + THID * W0_ROW;      // 5248: w2perm[h][o][32]
constexpr int L_B0 = L_W2P + 2 * TOUT * 32;      // 5440
constexpr int L_B1 = L_B0 + THID;
constexpr int L_B2 = L_B1 + THID;                // 5568
constexpr int L_WEND = L_B2 + 8;                 // 5576
// per-wave staging (backward)
constexpr int SD_ROW = 68, SIN_ROW = 20, SDO_ROW = 4;
#ifdef DSU_TEX_GEMM_F32
// (variant build: the contraction-over-samples GEMMs in exact f32, rounds 2-5)
constexpr int S_P = 0, S_H = 32 * SD_ROW, S_IN = 2 * 32 * SD_ROW, S_DO = S_IN + 32 * SIN_ROW;
constexpr int STAGE_F = S_DO + 32 * SDO_ROW;     // 5120
#else
// Transposed bf16 images of the contraction-over-samples GEMMs (gW2, gW1, gW0 as bf16 x 3):
// [part hi | mid][row = unit (or output / input column)][TROW] bf16, element (row, s) = the value of
// sample s of the 32-sample half.  TROW = 40 (80-byte rows): the 16-byte fragments of 16 lanes
// (rows l31, l31 + 1, ...) fall on 16 different bank quads.  Two 64-row images (X, Y) and one
// 20-row image (S: dz^T, then In^T) per wave.
constexpr int TROW = 40;
constexpr int TIMG_B = 2 * 64 * TROW;            // bf16 elements of a 64-row image (hi + mid) = 10 240 B
constexpr int TSML_B = 2 * 20 * TROW;            // 20-row image
constexpr int S_X = 0, S_Y = TIMG_B / 2, S_S = TIMG_B;          // float offsets
constexpr int STAGE_F = TIMG_B + TSML_B / 2;     // 5120 + 800 = 5920 floats
constexpr int S_P = 0, S_H = 0, S_IN = 0, S_DO = 0;              // (f32 staging areas: unused)
#endif
// bf16 images of the backward GEMMs' weight operands, one 16-byte fragment per lane and MFMA:
//   W1 image [hi|mid][To][Tin][g][lane] : 8 values t -> w1[feat_of(Tin, 8 g + t, h)][32 To + l31]
//   W0 image [hi|mid][T][g][lane]       : 8 values t -> w0[feat_of(T, 8 g + t, h)][l31] (0 for l31 >= 16)
constexpr int IMG1_F = 2 * 8 * 64 * 4, IMG0_F = 2 * 4 * 64 * 4;       // floats (16 B = 4 floats per fragment)
constexpr int L_IMG1 = L_WEND + 4 * STAGE_F;
constexpr int L_IMG0 = L_IMG1 + IMG1_F;
//   W1 image of the backward's forward RECOMPUTE (L1_MASK kernels) [hi|mid][To][Tin][g][lane] :
//   8 values t -> w1[32 To + l31][feat_of(Tin, 8 g + t, h)]
constexpr int L_IMGF = L_IMG0 + IMG0_F;
constexpr int BWD_LDS_F = L_IMGF + IMG1_F;       // 5576 + 4 x 5920 + 6144 + 4096 floats = 154 KB
// per-workgroup partial vector
constexpr int P_GW1 = 0, P_GW0 = 64 * 64, P_GW2 = P_GW0 + 64 * 32, P_GB1 = P_GW2 + 64 * 32,
              P_GB2 = P_GB1 + 64, PART_N = P_GB2 + 3, PART_STRIDE = 8320;
#ifndef DSU_TEX_MAX_BLOCKS
#define DSU_TEX_MAX_BLOCKS 256
#endif
constexpr int TEX_MAX_BLOCKS = DSU_TEX_MAX_BLOCKS;

__device__ __forceinline__ int feat_of(int T, int r, int h) {
  return 32 * T + (r & 3) + 8 * (r >> 2) + 4 * h;
}

__device__ __forceinline__ void swap_halves(float a, float b, float& o0, float& o1) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  o0 = __uint_as_float(r[0]);
  o1 = __uint_as_float(r[1]);
}

__device__ __forceinline__ void load_weights(float* lds, const dsu_tex_mlp& m) {
  for (int i = threadIdx.x; i < THID * THID; i += blockDim.x)
    lds[L_W1 + (i >> 6) * W1_ROW + (i & 63)] = m.w1[i];
  for (int i = threadIdx.x; i < THID * TIN; i += blockDim.x)
    lds[L_W0 + (i >> 4) * W0_ROW + (i & 15)] = m.w0[i];
  for (int i = threadIdx.x; i < 2 * TOUT * 32; i += blockDim.x) {
    const int h = i / (TOUT * 32), o = (i / 32) % TOUT, tr = i & 31;
    lds[L_W2P + i] = m.w2[o * THID + feat_of(tr >> 4, tr & 15, h)];
  }
  for (int i = threadIdx.x; i < THID; i += blockDim.x) {
    lds[L_B0 + i] = m.b0[i];
    lds[L_B1 + i] = m.b1[i];
  }
  for (int i = threadIdx.x; i < TOUT; i += blockDim.x) lds[L_B2 + i] = m.b2[i];
}

// Hidden activations of ONE sample half `a` (0: samples of lanes 0-31, 1: lanes 32-63).
// in: the lane's OWN sample (16 inputs).  H0/H1: [hidden tile T] accumulators, post-ReLU.
// L1_BF16 (the backward's recompute when the forward handed over its ReLU pattern of layer 1): layer 1
// as bf16 x 3 on v_mfma_f32_32x32x16_bf16 — the post-ReLU accumulators of layer 0 are the B operand
// as they stand (registers 8 g + t of tile Tin = units feat_of(Tin, 8 g + t, h) of the lane's sample
// column), the A fragments come from the image at L_IMGF: 24 MFMAs of 32 clocks instead of 64 of 64.
template <bool L1_BF16 = false>
__device__ __forceinline__ void forward_half(const float* lds, const float* in, int a, int l31,
                                             int h, f32x16 (&H0)[2], f32x16 (&H1)[2]) {
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) H0[T][r] = 0.0f;
  // layer 0: H0^T = W0 . In^T (+ b0 as the k-pair (1, 0))
#pragma unroll
  for (int t = 0; t < TIN / 2; ++t) {
    float b0v, b1v;
    swap_halves(in[2 * t], in[2 * t + 1], b0v, b1v);
    const float b = a == 0 ? b0v : b1v;
#pragma unroll
    for (int T = 0; T < 2; ++T)
      H0[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[L_W0 + (32 * T + l31) * W0_ROW + 2 * t + h],
                                                   b, H0[T], 0, 0, 0);
  }
  {
    const float one = h == 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int T = 0; T < 2; ++T)
      H0[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? lds[L_B0 + 32 * T + l31] : 0.0f, one,
                                                   H0[T], 0, 0, 0);
  }
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      H0[T][r] = fmaxf(H0[T][r], 0.0f);
      H1[T][r] = 0.0f;
    }
  // layer 1: H1^T = W1 . H0^T ; register (T, r) of H0 is the k-pair (feat_of(T,r,0), +4).
  // The weight operands of the NEXT four k-pairs are requested from LDS before the eight MFMAs of
  // the current four are issued (written per k-pair, the compiler put every `ds_read` right in
  // front of its MFMA with `s_waitcnt lgkmcnt(0)` in between: one MFMA per LDS round trip).
  if constexpr (L1_BF16) {
    // (one B fragment pair at a time: the kernel has no registers for all four)
    const bf16x8* imgf = reinterpret_cast<const bf16x8*>(lds + L_IMGF);
    const int lane = l31 + 32 * h;
#pragma unroll
    for (int Tin = 0; Tin < 2; ++Tin)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = H0[Tin][8 * g + t];
        bf16x8 bh, bm;
        bf16_split8(v, bh, bm);
#pragma unroll
        for (int To = 0; To < 2; ++To) {
          const int f = ((To * 2 + Tin) * 2 + g) * 64 + lane;
          const bf16x8 ah = imgf[f], am = imgf[8 * 64 + f];
          H1[To] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, H1[To], 0, 0, 0);
          H1[To] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, H1[To], 0, 0, 0);
          H1[To] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, H1[To], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);        // (keeps the next pair's fragments out of this one's registers)
      }
  } else {
    float a[2][8];
    auto load = [&](int s_, float* dst) {
      const int T = s_ >> 2, rq = s_ & 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = feat_of(T, 4 * rq + j, h);
        dst[2 * j + 0] = lds[L_W1 + l31 * W1_ROW + k];
        dst[2 * j + 1] = lds[L_W1 + (32 + l31) * W1_ROW + k];
      }
    };
    load(0, a[0]);
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
      if (s_ + 1 < 8) load(s_ + 1, a[(s_ + 1) & 1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float b = H0[s_ >> 2][4 * (s_ & 3) + j];
        H1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][2 * j + 0], b, H1[0], 0, 0, 0);
        H1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][2 * j + 1], b, H1[1], 0, 0, 0);
      }
    }
  }
  {
    const float one = h == 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int To = 0; To < 2; ++To)
      H1[To] = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? lds[L_B1 + 32 * To + l31] : 0.0f,
                                                    one, H1[To], 0, 0, 0);
  }
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) H1[T][r] = fmaxf(H1[T][r], 0.0f);
}

// partial dot products of the 3 outputs over the 32 hidden units this lane holds (sample l31 of
// half a); the other 32 units sit in lane ^ 32
__device__ __forceinline__ void layer2_partial(const float* lds, const f32x16 (&H1)[2], int h,
                                               float (&p)[TOUT]) {
#pragma unroll
  for (int o = 0; o < TOUT; ++o) {
    const float4* w4 = reinterpret_cast<const float4*>(lds + L_W2P + (h * TOUT + o) * 32);
    float s = 0.0f;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w = w4[T * 4 + q];
        s = fmaf(w.x, H1[T][4 * q + 0], s);
        s = fmaf(w.y, H1[T][4 * q + 1], s);
        s = fmaf(w.z, H1[T][4 * q + 2], s);
        s = fmaf(w.w, H1[T][4 * q + 3], s);
      }
    p[o] = s;
  }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// SHADE: the input row is not read from a (n,16) buffer but built here from the geometry network's
// outputs — cat(feature(13), normalize(sdf_grad)(3)) (neus.py:143, texture.py:22): what
// shade_prep_fwd_kernel materialised in a separate launch.  `normal_out` (n,3) is written for the
// compositing kernels.
struct ShadeIn {
  const float* feature;   // (n,13)
  const float* grad;      // (n,3)
};

__device__ __forceinline__ void load_shaded(const ShadeIn& si, int64_t ii, float (&in)[TIN],
                                            float& inv_len) {
#pragma unroll
  for (int k = 0; k < 13; ++k) in[k] = si.feature[ii * 13 + k];
  const float g0 = si.grad[ii * 3], g1 = si.grad[ii * 3 + 1], g2 = si.grad[ii * 3 + 2];
  inv_len = 1.0f / fmaxf(sqrtf(g0 * g0 + g1 * g1 + g2 * g2), 1e-12f);
  in[13] = g0 * inv_len; in[14] = g1 * inv_len; in[15] = g2 * inv_len;
}

template <bool SHADE>
__global__ __launch_bounds__(256) void texture_fwd_kernel(dsu_tex_mlp m,
                                                          const float* __restrict__ x, ShadeIn sh,
                                                          float* __restrict__ normal_out, int64_t n,
                                                          float* __restrict__ rgb,
                                                          uint32_t* __restrict__ h1_mask) {
  __shared__ __attribute__((aligned(16))) float lds[L_WEND];
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  load_weights(lds, m);
  __syncthreads();
  // one contiguous range of samples per workgroup: the remainder of n over 256 x 256 samples
  // becomes a short last iteration everywhere (one wave, one sample half) instead of a full extra
  // round on a few CUs
  const int wave = threadIdx.x >> 6;
  const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 31) / 32 * 32;
  const int64_t r0 = blockIdx.x * per;
  const int64_t r1 = r0 + per < n ? r0 + per : n;
  for (int64_t base = r0; base < r1; base += blockDim.x) {
    const int64_t i = base + threadIdx.x;
    const bool valid = i < r1;
    const int64_t ii = valid ? i : r1 - 1;
    const int64_t wave_first = base + wave * 64;
    if (wave_first >= r1) continue;                    // wave-uniform, no barrier in the loop
    float in[TIN];
    if (SHADE) {
      float inv_len;
      load_shaded(sh, ii, in, inv_len);
      if (valid) {
        normal_out[i * 3] = in[13]; normal_out[i * 3 + 1] = in[14]; normal_out[i * 3 + 2] = in[15];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(x + ii * TIN + 4 * q);
        in[4 * q] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
      }
    }
    float out[TOUT] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int a = 0; a < 2; ++a) {
      if (wave_first + a * 32 >= r1) continue;
      f32x16 H0[2], H1[2];
      forward_half(lds, in, a, l31, h, H0, H1);
      if (h1_mask) {
        // the ReLU pattern of layer 1 for the backward (bit 16 T + r = unit feat_of(T, r, h) of the
        // sample in column l31): its recompute may then round differently without moving a mask
        uint32_t mk = 0u;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
          for (int r = 0; r < 16; ++r) mk |= (H1[T][r] > 0.0f ? 1u : 0u) << (16 * T + r);
        const int64_t si = wave_first + a * 32 + l31;
        if (si < r1) h1_mask[si * 2 + h] = mk;
      }
      float p[TOUT];
      layer2_partial(lds, H1, h, p);
#pragma unroll
      for (int o = 0; o < TOUT; ++o) {
        const float tot = p[o] + __shfl_xor(p[o], 32);     // both lanes of the pair get the sum
        if (h == a) out[o] = tot;                          // the lane whose OWN sample is in half a
      }
    }
    if (valid) {
#pragma unroll
      for (int o = 0; o < TOUT; ++o) rgb[i * TOUT + o] = sigmoidf(out[o] + lds[L_B2 + o]);
    }
  }
}

// SHADE (backward): the gradient of the input row is not stored as (n,16) but pulled back through
// cat / normalize right here (what shade_prep_bwd_kernel did in a separate launch):
//   d_feature[s] = d_x[0:13];  dn = d_x[13:16] + d_normal[s];  d_grad[s] = (dn - n (n.dn)) / |grad|
#ifndef DSU_TEX_GEMM_F32
// rows = hidden units: the lane's 32 values X[T][r] (unit 32 T + (r & 3) + 8 (r >> 2) + 4 h of the
// sample in column l31) -> bf16 hi / mid at [unit][l31] of a transposed image
__device__ __forceinline__ void stage_units_T(__bf16* img, const f32x16 (&X)[2], int l31, int h) {
  __bf16* base = img + 4 * h * TROW + l31;
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = 32 * T + (r & 3) + 8 * (r >> 2);
      const float x = X[T][r];
      const __bf16 hi = (__bf16)x;
      base[u * TROW] = hi;
      base[(64 + u) * TROW] = (__bf16)(x - (float)hi);
    }
}
// 16-byte fragment of a transposed image: 8 consecutive samples (k = 16 ks + 8 h ..) of row `row`
__device__ __forceinline__ bf16x8 frag_T(const __bf16* img, int row, int ks, int h) {
  return *reinterpret_cast<const bf16x8*>(img + row * TROW + 16 * ks + 8 * h);
}
// acc[Ti][Tj] += A^T B over the 32 samples of the half, bf16 x 3: A / B images with 64 / (32 NJ) rows
// (`rows_b` rows of B are real: lanes beyond them feed zeros)
template <int NJ>
__device__ __forceinline__ void gemm_samples_T(f32x16 (&acc0)[NJ], f32x16 (&acc1)[NJ], const __bf16* A,
                                               int a_mid, const __bf16* B, int b_mid, int rows_b,
                                               int l31, int h) {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 ah[2], am[2], bh[NJ], bm[NJ];
#pragma unroll
    for (int Ti = 0; Ti < 2; ++Ti) {
      ah[Ti] = frag_T(A, 32 * Ti + l31, ks, h);
      am[Ti] = frag_T(A + a_mid, 32 * Ti + l31, ks, h);
    }
#pragma unroll
    for (int Tj = 0; Tj < NJ; ++Tj) {
      const int row = 32 * Tj + l31;
      const bool real = row < rows_b;
      const bf16x8 vh = frag_T(B, real ? row : 0, ks, h), vm = frag_T(B + b_mid, real ? row : 0, ks, h);
      bh[Tj] = real ? vh : z;
      bm[Tj] = real ? vm : z;
    }
#pragma unroll
    for (int Tj = 0; Tj < NJ; ++Tj) {
      acc0[Tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bh[Tj], acc0[Tj], 0, 0, 0);
      acc1[Tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bh[Tj], acc1[Tj], 0, 0, 0);
      acc0[Tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bm[Tj], acc0[Tj], 0, 0, 0);
      acc1[Tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bm[Tj], acc1[Tj], 0, 0, 0);
      acc0[Tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], bh[Tj], acc0[Tj], 0, 0, 0);
      acc1[Tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1], bh[Tj], acc1[Tj], 0, 0, 0);
    }
  }
}
#endif

struct ShadeOut {
  const float* d_normal;  // (n,3) from the compositing backward
  float* d_grad;          // (n,3)
  float* d_feature;       // (n + tail,13): rows n .. n + tail - 1 are set to zero (points that are
  int64_t tail;           // not ray samples: the regulariser points of the same geometry launch)
};

template <bool SHADE, bool L1_MASK = false>
__global__ __launch_bounds__(256) void texture_bwd_kernel(
    dsu_tex_mlp m, const float* __restrict__ x, ShadeIn sh, ShadeOut so,
    const float* __restrict__ rgb, const float* __restrict__ d_rgb, int64_t n,
    float* __restrict__ d_x, float* __restrict__ partials,
    const uint32_t* __restrict__ h1_mask = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  TEX_PROF_DECL
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  float* st = lds + L_WEND + wave * STAGE_F;
  float* sP = st + S_P;
  float* sH = st + S_H;
  float* sIn = st + S_IN;
  float* sDo = st + S_DO;
#ifndef DSU_TEX_GEMM_F32
  __bf16* tX = reinterpret_cast<__bf16*>(st + S_X);
  __bf16* tY = reinterpret_cast<__bf16*>(st + S_Y);
  __bf16* tS = reinterpret_cast<__bf16*>(st + S_S);
#endif
  load_weights(lds, m);
  __syncthreads();
  {   // bf16 hi / mid fragments of W1 and W0 in the order the backward MFMAs consume them
    bf16x8* img1 = reinterpret_cast<bf16x8*>(lds + L_IMG1);
    bf16x8* img0 = reinterpret_cast<bf16x8*>(lds + L_IMG0);
    for (int f = threadIdx.x; f < 8 * 64; f += blockDim.x) {
      const int ln = f & 63, c = f >> 6, g = c & 1, Tin = (c >> 1) & 1, To = c >> 2;
      float w[8];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        w[t] = lds[L_W1 + feat_of(Tin, 8 * g + t, ln >> 5) * W1_ROW + 32 * To + (ln & 31)];
      bf16_split8(w, img1[f], img1[8 * 64 + f]);
    }
    for (int f = threadIdx.x; f < 4 * 64; f += blockDim.x) {
      const int ln = f & 63, c = f >> 6, g = c & 1, T = c >> 1;
      float w[8];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        w[t] = (ln & 31) < TIN ? lds[L_W0 + feat_of(T, 8 * g + t, ln >> 5) * W0_ROW + (ln & 31)] : 0.0f;
      bf16_split8(w, img0[f], img0[4 * 64 + f]);
    }
    if constexpr (L1_MASK) {
      bf16x8* imgf = reinterpret_cast<bf16x8*>(lds + L_IMGF);
      for (int f = threadIdx.x; f < 8 * 64; f += blockDim.x) {
        const int ln = f & 63, c = f >> 6, g = c & 1, Tin = (c >> 1) & 1, To = c >> 2;
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          w[t] = lds[L_W1 + (32 * To + (ln & 31)) * W1_ROW + feat_of(Tin, 8 * g + t, ln >> 5)];
        bf16_split8(w, imgf[f], imgf[8 * 64 + f]);
      }
    }
  }
  __syncthreads();

  if (SHADE) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < so.tail * 13;
         t += (int64_t)gridDim.x * blockDim.x)
      so.d_feature[n * 13 + t] = 0.0f;
  }
  f32x16 gw1[2][2], gw0[2], gw2[2];     // D[i = row unit][j]: W1[i][j], W0[i][k | bias], W2^T[i][o]
#ifdef DSU_TEX_GEMM_F32
  float gb1[2][16];
#else
  float gb1row = 0.0f;                 // gb1[unit = lane]: row sums of the transposed dPre1 image (see gW1)
#endif
  float gb2[TOUT] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gw1[T][0][r] = gw1[T][1][r] = 0.0f;
      gw0[T][r] = gw2[T][r] = 0.0f;
#ifdef DSU_TEX_GEMM_F32
      gb1[T][r] = 0.0f;
#endif
    }

  const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 31) / 32 * 32;   // see texture_fwd_kernel
  const int64_t r0 = blockIdx.x * per;
  const int64_t r1 = r0 + per < n ? r0 + per : n;
  for (int64_t base = r0; base < r1; base += blockDim.x) {
    const int64_t i = base + threadIdx.x;
    const bool valid = i < r1;
    const int64_t ii = valid ? i : r1 - 1;
    const int64_t wave_first = base + wave * 64;
    if (wave_first >= r1) continue;                    // wave-uniform, no workgroup barrier in the loop
    TEX_PROF(9)   // (first block: kernel prologue; later: loop overhead)
    float in[TIN];
    if (SHADE) {
      float inv_len;
      load_shaded(sh, ii, in, inv_len);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(x + ii * TIN + 4 * q);
        in[4 * q] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
      }
    }
    // gradient on the pre-sigmoid outputs of the lane's own sample (0 for padding lanes)
    float dz[TOUT];
#pragma unroll
    for (int o = 0; o < TOUT; ++o) {
      const float s = rgb[ii * TOUT + o];
      dz[o] = valid ? d_rgb[ii * TOUT + o] * s * (1.0f - s) : 0.0f;
      gb2[o] += dz[o];
    }
    uint32_t mk2[2] = {0u, 0u};         // layer 1's ReLU pattern of the samples in column l31 of both halves
    if constexpr (L1_MASK) {
#pragma unroll
      for (int a_ = 0; a_ < 2; ++a_) {
        int64_t si = wave_first + a_ * 32 + l31;
        si = si < r1 ? si : r1 - 1;
        mk2[a_] = h1_mask[si * 2 + h];
      }
    }
#pragma unroll 1
    for (int a = 0; a < 2; ++a) {
      if (wave_first + a * 32 >= r1) continue;
      TEX_PROF(0)   // rows of the block / previous half's tail
      f32x16 H0[2], H1[2];
      forward_half<L1_MASK>(lds, in, a, l31, h, H0, H1);
      const uint32_t mk = a ? mk2[1] : mk2[0];
      if constexpr (L1_MASK) {
        // the forward's pattern decides which units are live; the bf16 x 3 recompute only supplies
        // their values (2^-16 relative)
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
          for (int r = 0; r < 16; ++r) H1[T][r] = ((mk >> (16 * T + r)) & 1u) ? H1[T][r] : 0.0f;
      }
      // dz of the samples of half a, in every lane of the pair
      TEX_PROF(1)   // forward recompute
      float dza[TOUT];
#pragma unroll
      for (int o = 0; o < TOUT; ++o) {
        const float other = __shfl_xor(dz[o], 32);
        dza[o] = h == a ? dz[o] : other;
      }
      // dPre1 = (W2^T dz) * relu'(H1)
      f32x16 D1[2];
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.0f;
#pragma unroll
          for (int o = 0; o < TOUT; ++o)
            v = fmaf(lds[L_W2P + (h * TOUT + o) * 32 + T * 16 + r], dza[o], v);
          D1[T][r] = (L1_MASK ? ((mk >> (16 * T + r)) & 1u) != 0u : H1[T][r] > 0.0f) ? v : 0.0f;
#ifdef DSU_TEX_GEMM_F32
          gb1[T][r] += D1[T][r];
#endif
        }
      TEX_PROF(2)   // dPre1
      // ---- gW2^T[unit][o] += sum_samples H1[sample][unit] * dz[sample][o]
#ifdef DSU_TEX_GEMM_F32
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<float4*>(&sH[l31 * SD_ROW + 32 * T + 8 * qd + 4 * h]) =
              make_float4(H1[T][4 * qd], H1[T][4 * qd + 1], H1[T][4 * qd + 2], H1[T][4 * qd + 3]);
      if (h == a)
        *reinterpret_cast<float4*>(&sDo[l31 * SDO_ROW]) = make_float4(dz[0], dz[1], dz[2], 0.0f);
      __builtin_amdgcn_wave_barrier();
      {   // operands of the next four k-pairs in flight while the current eight MFMAs run
        float q[2][12];
        const int lo = l31 < SDO_ROW ? l31 : 0;
        auto ld = [&](int tb, float* d) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int pr = 2 * (4 * tb + u) + h;
            const float b = sDo[pr * SDO_ROW + lo];
            d[3 * u + 0] = sH[pr * SD_ROW + l31];
            d[3 * u + 1] = sH[pr * SD_ROW + 32 + l31];
            d[3 * u + 2] = l31 < SDO_ROW ? b : 0.0f;
          }
        };
        ld(0, q[0]);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          if (tb + 1 < 4) ld(tb + 1, q[(tb + 1) & 1]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float* d = q[tb & 1] + 3 * u;
            gw2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[0], d[2], gw2[0], 0, 0, 0);
            gw2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[1], d[2], gw2[1], 0, 0, 0);
          }
        }
      }
#else
      // bf16 x 3 on v_mfma_f32_32x32x16_bf16 (K = 16 samples per MFMA): H1 and dz go to LDS
      // TRANSPOSED ([unit][sample], bf16 hi / mid, two-byte stores at compile-time offsets), so that
      // a lane's operand fragment — 8 consecutive samples of one row — is one 16-byte read.
      // 12 MFMAs of 32 clocks instead of 32 of 64.
      __builtin_amdgcn_wave_barrier();
      stage_units_T(tX, H1, l31, h);
      if (h == a) {
#pragma unroll
        for (int o = 0; o < TOUT; ++o) {
          const __bf16 hi = (__bf16)dz[o];
          tS[o * TROW + l31] = hi;
          tS[(20 + o) * TROW + l31] = (__bf16)(dz[o] - (float)hi);
        }
      }
      __builtin_amdgcn_wave_barrier();
      {
        f32x16 (&g0)[1] = *reinterpret_cast<f32x16 (*)[1]>(&gw2[0]);
        f32x16 (&g1)[1] = *reinterpret_cast<f32x16 (*)[1]>(&gw2[1]);
        gemm_samples_T<1>(g0, g1, tX, 64 * TROW, tS, 20 * TROW, TOUT, l31, h);
      }
#endif
      // ---- gW1[i][j] += sum_samples dPre1[sample][i] * H0[sample][j]
      TEX_PROF(3)   // gW2
#ifdef DSU_TEX_GEMM_F32
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          *reinterpret_cast<float4*>(&sP[l31 * SD_ROW + 32 * T + 8 * qd + 4 * h]) =
              make_float4(D1[T][4 * qd], D1[T][4 * qd + 1], D1[T][4 * qd + 2], D1[T][4 * qd + 3]);
          *reinterpret_cast<float4*>(&sH[l31 * SD_ROW + 32 * T + 8 * qd + 4 * h]) =
              make_float4(H0[T][4 * qd], H0[T][4 * qd + 1], H0[T][4 * qd + 2], H0[T][4 * qd + 3]);
        }
      __builtin_amdgcn_wave_barrier();
      {
        float q[2][8];
        auto ld = [&](int tb, float* d) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int pr = 2 * (2 * tb + u) + h;
            d[4 * u + 0] = sP[pr * SD_ROW + l31];
            d[4 * u + 1] = sP[pr * SD_ROW + 32 + l31];
            d[4 * u + 2] = sH[pr * SD_ROW + l31];
            d[4 * u + 3] = sH[pr * SD_ROW + 32 + l31];
          }
        };
        ld(0, q[0]);
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
          if (tb + 1 < 8) ld(tb + 1, q[(tb + 1) & 1]);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float* d = q[tb & 1] + 4 * u;
            gw1[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[0], d[2], gw1[0][0], 0, 0, 0);
            gw1[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[0], d[3], gw1[0][1], 0, 0, 0);
            gw1[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[1], d[2], gw1[1][0], 0, 0, 0);
            gw1[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[1], d[3], gw1[1][1], 0, 0, 0);
          }
        }
      }
#else
      __builtin_amdgcn_wave_barrier();           // (the gW2 reads of X are done: in-order LDS queue of the wave)
      stage_units_T(tY, D1, l31, h);
      stage_units_T(tX, H0, l31, h);
      __builtin_amdgcn_wave_barrier();
      {   // gb1[unit] += sum over the half's samples of dPre1: lane u adds up row u of the image it was
          // just staged into (hi + mid: 2^-16 relative, like the products) — one accumulator per lane
          // instead of 32 per-column partial sums
        float srow = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16x8 vh = *reinterpret_cast<const bf16x8*>(tY + lane * TROW + 8 * q);
          const bf16x8 vm = *reinterpret_cast<const bf16x8*>(tY + (64 + lane) * TROW + 8 * q);
#pragma unroll
          for (int e = 0; e < 8; ++e) srow += (float)vh[e] + (float)vm[e];
        }
        gb1row += srow;
      }
      gemm_samples_T<2>(gw1[0], gw1[1], tY, 64 * TROW, tX, 64 * TROW, 64, l31, h);   // 24 MFMAs instead of 64
#endif
      // ---- dH0^T = W1^T . dPre1^T, then dPre0 = dH0 * relu'(H0)
      TEX_PROF(4)   // gW1
      f32x16 D0[2];
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) D0[T][r] = 0.0f;
#ifdef DSU_TEX_F32
      {   // weight operands one batch of four k-pairs ahead of their MFMAs (see forward_half)
        float a[2][8];
        auto load = [&](int s_, float* dst) {
          const int T = s_ >> 2, rq = s_ & 3;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = feat_of(T, 4 * rq + j, h);
            dst[2 * j + 0] = lds[L_W1 + k * W1_ROW + l31];
            dst[2 * j + 1] = lds[L_W1 + k * W1_ROW + 32 + l31];
          }
        };
        load(0, a[0]);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
          if (s_ + 1 < 8) load(s_ + 1, a[(s_ + 1) & 1]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float b = D1[s_ >> 2][4 * (s_ & 3) + j];
            D0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][2 * j + 0], b, D0[0], 0, 0, 0);
            D0[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][2 * j + 1], b, D0[1], 0, 0, 0);
          }
        }
      }
#else
      {   // on the bf16 matrix pipe (bf16 x 3): registers 8 g + t of a lane's dPre1 accumulator hold
          // units feat_of(Tin, 8 g + t, h) of ITS sample — the B operand of one
          // v_mfma_f32_32x32x16_bf16 as they stand; the A fragment is W1 in the same order (image
          // above).  2 x 2 x 2 x 3 = 24 MFMAs of 32 clocks instead of 64 of 64.
        const bf16x8* img1 = reinterpret_cast<const bf16x8*>(lds + L_IMG1);
        bf16x8 bh[2][2], bm[2][2];
#pragma unroll
        for (int Tin = 0; Tin < 2; ++Tin)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = D1[Tin][8 * g + t];
            bf16_split8(v, bh[Tin][g], bm[Tin][g]);
          }
#pragma unroll
        for (int To = 0; To < 2; ++To)
#pragma unroll
          for (int Tin = 0; Tin < 2; ++Tin)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int f = ((To * 2 + Tin) * 2 + g) * 64 + lane;
              const bf16x8 ah = img1[f], am = img1[8 * 64 + f];
              D0[To] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[Tin][g], D0[To], 0, 0, 0);
              D0[To] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[Tin][g], D0[To], 0, 0, 0);
              D0[To] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[Tin][g], D0[To], 0, 0, 0);
            }
      }
#endif
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) D0[T][r] = H0[T][r] > 0.0f ? D0[T][r] : 0.0f;
      // ---- gW0[i][k] += sum_samples dPre0[sample][i] * In[sample][k]  (k = 16: ones -> gb0)
      TEX_PROF(5)   // dH0 + relu'
#ifdef DSU_TEX_GEMM_F32
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<float4*>(&sP[l31 * SD_ROW + 32 * T + 8 * qd + 4 * h]) =
              make_float4(D0[T][4 * qd], D0[T][4 * qd + 1], D0[T][4 * qd + 2], D0[T][4 * qd + 3]);
      if (h == a) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(&sIn[l31 * SIN_ROW + 4 * q]) =
              make_float4(in[4 * q], in[4 * q + 1], in[4 * q + 2], in[4 * q + 3]);
        sIn[l31 * SIN_ROW + 16] = 1.0f;
      }
      __builtin_amdgcn_wave_barrier();
      {
        float q[2][12];
        const int li = l31 <= TIN ? l31 : 0;
        auto ld = [&](int tb, float* d) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int pr = 2 * (4 * tb + u) + h;
            const float b = sIn[pr * SIN_ROW + li];
            d[3 * u + 0] = sP[pr * SD_ROW + l31];
            d[3 * u + 1] = sP[pr * SD_ROW + 32 + l31];
            d[3 * u + 2] = l31 <= TIN ? b : 0.0f;
          }
        };
        ld(0, q[0]);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          if (tb + 1 < 4) ld(tb + 1, q[(tb + 1) & 1]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float* d = q[tb & 1] + 3 * u;
            gw0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[0], d[2], gw0[0], 0, 0, 0);
            gw0[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[1], d[2], gw0[1], 0, 0, 0);
          }
        }
      }
#else
      __builtin_amdgcn_wave_barrier();
      stage_units_T(tY, D0, l31, h);
      if (h == a) {
#pragma unroll
        for (int k = 0; k < TIN; ++k) {
          const __bf16 hi = (__bf16)in[k];
          tS[k * TROW + l31] = hi;
          tS[(20 + k) * TROW + l31] = (__bf16)(in[k] - (float)hi);
        }
        tS[TIN * TROW + l31] = (__bf16)1.0f;               // the ones column -> gb0
        tS[(20 + TIN) * TROW + l31] = (__bf16)0.0f;
      }
      __builtin_amdgcn_wave_barrier();
      {
        f32x16 (&g0)[1] = *reinterpret_cast<f32x16 (*)[1]>(&gw0[0]);
        f32x16 (&g1)[1] = *reinterpret_cast<f32x16 (*)[1]>(&gw0[1]);
        gemm_samples_T<1>(g0, g1, tY, 64 * TROW, tS, 20 * TROW, TIN + 1, l31, h);   // 12 MFMAs instead of 32
      }
#endif
      // ---- dIn^T = W0^T . dPre0^T : row k = (r&3) + 8(r>>2) + 4h of the sample in column l31
      TEX_PROF(6)   // gW0
      f32x16 din;
#pragma unroll
      for (int r = 0; r < 16; ++r) din[r] = 0.0f;
#ifdef DSU_TEX_F32
      {   // same batching for the 32 weight operands of dIn
        float a[2][8];
        const int lc = l31 < TIN ? l31 : 0;
        auto load = [&](int s_, float* dst) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int T = s_ >> 1, r = 8 * (s_ & 1) + j;
            const float w = lds[L_W0 + feat_of(T, r, h) * W0_ROW + lc];
            dst[j] = l31 < TIN ? w : 0.0f;
          }
        };
        load(0, a[0]);
#ifdef DSU_TEX_DIN_2ACC
        // Staged variant (not measured yet): two alternating accumulators, so that the LDS reads of
        // the next batch sit between MFMAs on DIFFERENT accumulators (MI355X_MICROARCH.md: an
        // issue slot between two MFMAs on the same accumulator costs +43 cycles); summed at the end.
        f32x16 din_b;
#pragma unroll
        for (int r = 0; r < 16; ++r) din_b[r] = 0.0f;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          if (s_ + 1 < 4) load(s_ + 1, a[(s_ + 1) & 1]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j & 1)
              din_b = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][j], D0[s_ >> 1][8 * (s_ & 1) + j],
                                                          din_b, 0, 0, 0);
            else
              din = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][j], D0[s_ >> 1][8 * (s_ & 1) + j],
                                                        din, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) din[r] += din_b[r];
#else
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          if (s_ + 1 < 4) load(s_ + 1, a[(s_ + 1) & 1]);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            din = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_ & 1][j], D0[s_ >> 1][8 * (s_ & 1) + j],
                                                      din, 0, 0, 0);
        }
#endif
      }
#else
      {   // bf16 x 3 as above: 2 x 2 x 3 = 12 MFMAs instead of 32
        const bf16x8* img0 = reinterpret_cast<const bf16x8*>(lds + L_IMG0);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = D0[T][8 * g + t];
            bf16x8 bh, bm;
            bf16_split8(v, bh, bm);
            const int f = (T * 2 + g) * 64 + lane;
            const bf16x8 ah = img0[f], am = img0[4 * 64 + f];
            din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, din, 0, 0, 0);
            din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, din, 0, 0, 0);
            din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, din, 0, 0, 0);
          }
      }
#endif
      const int64_t si = wave_first + a * 32 + l31;              // the sample of column l31
      TEX_PROF(7)   // dIn
      if (si < r1) {
        if (SHADE) {
          // lane h holds components 8q + 4h + {0..3}: h = 0 -> 0-3, 8-11; h = 1 -> 4-7, 12-15
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int comp = 8 * q + 4 * h + e;
              if (comp < 13) so.d_feature[si * 13 + comp] = din[4 * q + e];
            }
          if (h == 1) {
            const float g0 = sh.grad[si * 3], g1 = sh.grad[si * 3 + 1], g2 = sh.grad[si * 3 + 2];
            const float len = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
            const float inv = 1.0f / fmaxf(len, 1e-12f);
            const float n0 = g0 * inv, n1 = g1 * inv, n2 = g2 * inv;
            float d0 = din[5], d1 = din[6], d2 = din[7];          // components 13, 14, 15
            if (so.d_normal) {
              d0 += so.d_normal[si * 3]; d1 += so.d_normal[si * 3 + 1]; d2 += so.d_normal[si * 3 + 2];
            }
            const float dot = n0 * d0 + n1 * d1 + n2 * d2;
            const bool ok = len > 1e-12f;
            so.d_grad[si * 3] = ok ? (d0 - n0 * dot) * inv : d0 * inv;
            so.d_grad[si * 3 + 1] = ok ? (d1 - n1 * dot) * inv : d1 * inv;
            so.d_grad[si * 3 + 2] = ok ? (d2 - n2 * dot) * inv : d2 * inv;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q)
            *reinterpret_cast<float4*>(d_x + si * TIN + 8 * q + 4 * h) =
                make_float4(din[4 * q], din[4 * q + 1], din[4 * q + 2], din[4 * q + 3]);
        }
      }
      TEX_PROF(11)  // write-out of the half (sdf gradient / d_normal loads + stores)
    }
  }

  // ---- workgroup reduction of the parameter-gradient tiles -> one partial vector per workgroup.
  // Plain stores into per-wave images, one barrier, element-wise sums by all threads.  The first form added all four waves' tiles into ONE image with LDS float
  // atomics (160 per lane, every address hit by all four waves): 112 k of the kernel's 332 k clocks
  // per wave (phase clocks, profiles/round6_texture_phase_clocks.txt).  The whole LDS is free here.
  __syncthreads();
  TEX_PROF(8)   // last epilogue stores + wait for the other waves
#if defined(DSU_TEX_ABL_RED) && DSU_TEX_ABL_RED == 1
  // (timing ablation, variant build: no reduction at all — one value per lane keeps the accumulators alive)
  partials[(size_t)blockIdx.x * PART_STRIDE + threadIdx.x] =
      gw1[0][0][0] + gw1[0][1][1] + gw1[1][0][2] + gw1[1][1][3] + gw0[0][4] + gw0[1][5] + gw2[0][6] + gw2[1][7] + gb2[0];
  return;
#endif
  static_assert(3 * PART_STRIDE <= BWD_LDS_F, "three partial images fit the kernel's LDS");
  float gb2s[TOUT];
#ifdef DSU_TEX_GEMM_F32
  float gb1s[2][16];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s_ = gb1[T][r];                         // sum over the 32 sample columns of this half
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s_ += __shfl_xor(s_, o);
      gb1s[T][r] = s_;
    }
#endif
#pragma unroll
  for (int o = 0; o < TOUT; ++o) {
    float s_ = gb2[o];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s_ += __shfl_xor(s_, k);
    gb2s[o] = s_;
  }
  auto store_own = [&](float* dst) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = feat_of(T, r, h);
        dst[P_GW1 + row * 64 + l31] = gw1[T][0][r];
        dst[P_GW1 + row * 64 + 32 + l31] = gw1[T][1][r];
        dst[P_GW0 + row * 32 + l31] = gw0[T][r];
        dst[P_GW2 + row * 32 + l31] = gw2[T][r];
#ifdef DSU_TEX_GEMM_F32
        if (l31 == 0) dst[P_GB1 + row] = gb1s[T][r];
#endif
      }
#ifndef DSU_TEX_GEMM_F32
    dst[P_GB1 + lane] = gb1row;
#endif
    if (lane == 0) {
#pragma unroll
      for (int o = 0; o < TOUT; ++o) dst[P_GB2 + o] = gb2s[o];
    }
  };
  auto add_from = [&](const float* src) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = feat_of(T, r, h);
        gw1[T][0][r] += src[P_GW1 + row * 64 + l31];
        gw1[T][1][r] += src[P_GW1 + row * 64 + 32 + l31];
        gw0[T][r] += src[P_GW0 + row * 32 + l31];
        gw2[T][r] += src[P_GW2 + row * 32 + l31];
#ifdef DSU_TEX_GEMM_F32
        gb1s[T][r] += src[P_GB1 + row];
#endif
      }
#ifndef DSU_TEX_GEMM_F32
    gb1row += src[P_GB1 + lane];
#endif
#pragma unroll
    for (int o = 0; o < TOUT; ++o) gb2s[o] += src[P_GB2 + o];
  };
  if constexpr (4 * PART_STRIDE <= BWD_LDS_F) {
    // every wave's image, one barrier, then all 256 threads sum the four images element by element
    // (fixed order) and write the partial vector
    store_own(lds + wave * PART_STRIDE);
    __syncthreads();
    float* part = partials + (size_t)blockIdx.x * PART_STRIDE;
    for (int v = threadIdx.x; v < PART_N; v += blockDim.x)
      part[v] = (lds[v] + lds[PART_STRIDE + v]) + (lds[2 * PART_STRIDE + v] + lds[3 * PART_STRIDE + v]);
  } else {
    // (exact-f32 GEMM variant: three images fit) waves 2, 3 -> 0, 1; wave 1 -> 0; wave 0 writes
    if (wave >= 2) store_own(lds + (wave - 2) * PART_STRIDE);
    __syncthreads();
    if (wave < 2) add_from(lds + wave * PART_STRIDE);
    if (wave == 1) store_own(lds + 2 * PART_STRIDE);
    __syncthreads();
    if (wave == 0) {
      add_from(lds + 2 * PART_STRIDE);
      store_own(partials + (size_t)blockIdx.x * PART_STRIDE);
    }
  }
  TEX_PROF(10)  // workgroup reduction
  TEX_PROF_END
}

__global__ void texture_reduce_kernel(const float* __restrict__ partials, int nblocks,
                                      float* __restrict__ g_w0, float* __restrict__ g_b0,
                                      float* __restrict__ g_w1, float* __restrict__ g_b1,
                                      float* __restrict__ g_w2, float* __restrict__ g_b2) {
  // 64 elements x 16 slices of the workgroup range per 1024-thread workgroup
  __shared__ float red[16][64];
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int v = blockIdx.x * 64 + e;
  const bool in_range = v < PART_N;
  float acc = 0.0f;
  if (in_range)
    for (int b = sl; b < nblocks; b += 16) acc += partials[(size_t)b * PART_STRIDE + v];
  red[sl][e] = acc;
  __syncthreads();
  if (sl != 0 || !in_range) return;
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k][e];
  if (v < P_GW0) {
    g_w1[v] += s;                                   // [i][j] row-major = W1's layout
  } else if (v < P_GW2) {
    const int i = (v - P_GW0) >> 5, k = (v - P_GW0) & 31;
    if (k < TIN) g_w0[i * TIN + k] += s;
    else if (k == TIN) g_b0[i] += s;
  } else if (v < P_GB1) {
    const int i = (v - P_GW2) >> 5, o = (v - P_GW2) & 31;
    if (o < TOUT) g_w2[o * THID + i] += s;
  } else if (v < P_GB2) {
    g_b1[v - P_GB1] += s;
  } else {
    g_b2[v - P_GB2] += s;
  }
}

bool mlp_ok(const dsu_tex_mlp* m) {
  return m && m->w0 && m->b0 && m->w1 && m->b1 && m->w2 && m->b2;
}

}  // namespace

extern "C" {

int dsu_texture_fwd(const dsu_tex_mlp* mlp, const float* tex_in, int64_t n, float* rgb,
                    void* stream) {
  if (!mlp_ok(mlp) || n < 0 || (n && (!tex_in || !rgb))) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  texture_fwd_kernel<false><<<dsu_capped_blocks(n, 256, 2048), 256, 0, (hipStream_t)stream>>>(
      *mlp, tex_in, ShadeIn{nullptr, nullptr}, nullptr, n, rgb, nullptr);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_texture_fwd_shaded_m(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                             int64_t n, float* normal, float* rgb, uint32_t* h1_mask, void* stream) {
  if (!mlp_ok(mlp) || n < 0 || (n && (!feature || !grad || !normal || !rgb))) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  texture_fwd_kernel<true><<<dsu_capped_blocks(n, 256, 2048), 256, 0, (hipStream_t)stream>>>(
      *mlp, nullptr, ShadeIn{feature, grad}, normal, n, rgb, h1_mask);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_texture_fwd_shaded(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                           int64_t n, float* normal, float* rgb, void* stream) {
  return dsu_texture_fwd_shaded_m(mlp, feature, grad, n, normal, rgb, nullptr, stream);
}

int64_t dsu_texture_bwd_workspace_bytes(int64_t n) {
  if (n < 0) return DSU_EINVAL;
  return (int64_t)dsu_capped_blocks(n, 256, TEX_MAX_BLOCKS) * PART_STRIDE * sizeof(float);
}

int dsu_texture_bwd(const dsu_tex_mlp* mlp, const float* tex_in, const float* rgb,
                    const float* d_rgb, int64_t n, float* d_tex_in, float* g_w0, float* g_b0,
                    float* g_w1, float* g_b1, float* g_w2, float* g_b2, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  if (!mlp_ok(mlp) || n < 0) return DSU_EINVAL;
  if (!g_w0 || !g_b0 || !g_w1 || !g_b1 || !g_w2 || !g_b2) return DSU_EINVAL;
  if (n && (!tex_in || !rgb || !d_rgb || !d_tex_in)) return DSU_EINVAL;
  if (n == 0) return DSU_OK;
  const int64_t need = dsu_texture_bwd_workspace_bytes(n);
  if (!workspace || workspace_bytes < need) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = dsu_onewave_blocks(n, 256, TEX_MAX_BLOCKS);
  const size_t shm = (size_t)BWD_LDS_F * sizeof(float);
  static_assert(PART_N <= 4 * STAGE_F, "reduction buffer must fit the staging area");
  DSU_ENSURE_DYN_LDS(texture_bwd_kernel<false>, shm);
  texture_bwd_kernel<false><<<blocks, 256, shm, s>>>(*mlp, tex_in, ShadeIn{nullptr, nullptr},
                                                    ShadeOut{nullptr, nullptr, nullptr, 0}, rgb, d_rgb,
                                                    n, d_tex_in, (float*)workspace);
  texture_reduce_kernel<<<(PART_N + 63) / 64, 1024, 0, s>>>((const float*)workspace, blocks, g_w0,
                                                           g_b0, g_w1, g_b1, g_w2, g_b2);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_texture_bwd_shaded(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                           const float* rgb, const float* d_rgb, const float* d_normal, int64_t n,
                           int64_t tail_rows, float* d_grad, float* d_feature, float* g_w0,
                           float* g_b0, float* g_w1,
                           float* g_b1, float* g_w2, float* g_b2, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  if (!mlp_ok(mlp) || n < 0) return DSU_EINVAL;
  if (!g_w0 || !g_b0 || !g_w1 || !g_b1 || !g_w2 || !g_b2) return DSU_EINVAL;
  if (tail_rows < 0 || (n && (!feature || !grad || !rgb || !d_rgb || !d_grad || !d_feature)))
    return DSU_EINVAL;
  if (n == 0) return DSU_EUNSUP;                    // the caller zeroes the tail itself
  const int64_t need = dsu_texture_bwd_workspace_bytes(n);
  if (!workspace || workspace_bytes < need) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = dsu_onewave_blocks(n, 256, TEX_MAX_BLOCKS);
  const size_t shm = (size_t)BWD_LDS_F * sizeof(float);
  DSU_ENSURE_DYN_LDS(texture_bwd_kernel<true>, shm);
  texture_bwd_kernel<true><<<blocks, 256, shm, s>>>(*mlp, nullptr, ShadeIn{feature, grad},
                                                   ShadeOut{d_normal, d_grad, d_feature, tail_rows}, rgb, d_rgb,
                                                   n, nullptr, (float*)workspace);
  texture_reduce_kernel<<<(PART_N + 63) / 64, 1024, 0, s>>>((const float*)workspace, blocks, g_w0,
                                                           g_b0, g_w1, g_b1, g_w2, g_b2);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int32_t dsu_texture_partial_map(int32_t* map_host) {
  if (map_host) {
    // offsets into one contiguous gradient block [w0 (64,16) | b0 | w1 (64,64) | b1 | w2 (3,64) | b2]
    constexpr int O_W0 = 0, O_B0 = THID * TIN, O_W1 = O_B0 + THID, O_B1 = O_W1 + THID * THID,
                  O_W2 = O_B1 + THID, O_B2 = O_W2 + TOUT * THID;
    for (int v = 0; v < PART_N; ++v) {
      int d = -1;
      if (v < P_GW0) {
        d = O_W1 + v;
      } else if (v < P_GW2) {
        const int i = (v - P_GW0) >> 5, k = (v - P_GW0) & 31;
        if (k < TIN) d = O_W0 + i * TIN + k;
        else if (k == TIN) d = O_B0 + i;
      } else if (v < P_GB1) {
        const int i = (v - P_GW2) >> 5, o = (v - P_GW2) & 31;
        if (o < TOUT) d = O_W2 + o * THID + i;
      } else if (v < P_GB2) {
        d = O_B1 + (v - P_GB1);
      } else {
        d = O_B2 + (v - P_GB2);
      }
      map_host[v] = d;
    }
  }
  return PART_N;
}

int dsu_texture_bwd_shaded_partials(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                                    const float* rgb, const float* d_rgb, const float* d_normal,
                                    int64_t n, int64_t tail_rows, float* d_grad, float* d_feature,
                                    void* workspace, int64_t workspace_bytes,
                                    dsu_partial_reduce* red, void* stream) {
  return dsu_texture_bwd_shaded_partials_m(mlp, feature, grad, rgb, d_rgb, d_normal, n, tail_rows, d_grad,
                                           d_feature, nullptr, workspace, workspace_bytes, red, stream);
}

int dsu_texture_bwd_shaded_partials_m(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                                      const float* rgb, const float* d_rgb, const float* d_normal,
                                      int64_t n, int64_t tail_rows, float* d_grad, float* d_feature,
                                      const uint32_t* h1_mask, void* workspace, int64_t workspace_bytes,
                                      dsu_partial_reduce* red, void* stream) {
  if (!mlp_ok(mlp) || n < 0 || !red) return DSU_EINVAL;
  if (tail_rows < 0 || (n && (!feature || !grad || !rgb || !d_rgb || !d_grad || !d_feature)))
    return DSU_EINVAL;
  if (n == 0) return DSU_EUNSUP;                    // the caller zeroes the tail itself
  const int64_t need = dsu_texture_bwd_workspace_bytes(n);
  if (!workspace || workspace_bytes < need) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = dsu_onewave_blocks(n, 256, TEX_MAX_BLOCKS);
  const size_t shm = (size_t)BWD_LDS_F * sizeof(float);
  if (h1_mask) {
    DSU_ENSURE_DYN_LDS((texture_bwd_kernel<true, true>), shm);
    texture_bwd_kernel<true, true><<<blocks, 256, shm, s>>>(*mlp, nullptr, ShadeIn{feature, grad},
                                                           ShadeOut{d_normal, d_grad, d_feature, tail_rows}, rgb,
                                                           d_rgb, n, nullptr, (float*)workspace, h1_mask);
  } else {
    DSU_ENSURE_DYN_LDS(texture_bwd_kernel<true>, shm);
    texture_bwd_kernel<true><<<blocks, 256, shm, s>>>(*mlp, nullptr, ShadeIn{feature, grad},
                                                     ShadeOut{d_normal, d_grad, d_feature, tail_rows}, rgb, d_rgb,
                                                     n, nullptr, (float*)workspace);
  }
  DSU_CHECK_LAUNCH();
  red->partials = (const float*)workspace;
  red->nblocks = blocks;
  red->stride = PART_STRIDE;
  red->n = PART_N;
  return DSU_OK;
}

#ifdef DSU_TEX_PROF
int dsu_debug_tex_prof(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(dsu_tex_prof), 16 * sizeof(unsigned long long)) != hipSuccess)
    return DSU_ELAUNCH;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(dsu_tex_prof), z, sizeof(z)) != hipSuccess) return DSU_ELAUNCH;
  }
  return DSU_OK;
}
#endif

}  // extern "C"
