// Fused hash-grid + SDF-MLP kernels, MFMA formulation (gfx950, v_mfma_f32_32x32x2_f32).
//
// Same arithmetic contract and C entry points as the VALU kernels in hashgrid.hip (which stay
// selectable with DSU_SDF_IMPL=valu for A/B runs); what changes is WHERE the 23->64->13 MLP
// and its backward run: on the matrix pipe, in exact f32 (the f32 MFMA is bit-for-bit an fmaf
// chain), with the activations never leaving registers between layers.
//
// One wave = 64 points, one point per lane for the hash-grid gathers.  Layer 0 is
//     Pre^T[64 hidden x 64 points] = W0'[64 x K] . In'^T[K x 64 points]
// as 2 (hidden tiles) x 2 (point halves) 32x32 MFMA tiles; In' is the MLP input re-ordered
// {features | xyz | 1} so that the bias is a weight column and the two features of a level sit
// in one accumulator quad.  The B operand of the 32x32x2 MFMA wants In'[point][2t] from lanes
// 0-31 and In'[point][2t+1] from lanes 32-63: ONE v_permlane32_swap per k-pair turns the
// point-per-lane registers into the operands of both point halves.  The accumulator layout
// (lane = point, register quad = 4 consecutive hidden units, lane>>5 selects the quad's half)
// is fed straight back as the B operand of the backward GEMM dIn^T = W0'^T . dPre^T by
// permuting the weight rows instead of the data.  Only the parameter-gradient GEMMs
// (contraction over points = lanes) go through LDS, 32 points at a time.
#include "hashgrid_dev.h"
#include "partial_reduce.h"

#include <stdlib.h>
#include <string.h>

#include <type_traits>

using namespace dsu_hg;

// Timing ablations (DSU_BWD_ABLATE=<bits>, tools/sdf_bwd_ablation.py) switch phases of the backward
// off at run time.  -DDSU_NO_ABLATE (variant build) compiles the checks out.
#ifdef DSU_NO_ABLATE
#define DSU_ABL(bits) false
#else
#define DSU_ABL(bits) ((ablate & (bits)) != 0)
#endif

// -DDSU_BWD_PROF (variant build only): per-phase shader-clock totals of the backward kernel,
// accumulated by every wave into dsu_bwd_prof[] and read back with dsu_debug_bwd_prof().
#ifdef DSU_BWD_PROF
__device__ unsigned long long dsu_bwd_prof[16];
#define DSU_PROF_DECL unsigned long long pt__[16] = {0}; unsigned long long pc__ = __builtin_readcyclecounter();
#define DSU_PROF(i) { const unsigned long long n__ = __builtin_readcyclecounter(); pt__[i] += n__ - pc__; pc__ = n__; }
#define DSU_PROF_END if ((threadIdx.x & 63) == 0) { for (int i__ = 0; i__ < 16; ++i__) atomicAdd(&dsu_bwd_prof[i__], pt__[i__]); }
#else
#define DSU_PROF_DECL
#define DSU_PROF(i)
#define DSU_PROF_END
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Softplus of two neighbouring accumulator values in packed-f32 instructions (softplus100_pair: same
// bits as softplus100); DSU_PIPE_SP_SCALAR keeps the one-value form for A/B builds
__device__ __forceinline__ void softplus_inplace2(f32x16& acc, int r) {
#ifdef DSU_PIPE_SP_SCALAR
  acc[r] = softplus100(acc[r]);
  acc[r + 1] = softplus100(acc[r + 1]);
#else
  const f32x2 h = softplus100_pair(f32x2{acc[r], acc[r + 1]});
  acc[r] = h.x;
  acc[r + 1] = h.y;
#endif
}

// x = hi + mid + O(2^-16 |x|) with hi, mid in bf16 (round to nearest even both times): the operands
// of the "bf16 x 3" products a b ~ a_hi b_hi + a_hi b_mid + a_mid b_hi (relative error ~2^-15 per
// product, f32 accumulation) on v_mfma_f32_32x32x16_bf16 — 16x the rate of the f32 MFMA, used for the
// backward GEMMs only (the forward recompute stays exact f32).
__device__ __forceinline__ void bf16_split8(const float* x, bf16x8& hi, bf16x8& mid) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    mid[i] = (__bf16)(x[i] - (float)h);
  }
}

template <int NL>
struct MC {
  static constexpr int DIN = 3 + 2 * NL;   // reference input width (xyz first)
  static constexpr int KIN = 2 * NL + 4;   // permuted input: features | xyz | bias-one
  static constexpr int KP = KIN / 2;       // k-pairs of the 32x32x2 MFMA
};

// column of the reference W0 (row-major (64, DIN)) that permuted input k multiplies; -1 = bias
template <int NL>
__device__ __host__ constexpr int ref_col(int k) {
  return k < 2 * NL ? 3 + k : (k < 2 * NL + 3 ? k - 2 * NL : -1);
}

// hidden unit held by accumulator register r of hidden tile T in a lane of half h = lane>>5
__device__ __forceinline__ int feat_of(int T, int r, int h) {
  return 32 * T + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// a' = {lanes 0-31: a ; lanes 32-63: b of lane-32}   b' = {lanes 0-31: a of lane+32 ; 32-63: b}
__device__ __forceinline__ void swap_halves(float a, float b, float& o0, float& o1) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  o0 = __uint_as_float(r[0]);
  o1 = __uint_as_float(r[1]);
}

// value of lane ^ 32 (one v_permlane32_swap instead of a ds_bpermute round trip through LDS)
__device__ __forceinline__ float partner32(float x, int h) {
  float a, b;
  swap_halves(x, x, a, b);
  return h ? a : b;
}

// DPP row shifts inside 16-lane rows (0x101.. = row_shl:n -> lane i reads lane i+n,
// 0x111.. = row_shr:n -> lane i reads lane i-n); lanes shifted in from outside the row read 0.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// Takes a value over from the vector-memory pipeline: the compiler has to wait for the load that
// produces it HERE, and every later use sees an ordinary register (no instruction is emitted).
// (-DDSU_K1_ROLLED, variant build for A/B runs: no hand-overs and the point halves as a loop — the
// waits then sit where the compiler puts them, as before round 3)
#ifdef DSU_K1_ROLLED
__device__ __forceinline__ void vm_take(float&) {}
__device__ __forceinline__ void vm_take(__half2&) {}
#define DSU_K1_HALF_UNROLL(c) 1
#else
#define DSU_K1_HALF_UNROLL(c) ((c) ? 2 : 1)
__device__ __forceinline__ void vm_take(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void vm_take(__half2& x) {
  uint32_t u;
  __builtin_memcpy(&u, &x, 4);
  asm volatile("" : "+v"(u));
  __builtin_memcpy(&x, &u, 4);
}
#endif

template <int NL>
struct Frags {
  float w0[2][MC<NL>::KP];   // A operand of layer 0: W0'[32T + (lane&31)][2t + (lane>>5)]
  float w1o0[2][16];         // W1[0][feat_of(T, r, h)]
  float b1o0;
};

template <int NL>
__device__ __forceinline__ float w0p(const dsu_sdf_mlp& m, int feat, int k) {
  const int c = ref_col<NL>(k);
  return c >= 0 ? m.w0[feat * MC<NL>::DIN + c] : m.b0[feat];
}

template <int NL>
__device__ __forceinline__ void load_frags(const dsu_sdf_mlp& m, Frags<NL>& f, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int T = 0; T < 2; ++T) {
#pragma unroll
    for (int t = 0; t < MC<NL>::KP; ++t) f.w0[T][t] = w0p<NL>(m, 32 * T + l31, 2 * t + h);
#pragma unroll
    for (int r = 0; r < 16; ++r) f.w1o0[T][r] = m.w1[feat_of(T, r, h)];
  }
  f.b1o0 = m.b1[0];
}

// MLP input of one contracted point in the permuted order
template <int NL>
__device__ __forceinline__ void encode_input_p(const __half2* __restrict__ table,
                                               const GridMeta& m, uint32_t active, float x,
                                               float y, float z, float* in /*KIN*/) {
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    float2 f = make_float2(0.0f, 0.0f);
    if ((uint32_t)l < active) f = __half22float2(lookup_level(table, m, l, x, y, z));
    in[2 * l] = f.x;
    in[2 * l + 1] = f.y;
  }
  in[2 * NL + 0] = x * 2.0f + -1.0f;
  in[2 * NL + 1] = y * 2.0f + -1.0f;
  in[2 * NL + 2] = z * 2.0f + -1.0f;
  in[2 * NL + 3] = 1.0f;
}

// acc[half][T]: hidden pre-activations; then softplus in place.
template <int NL>
__device__ __forceinline__ void layer0_mfma(const Frags<NL>& f, const float* in, uint32_t active,
                                            f32x16 (&acc)[2][2], int ablate = 0) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][T][r] = 0.0f;
#pragma unroll
  for (int t = 0; t < MC<NL>::KP; ++t) {
    if (t < NL && (uint32_t)t >= active) continue;   // masked level: both inputs are zero
    float b0, b1;
    swap_halves(in[2 * t], in[2 * t + 1], b0, b1);
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      acc[0][T] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w0[T][t], b0, acc[0][T], 0, 0, 0);
      acc[1][T] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w0[T][t], b1, acc[1][T], 0, 0, 0);
    }
  }
  if (DSU_ABL(32)) return;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][T][r] = softplus100(acc[a][T][r]);
}

// Hidden activations of ONE point half (the backward kernel walks the halves one at a time: a
// `H[half]` array indexed by the run-time half lived in scratch memory, 320 B per lane).
template <int NL>
__device__ __forceinline__ void layer0_mfma_half(const Frags<NL>& f, const float* in,
                                                 uint32_t active, int half, f32x16 (&acc)[2],
                                                 int ablate = 0) {
  // B operands of this half's points, one per k-pair
  float b[MC<NL>::KP];
#pragma unroll
  for (int t = 0; t < MC<NL>::KP; ++t) {
    float b0, b1;
    swap_halves(in[2 * t], in[2 * t + 1], b0, b1);
    b[t] = half ? b1 : b0;
  }
#ifdef DSU_L0_INTERLEAVED
  // A/B variant: the two hidden tiles alternate per k-pair (the form before round 2's tile-major
  // order; MFMAs on different accumulators back to back, all Softplus work afterwards)
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[T][r] = 0.0f;
#pragma unroll
  for (int t = 0; t < MC<NL>::KP; ++t) {
    if (t < NL && (uint32_t)t >= active) continue;     // masked level: both inputs are zero
#pragma unroll
    for (int T = 0; T < 2; ++T)
      acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w0[T][t], b[t], acc[T], 0, 0, 0);
  }
#else
  // tile by tile: the Softplus of tile 0 (VALU + transcendental unit) runs while the matrix pipe
  // works through the k-pairs of tile 1 (both tiles interleaved finished together and the 64
  // Softplus evaluations started only then: 1.5 k + 1.6 k clocks in sequence per half)
#pragma unroll
  for (int T = 0; T < 2; ++T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[T][r] = 0.0f;
#pragma unroll
    for (int t = 0; t < MC<NL>::KP; ++t) {
      if (t < NL && (uint32_t)t >= active) continue;   // masked level: both inputs are zero
      acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w0[T][t], b[t], acc[T], 0, 0, 0);
    }
  }
#endif
  if (DSU_ABL(32)) return;
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[T][r] = softplus100(acc[T][r]);
}

// out[0] of the lane's OWN point from the hidden activations spread over lane and lane^32
template <int NL>
__device__ __forceinline__ float layer1_o0(const Frags<NL>& f, const f32x16 (&H)[2][2], int h) {
  float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p0 = fmaf(f.w1o0[T][r], H[0][T][r], p0);
      p1 = fmaf(f.w1o0[T][r], H[1][T][r], p1);
    }
  const float send = h == 0 ? p1 : p0;
  const float recv = __shfl_xor(send, 32);
  return (h == 0 ? p0 : p1) + recv + f.b1o0;
}

// outputs 1..12 (the centre evaluation's feature vector); W1 rows permuted in LDS:
// w1perm[h][o][T*16 + r] = W1[o][feat_of(T, r, h)]
__device__ __forceinline__ void layer1_rest(const float* w1perm, const float* b1s,
                                            const f32x16 (&H)[2][2], int h, float* out /*13*/) {
  const float* wp = w1perm + h * NOUT * 32;
#pragma unroll 1
  for (int o = 1; o < NOUT; ++o) {
    float p0 = 0.0f, p1 = 0.0f;
    const float4* w4 = reinterpret_cast<const float4*>(wp + o * 32);
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w = w4[T * 4 + q];
        p0 = fmaf(w.x, H[0][T][4 * q + 0], p0); p1 = fmaf(w.x, H[1][T][4 * q + 0], p1);
        p0 = fmaf(w.y, H[0][T][4 * q + 1], p0); p1 = fmaf(w.y, H[1][T][4 * q + 1], p1);
        p0 = fmaf(w.z, H[0][T][4 * q + 2], p0); p1 = fmaf(w.z, H[1][T][4 * q + 2], p1);
        p0 = fmaf(w.w, H[0][T][4 * q + 3], p0); p1 = fmaf(w.w, H[1][T][4 * q + 3], p1);
      }
    const float send = h == 0 ? p1 : p0;
    const float recv = __shfl_xor(send, 32);
    out[o] = (h == 0 ? p0 : p1) + recv + b1s[o];
  }
}

__device__ __forceinline__ void load_w1perm(float* w1perm, float* b1s, const dsu_sdf_mlp& m) {
  for (int i = threadIdx.x; i < 2 * NOUT * 32; i += blockDim.x) {
    const int h = i / (NOUT * 32), o = (i / 32) % NOUT, tr = i % 32;
    w1perm[i] = m.w1[o * HID + feat_of(tr >> 4, tr & 15, h)];
  }
  for (int i = threadIdx.x; i < 16; i += blockDim.x) b1s[i] = i < NOUT ? m.b1[i] : 0.0f;
}

__device__ __forceinline__ void fd_point(const float p[3], int e, float eps, float radius,
                                         float q[3]) {
  q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
  if (e > 0) {
    const int ax = (e - 1) >> 1;
    const float d = ((e - 1) & 1) ? -eps : eps;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = q[a] + (a == ax ? d : 0.0f);
      q[a] = fminf(fmaxf(v, -radius), radius);      // (points_ + offsets).clamp  (geometry.py:170)
    }
  }
}

// ------------------------------------------------------------------------------- forward
template <int NL, int NO>
__global__ __launch_bounds__(256) void sdf_fwd_mfma_kernel(const __half2* __restrict__ table,
                                                           GridMeta m, dsu_sdf_mlp mlp,
                                                           const float* __restrict__ pts,
                                                           int64_t n, float radius,
                                                           uint32_t active,
                                                           float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float w1perm[2 * NOUT * 32];
  __shared__ float b1s[16];
  const int lane = threadIdx.x & 63, h = lane >> 5;
  Frags<NL> fr;
  load_frags<NL>(mlp, fr, lane);
  if (NO > 1) {
    load_w1perm(w1perm, b1s, mlp);
    __syncthreads();
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = blockIdx.x * (int64_t)blockDim.x; base < n; base += stride) {
    const int64_t i = base + threadIdx.x;
    const bool valid = i < n;
    const int64_t ii = valid ? i : n - 1;
    float in[MC<NL>::KIN];
    encode_input_p<NL>(table, m, active, contract(pts[ii * 3], radius),
                       contract(pts[ii * 3 + 1], radius), contract(pts[ii * 3 + 2], radius), in);
    f32x16 H[2][2];
    layer0_mfma<NL>(fr, in, active, H);
    float o[NOUT];
    o[0] = layer1_o0<NL>(fr, H, h);
    if (NO > 1) layer1_rest(w1perm, b1s, H, h, o);
    if (valid) {
#pragma unroll
      for (int k = 0; k < NO; ++k) out[i * NO + k] = o[k];
    }
  }
}

template <int NL>
__global__ __launch_bounds__(256) void sdf_fd_fwd_mfma_kernel(
    const __half2* __restrict__ table, GridMeta m, dsu_sdf_mlp mlp,
    const float* __restrict__ pts, int64_t n, float radius, float eps, float eps2,
    uint32_t active, float* __restrict__ sdf, float* __restrict__ grad,
    float* __restrict__ feature, float* __restrict__ laplace) {
  __shared__ __attribute__((aligned(16))) float w1perm[2 * NOUT * 32];
  __shared__ float b1s[16];
  const int lane = threadIdx.x & 63, h = lane >> 5;
  Frags<NL> fr;
  load_frags<NL>(mlp, fr, lane);
  load_w1perm(w1perm, b1s, mlp);
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = blockIdx.x * (int64_t)blockDim.x; base < n; base += stride) {
    const int64_t i = base + threadIdx.x;
    const bool valid = i < n;
    const int64_t ii = valid ? i : n - 1;
    const float p[3] = {pts[ii * 3], pts[ii * 3 + 1], pts[ii * 3 + 2]};
    float s[7];
#pragma unroll 1
    for (int e = 0; e < 7; ++e) {
      float q[3];
      fd_point(p, e, eps, radius, q);
      float in[MC<NL>::KIN];
      encode_input_p<NL>(table, m, active, contract(q[0], radius), contract(q[1], radius),
                         contract(q[2], radius), in);
      f32x16 H[2][2];
      layer0_mfma<NL>(fr, in, active, H);
      s[e] = layer1_o0<NL>(fr, H, h);
      if (e == 0 && feature != nullptr) {
        float o[NOUT];
        o[0] = s[0];
        layer1_rest(w1perm, b1s, H, h, o);
        if (valid) {
#pragma unroll
          for (int k = 0; k < NOUT; ++k) feature[i * NOUT + k] = o[k];
        }
      }
    }
    if (valid) {
      sdf[i] = s[0];
      if (grad != nullptr) {
        grad[i * 3 + 0] = 0.5f * (s[1] - s[2]) / eps;
        grad[i * 3 + 1] = 0.5f * (s[3] - s[4]) / eps;
        grad[i * 3 + 2] = 0.5f * (s[5] - s[6]) / eps;
      }
      if (laplace != nullptr) {
        const float t0 = s[1] + s[2] - 2.0f * s[0];
        const float t1 = s[3] + s[4] - 2.0f * s[0];
        const float t2 = s[5] + s[6] - 2.0f * s[0];
        laplace[i] = ((t0 + t1) + t2) / eps2;
      }
    }
  }
}

// ------------------------------------------------------------------------------- backward
// LDS (floats): w1perm | b1s | 4 x per-wave staging {sd[32][68], sin[32][36], sdo[32][36]} | cache
constexpr int SD_ROW = 68, SIN_ROW = 36;
constexpr int STAGE_F = 32 * SD_ROW + 2 * 32 * SIN_ROW;   // 4480 floats per wave
constexpr int W1P_F = 2 * NOUT * 32 + 16;
constexpr int BWD_CACHE_OFF = W1P_F + 4 * STAGE_F;
// gradient cache: keys u32[GC_SLOTS] | accumulators i64[GC_SLOTS][2] in fixed point (see gc_fix)
constexpr int BWD_LDS_F = BWD_CACHE_OFF + GC_SLOTS + 4 * GC_SLOTS;
// per-wave contribution queue (aliases the staging area while the scatter of a half runs)
constexpr int QCAP = 1408;                                 // (entry, d0, d1) triples
static_assert(3 * QCAP <= STAGE_F, "queue must fit the staging area");
static_assert((BWD_CACHE_OFF + GC_SLOTS) % 2 == 0, "i64 accumulators need 8-byte alignment");

// LDS float atomics retire ~1 lane per 3 clocks per CU (measured, profiles/round1_lds_atomics.txt)
// while integer atomics cost ~10 clocks per instruction whatever the lane count, so the cache
// accumulates in 64-bit fixed point (2^-40 resolution, +-8.4e6 range): ~20x cheaper per
// contribution when all lanes are busy, and order-independent.  A single contribution is
// clamped to +-1023 (a per-sample table gradient that large means the run has diverged).
constexpr float GC_FIX_SCALE = 1099511627776.0f;           // 2^40
constexpr float GC_FIX_MAX = 1023.0f;                      // per-contribution clamp (see gc_fix)
// float -> fixed point without the (slow, emulated) f32->i64 conversion: adding 1.5 * 2^52 to
// v * 2^40 in double leaves round-to-nearest(v * 2^40) in the low mantissa bits, so the
// difference of the two bit patterns IS the integer (exact for |v * 2^40| < 2^51).
__device__ __forceinline__ unsigned long long gc_fix(float v) {
  const double magic = 6755399441055744.0;                 // 1.5 * 2^52
  const float c = fminf(fmaxf(v, -GC_FIX_MAX), GC_FIX_MAX);
  const double d = (double)c * (double)GC_FIX_SCALE + magic;
  return (unsigned long long)(__double_as_longlong(d) - __double_as_longlong(magic));
}
__device__ __forceinline__ float gc_unfix(unsigned long long a) {
  return (float)((double)(long long)a * (1.0 / (double)GC_FIX_SCALE));
}

// a queued contribution whose first cache slot is owned by another entry: two more probes, then
// a global float atomic (the cache is full around that slot)
__device__ __forceinline__ void gc_commit_from(uint32_t* keys, unsigned long long* acc,
                                               float* __restrict__ gtable, uint32_t entry,
                                               uint32_t slot, float v0, float v1) {
#pragma unroll
  for (int probe = 1; probe < 3; ++probe) {
    slot = (slot + 1) & (GC_SLOTS - 1);
    const uint32_t old = atomicCAS(&keys[slot], GC_EMPTY, entry);
    if (old == GC_EMPTY || old == entry) {
      atomicAdd(&acc[2 * slot], gc_fix(v0));
      atomicAdd(&acc[2 * slot + 1], gc_fix(v1));
      return;
    }
  }
  unsafeAtomicAdd(gtable + (size_t)entry * 2, v0);
  unsafeAtomicAdd(gtable + (size_t)entry * 2 + 1, v1);
}

// per-workgroup partial vector: gw0p[64 feat][32 k'] | gw1p[64 feat][32 o'] | gb1[16]
constexpr int PART_GW0 = 0, PART_GW1 = 64 * 32, PART_GB1 = 2 * 64 * 32;
constexpr int PART_STRIDE = 2 * 64 * 32 + 64;

// SPLIT = false: one kernel does everything (table-gradient scatter included).
// SPLIT = true : the kernel stops at dIn — the gradient on the interpolated features of every
//   (evaluation, point, level) is written to `dinbuf` [eval][point][active level] float2 and
//   sdf_fd_scatter_kernel turns it into table gradients.  Without the 80 KB gradient cache the
//   MLP part fits two workgroups per CU (two waves per SIMD instead of one).
// ENC = true: the interpolated features come from the forward pass's cache (the table-gather path
// and its 40+ scalar registers of level metadata are compiled out).
template <int NL, bool SPLIT, bool ENC>
__global__ __launch_bounds__(256) void sdf_fd_bwd_mfma_kernel(
    const __half2* __restrict__ table, GridMeta m, dsu_sdf_mlp mlp,
    const float* __restrict__ pts, int64_t n, float radius, float eps, float eps2,
    uint32_t active, const float* __restrict__ d_sdf, const float* __restrict__ d_grad,
    const float* __restrict__ d_feature, const float* __restrict__ d_laplace,
    float* __restrict__ gtable, float* __restrict__ partials, const __half2* __restrict__ enc,
    float2* __restrict__ dinbuf, const int32_t* __restrict__ perm, int ablate) {
  constexpr int KIN = MC<NL>::KIN;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* w1perm = lds;
  float* b1s = lds + 2 * NOUT * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  float* sd = lds + W1P_F + wave * STAGE_F;
  float* sin_ = sd + 32 * SD_ROW;
  float* sdo = sin_ + 32 * SIN_ROW;
  uint32_t* c_keys = reinterpret_cast<uint32_t*>(lds + BWD_CACHE_OFF);
  unsigned long long* c_acc =
      reinterpret_cast<unsigned long long*>(lds + BWD_CACHE_OFF + GC_SLOTS);
  uint32_t* q_ent = reinterpret_cast<uint32_t*>(sd);       // queue aliases sd/sin/sdo
  float* q_v0 = sd + QCAP;
  float* q_v1 = sd + 2 * QCAP;

  // the scatter kernel's work counter (behind the dIn buffer) starts every launch at zero
  if (SPLIT && blockIdx.x == 0 && threadIdx.x == 0)
    *reinterpret_cast<int*>(dinbuf + (size_t)7 * (size_t)n * NL) = 0;
  Frags<NL> fr;
  load_frags<NL>(mlp, fr, lane);
  // A operand of dIn^T = W0'^T . dPre^T : W0'[feat_of(T, r, h)][k = lane & 31]
  float w0t[2][16];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      w0t[T][r] = l31 < KIN ? w0p<NL>(mlp, feat_of(T, r, h), l31) : 0.0f;
#ifndef DSU_DIN_F32
  // dIn^T = W0'^T . dPre^T on the bf16 matrix pipe (bf16 x 3).  The contraction index of one
  // v_mfma_f32_32x32x16_bf16 is 8 values per lane half: lane (l31, h) supplies k = 8 h + t.  The
  // 32x32 accumulator layout of dPre already holds, in registers 8 g + t of a lane, hidden units
  // feat_of(T, 8 g + t, h) of ITS point: they are the B operand as they are (no lane exchange), and
  // the A operand is w0t[T][8 g + t] of the same lane — W0'[feat_of(T, 8 g + t, h)][k' = l31] —
  // i.e. the SAME permutation of the contraction index on both sides.  2 tiles x 2 register groups
  // x 3 products = 12 MFMAs of 32 clocks instead of 32 of 64.
  bf16x8 w0t_hi[2][2], w0t_mid[2][2];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int g = 0; g < 2; ++g) bf16_split8(&w0t[T][8 * g], w0t_hi[T][g], w0t_mid[T][g]);
#endif
  load_w1perm(w1perm, b1s, mlp);
  if (!SPLIT) {
    for (int t = threadIdx.x; t < GC_SLOTS; t += blockDim.x) {
      c_keys[t] = GC_EMPTY;
      c_acc[2 * t] = 0ull;
      c_acc[2 * t + 1] = 0ull;
    }
  }
  __syncthreads();

  f32x16 gw0[2], gw1[2];   // [feat tile]: D[i = feat][j = k' or o']
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) gw0[T][r] = gw1[T][r] = 0.0f;
  float gb1[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) gb1[o] = 0.0f;
  float gw1c0[2][16];      // column o = 0 of gW1 from the offset evaluations, per point column
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) gw1c0[T][r] = 0.0f;

  DSU_PROF_DECL
  // One contiguous range of points per workgroup (one workgroup per CU, one wave per SIMD): the
  // remainder of n over 256 CUs x 256 points is spread over ALL workgroups as a last partial
  // iteration of <= 32 points — one wave, one point half — instead of a full extra round on a
  // few CUs (n ~ 2^18 + 4096 made the grid-stride form run 5 rounds where 4.06 were needed).
  const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 31) / 32 * 32;
  const int64_t r0 = blockIdx.x * per;
  const int64_t r1 = r0 + per < n ? r0 + per : n;
  for (int64_t bbase = r0; bbase < r1; bbase += blockDim.x) {
    // The last, partial iteration of a range holds one or two 64-point blocks (n ~ 2^18 + 4096
    // leaves 32 points per workgroup): walked like the full ones it costs wave 0 (and 1) seven
    // evaluations while the others wait — a ninth point half where 8.1 are needed, 1/9 of the
    // kernel.  The evaluations of a point are independent here (per-wave partial sums, dIn rows
    // per evaluation), so the waves share the blocks instead: all four take block 0 and every
    // fourth evaluation (two evaluations at most), or two waves per block and every second one.
    int blk = wave, e_start = 0, e_step = 1;
#ifndef DSU_K1_PLAIN_TAIL
    if (SPLIT && ENC) {
      const int64_t left = r1 - bbase;
      if (left <= 64) { blk = 0; e_start = wave; e_step = 4; }
      else if (left <= 128) { blk = wave & 1; e_start = wave >> 1; e_step = 2; }
    }
#endif
    const int64_t i = bbase + blk * 64 + lane;
    const bool valid = i < r1;
    const int64_t ii = valid ? i : r1 - 1;
    const int64_t wave_first = bbase + blk * 64;             // wave-uniform
    const float p[3] = {pts[ii * 3], pts[ii * 3 + 1], pts[ii * 3 + 2]};
    float ds = 0.f, dl = 0.f, dg[3] = {0.f, 0.f, 0.f};
    // sorted evaluation order: the upstream gradients stay in the caller's row order
    const int64_t gi = perm ? (int64_t)perm[ii] : ii;
    // (NL = 12 has no registers left for the hoisted loads and the second row below: it keeps the
    // form with the loads inside the evaluation loop)
    constexpr bool HANDOVER = NL <= 10;
    float dfe[HANDOVER ? NOUT : 1];    // upstream gradient on the centre evaluation's feature vector
#pragma unroll
    for (int o = 0; o < (HANDOVER ? NOUT : 1); ++o) dfe[o] = 0.0f;
    if (valid) {
      if (d_sdf) ds = d_sdf[gi];
      if (d_laplace) dl = d_laplace[gi];
      if (d_grad) { dg[0] = d_grad[gi * 3]; dg[1] = d_grad[gi * 3 + 1]; dg[2] = d_grad[gi * 3 + 2]; }
      if (HANDOVER && d_feature) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) dfe[HANDOVER ? o : 0] = d_feature[gi * NOUT + o];
      }
    }
    // Feature-cache rows (ENC path): `rw` is the row of the CURRENT evaluation, `rwn` the one in
    // flight for the next.  Where the kernel waits for a load matters at one wave per SIMD, and the
    // wait counter is shared by loads and stores (a wait for a load while stores are in flight is
    // vmcnt(0)): a use of a loaded register right behind the dIn stores of an evaluation, or a
    // use inside a loop entered right behind the loads (the compiler then drains the counter in
    // the loop preheader), cost ~3.6 k clocks per evaluation (10 % of the kernel, phase clocks).
    // Every loaded value is therefore taken over (`vm_take`) at a point where everything in flight
    // was issued thousands of clocks earlier: the prologue values here, the next row in front of
    // the second point half's stores.
    __half2 rw[NL], rwn[HANDOVER ? NL : 1];
#pragma unroll
    for (int l = 0; l < NL; ++l) rw[l] = __float2half2_rn(0.0f);
#pragma unroll
    for (int l = 0; l < (HANDOVER ? NL : 1); ++l) rwn[l] = __float2half2_rn(0.0f);
    if (ENC) {
      // (unconditional loads from a clamped column: a branch per level kept the compiler from
      // issuing the row's loads together; masked levels are zeroed where the row is used)
      const __half2* row = enc + ((size_t)e_start * n + ii) * active;
#pragma unroll
      for (int l = 0; l < NL; ++l) rw[l] = row[(uint32_t)l < active ? l : 0];
      if (HANDOVER) {
#pragma unroll
        for (int l = 0; l < NL; ++l) vm_take(rw[l]);
        vm_take(ds); vm_take(dl); vm_take(dg[0]); vm_take(dg[1]); vm_take(dg[2]);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) vm_take(dfe[HANDOVER ? o : 0]);
      }
    }
    const int n_eval = wave_first < r1 ? 7 : 0;             // a wave beyond the range only joins the barriers
#pragma unroll 1
    for (int e = e_start; e < n_eval; e += e_step) {
      float q[3];
      fd_point(p, e, eps, radius, q);
      const float cx = contract(q[0], radius), cy = contract(q[1], radius),
                  cz = contract(q[2], radius);
      float in[KIN];
      if (DSU_ABL(16)) {
#pragma unroll
        for (int k = 0; k < KIN; ++k) in[k] = cx * (float)k + cy;
      } else if (ENC) {
        // features saved by the forward pass: no table gathers in the backward pass.  The row of
        // evaluation e was requested one evaluation earlier (at one wave per SIMD nothing else
        // hides the ~1.2 us of a dependent global load: 10 % of the kernel's clocks)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          const float2 f = __half22float2(rw[l]);
          in[2 * l] = (uint32_t)l < active ? f.x : 0.0f;
          in[2 * l + 1] = (uint32_t)l < active ? f.y : 0.0f;
        }
        if (e + e_step < 7) {
          const __half2* row = enc + ((size_t)(e + e_step) * n + ii) * active;
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            if (HANDOVER) rwn[HANDOVER ? l : 0] = row[(uint32_t)l < active ? l : 0];
            else rw[l] = row[(uint32_t)l < active ? l : 0];
          }
        }
        in[2 * NL + 0] = cx * 2.0f + -1.0f;
        in[2 * NL + 1] = cy * 2.0f + -1.0f;
        in[2 * NL + 2] = cz * 2.0f + -1.0f;
        in[2 * NL + 3] = 1.0f;
      } else {
        encode_input_p<NL>(table, m, active, cx, cy, cz, in);
      }
      DSU_PROF(0)   // positions + feature-cache row
      // upstream gradient on this evaluation's outputs (own point)
      float dout[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) dout[o] = 0.0f;
      if (valid) {
        if (e == 0) {
          if (HANDOVER) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) dout[o] = dfe[HANDOVER ? o : 0];
          } else if (d_feature) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) dout[o] = d_feature[gi * NOUT + o];
          }
          dout[0] += ds - 6.0f * dl / eps2;
        } else {
          const int ax = (e - 1) >> 1;
          const float sgn = ((e - 1) & 1) ? -1.0f : 1.0f;
          dout[0] = sgn * 0.5f * dg[ax] / eps + dl / eps2;
        }
      }
      const int no = e == 0 ? NOUT : 1;
#pragma unroll
      for (int o = 0; o < NOUT; ++o)
        if (o < no) gb1[o] += dout[o];
      // partner's position (for the scatter of the other point half)
      const float pcx = partner32(cx, h), pcy = partner32(cy, h), pcz = partner32(cz, h);

      DSU_PROF(1)   // upstream gradient loads
      // (SPLIT && ENC: both halves as straight-line code, so that the hand-over of the next row
      // below is on every path and not inside a loop)
#pragma unroll DSU_K1_HALF_UNROLL(SPLIT && ENC && NL <= 10)
      for (int half = 0; half < 2; ++half) {
        const bool live = wave_first + half * 32 < r1;       // a point in this half (wave-uniform)
        f32x16 din;
        f32x16 Hh[2];
        float d[NOUT];
        if (live) {
        // gradient on the outputs of the points of this half: own if this lane owns the half
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          float other = 0.0f;
          if (o < no) other = partner32(dout[o], h);
          d[o] = (h == half) ? dout[o] : other;
        }
        DSU_PROF(2)   // partner shuffles
        layer0_mfma_half<NL>(fr, in, active, half, Hh, ablate);
        DSU_PROF(3)   // layer 0 + softplus
#pragma unroll
        for (int r = 0; r < 16; ++r) din[r] = 0.0f;
#ifdef DSU_DIN_2ACC
        f32x16 din_b;
#pragma unroll
        for (int r = 0; r < 16; ++r) din_b[r] = 0.0f;
#endif
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          float dpre[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) dpre[r] = fr.w1o0[T][r] * d[0];
          if (e == 0) {
            const float* wp = w1perm + h * NOUT * 32 + T * 16;
#pragma unroll 1
            for (int o = 1; o < NOUT; ++o) {
              const float4* w4 = reinterpret_cast<const float4*>(wp + o * 32);
              const float dv = d[o];
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                const float4 w = w4[qd];
                dpre[4 * qd + 0] = fmaf(w.x, dv, dpre[4 * qd + 0]);
                dpre[4 * qd + 1] = fmaf(w.y, dv, dpre[4 * qd + 1]);
                dpre[4 * qd + 2] = fmaf(w.z, dv, dpre[4 * qd + 2]);
                dpre[4 * qd + 3] = fmaf(w.w, dv, dpre[4 * qd + 3]);
              }
            }
          }
#ifdef DSU_DIN_BATCH
          // Staged variant (not measured yet): the 16 derivative factors first, then the 16 MFMAs of
          // the tile back to back.  Interleaved as below, every MFMA on the ONE accumulator `din`
          // has ~4 VALU issue slots in front of it, and MI355X_MICROARCH.md prices an extra issue
          // slot between two MFMAs on the same accumulator at +43 cycles (+6 per further one): the
          // phase runs at ~116 cycles per MFMA (profiles/round2_sdf_bwd_k1_phase_clocks.txt: 28 %
          // of the kernel) instead of 64.  Same operations in the same order per value.
          if (!DSU_ABL(64)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              dpre[r] *= 1.0f - __builtin_amdgcn_exp2f(Hh[T][r] * -144.26950408889634f);
#pragma unroll
            for (int r = 0; r < 16; ++r)
              din = __builtin_amdgcn_mfma_f32_32x32x2f32(w0t[T][r], dpre[r], din, 0, 0, 0);
          }
#elif defined(DSU_DIN_2ACC)
          // Staged variant (not measured yet): even / odd hidden units accumulate into two
          // accumulators, so that the VALU work sits between MFMAs on DIFFERENT accumulators
          // (~6 cycles per issue slot instead of the +43 cliff); the two partial sums are added
          // after the second tile (summation order differs from the default: not bit-identical).
          if (!DSU_ABL(64))
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            dpre[r] *= 1.0f - __builtin_amdgcn_exp2f(Hh[T][r] * -144.26950408889634f);
            if (r & 1) din_b = __builtin_amdgcn_mfma_f32_32x32x2f32(w0t[T][r], dpre[r], din_b, 0, 0, 0);
            else din = __builtin_amdgcn_mfma_f32_32x32x2f32(w0t[T][r], dpre[r], din, 0, 0, 0);
          }
#elif defined(DSU_DIN_F32)
          if (!DSU_ABL(64))
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            // softplus'(pre) = sigmoid(100 pre) = 1 - exp(-100 softplus(pre))
            dpre[r] *= 1.0f - __builtin_amdgcn_exp2f(Hh[T][r] * -144.26950408889634f);
            din = __builtin_amdgcn_mfma_f32_32x32x2f32(w0t[T][r], dpre[r], din, 0, 0, 0);
          }
#else
          if (!DSU_ABL(64)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              // softplus'(pre) = sigmoid(100 pre) = 1 - exp(-100 softplus(pre))
              dpre[r] *= 1.0f - __builtin_amdgcn_exp2f(Hh[T][r] * -144.26950408889634f);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              bf16x8 bh, bm;
              bf16_split8(&dpre[8 * g], bh, bm);
              din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bh, din, 0, 0, 0);
              din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bm, din, 0, 0, 0);
              din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_mid[T][g], bh, din, 0, 0, 0);
            }
          }
#endif
          // dPre of this half's points -> LDS rows [point][hidden] for the W0 gradient GEMM
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<float4*>(&sd[l31 * SD_ROW + 32 * T + 8 * qd + 4 * h]) =
                make_float4(dpre[4 * qd], dpre[4 * qd + 1], dpre[4 * qd + 2], dpre[4 * qd + 3]);
        }
#ifdef DSU_DIN_2ACC
#pragma unroll
        for (int r = 0; r < 16; ++r) din[r] += din_b[r];
#endif
        DSU_PROF(4)   // dPre, sigmoid, dIn MFMAs, dPre staging
        if (h == half) {
#pragma unroll
          for (int k4 = 0; k4 < 8; ++k4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * k4 + 0 < KIN) v.x = in[(4 * k4 + 0) < KIN ? 4 * k4 + 0 : 0];
            if (4 * k4 + 1 < KIN) v.y = in[(4 * k4 + 1) < KIN ? 4 * k4 + 1 : 0];
            if (4 * k4 + 2 < KIN) v.z = in[(4 * k4 + 2) < KIN ? 4 * k4 + 2 : 0];
            if (4 * k4 + 3 < KIN) v.w = in[(4 * k4 + 3) < KIN ? 4 * k4 + 3 : 0];
            *reinterpret_cast<float4*>(&sin_[l31 * SIN_ROW + 4 * k4]) = v;
          }
          if (e == 0)
#pragma unroll
          for (int o4 = 0; o4 < 8; ++o4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * o4 + 0 < NOUT) v.x = dout[(4 * o4 + 0) < NOUT ? 4 * o4 + 0 : 0];
            if (4 * o4 + 1 < NOUT) v.y = dout[(4 * o4 + 1) < NOUT ? 4 * o4 + 1 : 0];
            if (4 * o4 + 2 < NOUT) v.z = dout[(4 * o4 + 2) < NOUT ? 4 * o4 + 2 : 0];
            if (4 * o4 + 3 < NOUT) v.w = dout[(4 * o4 + 3) < NOUT ? 4 * o4 + 3 : 0];
            *reinterpret_cast<float4*>(&sdo[l31 * SIN_ROW + 4 * o4]) = v;
          }
        }
        __builtin_amdgcn_wave_barrier();
        DSU_PROF(5)   // input / dOut staging
        // gW0[feat][k'] += sum_points dPre[point][feat] * In'[point][k']
        if (!DSU_ABL(2) && !SPLIT) {
          // the fused form (153 KB of LDS, scatter code resident) has no registers to spare for the
          // read-ahead below (NL = 12 would spill)
#pragma unroll 4
          for (int t = 0; t < 16; ++t) {
            const int pr = 2 * t + h;
            const float b = sin_[pr * SIN_ROW + l31];
            gw0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[pr * SD_ROW + l31], b, gw0[0], 0, 0, 0);
            gw0[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[pr * SD_ROW + 32 + l31], b, gw0[1], 0, 0, 0);
          }
        }
        if (!DSU_ABL(2) && SPLIT) {
          // operands of the next four k-pairs are requested from LDS before the eight MFMAs of the
          // current four are issued (with `#pragma unroll 4` the reads sat right in front of their
          // MFMAs behind `s_waitcnt lgkmcnt(0)`; the same change took texture_bwd from 231 to 169 us)
          float q[2][12];
          auto ld = [&](int tb, float* d) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int pr = 2 * (4 * tb + u) + h;
              d[3 * u + 0] = sd[pr * SD_ROW + l31];
              d[3 * u + 1] = sd[pr * SD_ROW + 32 + l31];
              d[3 * u + 2] = sin_[pr * SIN_ROW + l31];
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
        __builtin_amdgcn_wave_barrier();
        DSU_PROF(6)   // gW0 GEMM
        if (e == 0) {
          // hidden activations of this half's points -> LDS, then gW1[feat][o'] += H^T . dOut
#pragma unroll
          for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              *reinterpret_cast<float4*>(&sd[l31 * SD_ROW + 32 * T + 8 * qd + 4 * h]) =
                  make_float4(Hh[T][4 * qd], Hh[T][4 * qd + 1], Hh[T][4 * qd + 2], Hh[T][4 * qd + 3]);
          __builtin_amdgcn_wave_barrier();
          if (!DSU_ABL(2) && !SPLIT) {
#pragma unroll 4
            for (int t = 0; t < 16; ++t) {
              const int pr = 2 * t + h;
              const float b = sdo[pr * SIN_ROW + l31];
              gw1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[pr * SD_ROW + l31], b, gw1[0], 0, 0, 0);
              gw1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[pr * SD_ROW + 32 + l31], b, gw1[1], 0, 0, 0);
            }
          }
          if (!DSU_ABL(2) && SPLIT) {
            float q[2][12];
            auto ld = [&](int tb, float* d) {
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int pr = 2 * (4 * tb + u) + h;
                d[3 * u + 0] = sd[pr * SD_ROW + l31];
                d[3 * u + 1] = sd[pr * SD_ROW + 32 + l31];
                d[3 * u + 2] = sdo[pr * SIN_ROW + l31];
              }
            };
            ld(0, q[0]);
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
              if (tb + 1 < 4) ld(tb + 1, q[(tb + 1) & 1]);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float* d = q[tb & 1] + 3 * u;
                gw1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[0], d[2], gw1[0], 0, 0, 0);
                gw1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[1], d[2], gw1[1], 0, 0, 0);
              }
            }
          }
        } else {
          // the six offset evaluations only feed output 0: gW1[feat][0] += H[point][feat] * d0
          // as per-lane partial sums over this lane's point column (reduced over lanes at the end)
#pragma unroll
          for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) gw1c0[T][r] = fmaf(Hh[T][r], d[0], gw1c0[T][r]);
        }
        DSU_PROF(7)   // gW1 GEMM / column
        }  // live
        if (ENC && HANDOVER && half == 1 && e + e_step < 7) {
          // the next evaluation's row, requested ~30 k clocks ago; the stores in flight are the
          // first half's (~8 k clocks old)
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            vm_take(rwn[HANDOVER ? l : 0]);
            rw[l] = rwn[HANDOVER ? l : 0];
          }
        }
        if (!live) continue;
        // scatter dIn rows held by this lane: input row i = (r&3) + 8(r>>2) + 4h, levels (i>>1)
        if (SPLIT) {
          // dIn of this half's points -> dinbuf: lane (l31, h) holds the feature pairs of levels
          // {0,1,4,5,8,9} (h = 0) or {2,3,6,7} (h = 1) of point (half, l31)
          // layout [evaluation][level][point]: the 32 lanes of a half write 256 contiguous bytes
          const int64_t pi = wave_first + half * 32 + l31;
          if (pi < r1) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const int lev = ((r & 3) + 8 * (r >> 2)) / 2 + 2 * h;
              if (lev < NL && (uint32_t)lev < active)
                dinbuf[((size_t)e * active + lev) * n + pi] = make_float2(din[r], din[r + 1]);
            }
          }
        } else if (!DSU_ABL(1)) {
          const bool own = h == half;
          const float sx = own ? cx : pcx, sy = own ? cy : pcy, sz = own ? cz : pcz;
          const bool pv = __shfl(valid ? 1 : 0, half * 32 + l31) != 0;
          // The staging area is free from here to the end of this half: it becomes the wave's
          // queue of (table entry, d0, d1) contributions.  Producers are the few "leader" lanes
          // of each level step; the queue is drained 64 contributions per instruction.
          __builtin_amdgcn_wave_barrier();
          int qn = 0;
          auto drain = [&]() {
            __builtin_amdgcn_wave_barrier();
            // 4 queue items per lane per round: their slot claims (returning LDS atomics) are in
            // flight together instead of one dependent claim -> add round trip per 64 items
            for (int i0 = 0; i0 < qn; i0 += 256) {
              uint32_t ent[4], slot[4], old[4];
              float a0[4], a1[4];
              bool on[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u + lane;
                on[u] = i < qn;
                const int ii = on[u] ? i : 0;
                ent[u] = q_ent[ii];
                a0[u] = q_v0[ii];
                a1[u] = q_v1[ii];
                slot[u] = grad_cache_slot(ent[u]);
              }
#pragma unroll
              for (int u = 0; u < 4; ++u)
                old[u] = on[u] ? atomicCAS(&c_keys[slot[u]], GC_EMPTY, ent[u]) : ent[u];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                if (!on[u]) continue;
                if (old[u] == GC_EMPTY || old[u] == ent[u]) {
                  atomicAdd(&c_acc[2 * slot[u]], gc_fix(a0[u]));
                  atomicAdd(&c_acc[2 * slot[u] + 1], gc_fix(a1[u]));
                } else {
                  gc_commit_from(c_keys, c_acc, gtable, ent[u], slot[u], a0[u], a1[u]);
                }
              }
            }
            __builtin_amdgcn_wave_barrier();
            qn = 0;
          };
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int lev = ((r & 3) + 8 * (r >> 2)) / 2 + 2 * h;   // compile-time part + lane half
            const bool lev_on = lev < NL && (uint32_t)lev < active;
            if (((r & 3) + 8 * (r >> 2)) / 2 >= NL) continue;        // no lane has this level
            // level metadata of BOTH lane halves with compile-time indices + a select: a
            // lane-varying index into the kernel-argument struct would be served from scratch
            // memory (a global-memory round trip per lookup at one wave per SIMD)
            const int la = ((r & 3) + 8 * (r >> 2)) / 2;
            const int lb = la + 2 < DSU_MAX_LEVELS ? la + 2 : la;
            const float l_scale = h ? m.scale[lb] : m.scale[la];
            const uint32_t l_off = h ? m.off[lb] : m.off[la];
            const uint32_t l_end = h ? m.off[lb + 1] : m.off[la + 1];
            const uint32_t l_res = h ? m.res[lb] : m.res[la];
            const uint32_t l_hashed = h ? m.hashed[lb] : m.hashed[la];
            bool lead = false;
            float v[16];
            CellPos cp;
            if (lev_on) {
              const bool on = pv;
              const float d0 = on ? din[r] : 0.0f, d1 = on ? din[r + 1] : 0.0f;
              cp = cell_of(l_scale, sx, sy, sz);
              // Neighbouring lanes are neighbouring samples of a ray and mostly sit in the SAME
              // cell: sum the 8x2 corner contributions over runs of equal cells inside each
              // 16-lane row with DPP row shifts (segmented suffix scan); only the first lane of
              // a run (the leader) emits the run's 8 corner contributions.
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(cp, c);
                v[2 * c] = w * d0;
                v[2 * c + 1] = w * d1;
              }
              const int key = (int)((cp.c[0] & 1023u) | ((cp.c[1] & 1023u) << 10) | ((cp.c[2] & 1023u) << 20));
              const int l15 = lane & 15;
              // The neighbour keys are fetched with ALL lanes active: written as
              // `l15 == 15 || dpp(key) != key`, the short-circuit put the DPP move under a reduced
              // EXEC mask, a disabled source lane reads as 0 (bound_ctrl), and lanes 14 / 1 of every
              // row then always saw "different cell": lane 15's share of a run was dropped and
              // lane 1's counted twice whenever they shared the cell of their neighbour.
              const int key_next = dpp_i<0x101>(key), key_prev = dpp_i<0x111>(key);
              int e = ((l15 == 15) | (key_next != key)) ? 1 : 0;   // run ends at this lane
#define DSU_SEG_STEP(CTRL)                                              \
              {                                                         \
                const int eo = dpp_i<CTRL>(e);                          \
                _Pragma("unroll") for (int k = 0; k < 16; ++k) {        \
                  const float vo = dpp_f<CTRL>(v[k]);                   \
                  v[k] += e ? 0.0f : vo;                                \
                }                                                       \
                e |= eo;                                                \
              }
              DSU_SEG_STEP(0x101) DSU_SEG_STEP(0x102) DSU_SEG_STEP(0x104) DSU_SEG_STEP(0x108)
#undef DSU_SEG_STEP
              lead = ((l15 == 0) | (key_prev != key)) && !DSU_ABL(4);  // first of its run
            }
            const unsigned long long bal = __ballot(lead);
            if (lead) {
              const int pos = qn + 8 * __popcll(bal & ((1ull << lane) - 1ull));
              const uint32_t hsize = l_end - l_off;
              uint32_t ent[8];
#pragma unroll
              for (int c = 0; c < 8; ++c)
                ent[c] = l_off + grid_index(l_hashed, hsize, l_res,
                                                 cp.c[0] + (c & 1), cp.c[1] + ((c >> 1) & 1),
                                                 cp.c[2] + ((c >> 2) & 1));
#pragma unroll
              for (int q4 = 0; q4 < 2; ++q4) {
                *reinterpret_cast<uint4*>(&q_ent[pos + 4 * q4]) =
                    make_uint4(ent[4 * q4], ent[4 * q4 + 1], ent[4 * q4 + 2], ent[4 * q4 + 3]);
                *reinterpret_cast<float4*>(&q_v0[pos + 4 * q4]) =
                    make_float4(v[8 * q4], v[8 * q4 + 2], v[8 * q4 + 4], v[8 * q4 + 6]);
                *reinterpret_cast<float4*>(&q_v1[pos + 4 * q4]) =
                    make_float4(v[8 * q4 + 1], v[8 * q4 + 3], v[8 * q4 + 5], v[8 * q4 + 7]);
              }
            }
            qn += 8 * __popcll(bal);
            if (qn + 512 > QCAP) drain();
          }
          drain();
        }
        DSU_PROF(8)   // dIn write-out or scatter
      }
    }
    // flush the gradient cache: one global atomic pair per touched entry, then reset
    if (SPLIT) continue;
    __syncthreads();
    for (int t = threadIdx.x; t < GC_SLOTS; t += blockDim.x) {
      const uint32_t key = c_keys[t];
      if (key != GC_EMPTY) {
        if (!DSU_ABL(8)) {
          unsafeAtomicAdd(gtable + (size_t)key * 2, gc_unfix(c_acc[2 * t]));
          unsafeAtomicAdd(gtable + (size_t)key * 2 + 1, gc_unfix(c_acc[2 * t + 1]));
        }
        c_keys[t] = GC_EMPTY;
        c_acc[2 * t] = 0ull;
        c_acc[2 * t + 1] = 0ull;
      }
    }
    __syncthreads();
    DSU_PROF(9)     // cache flush
  }
  DSU_PROF(10)
  DSU_PROF_END

  // ---- workgroup reduction of the parameter-gradient tiles -> one partial vector per workgroup
  __syncthreads();
  float* red = lds + W1P_F;                     // [4 waves][PART_STRIDE] (fits: 4*4160 floats)
  {
    float* r0 = red + wave * PART_STRIDE;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int feat = feat_of(T, r, h);
        float c0 = gw1c0[T][r];                      // sum over the 32 point columns of this half
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) c0 += __shfl_xor(c0, off);
        r0[PART_GW0 + feat * 32 + l31] = gw0[T][r];
        r0[PART_GW1 + feat * 32 + l31] = gw1[T][r] + (l31 == 0 ? c0 : 0.0f);
      }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) gb1[o] += __shfl_xor(gb1[o], 32);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) gb1[o] += __shfl_xor(gb1[o], off);
    if (lane == 0) {
#pragma unroll
      for (int o = 0; o < NOUT; ++o) r0[PART_GB1 + o] = gb1[o];
    }
  }
  __syncthreads();
  float* part = partials + (size_t)blockIdx.x * PART_STRIDE;
  for (int v = threadIdx.x; v < PART_GB1 + NOUT; v += blockDim.x)
    part[v] = (red[v] + red[PART_STRIDE + v]) + (red[2 * PART_STRIDE + v] + red[3 * PART_STRIDE + v]);
}

// ------------------------------------------------------------ K1, lean / software-pipelined form
// The MLP part of the split backward again, for the case the optimisation runs (NL = 10 levels,
// features from the forward's cache, ACT active levels known at compile time), restructured around
// what the counters of the general kernel above say (profiles/round5_pmc.json): ONE wave per SIMD
// whose instruction stream alternates VALU phases (48 % of the wave cycles: Softplus, sigmoid, bf16
// splits, register shuffles — 22 % of the VALU instructions were v_accvgpr moves of values parked
// in AGPRs) and MFMA phases during which the wave only waits to issue the next MFMA (27 %).
//   * The six OFFSET evaluations (e = 1..6) carry one upstream gradient (on the SDF output) instead
//     of the centre's 13: their own straight-line body without the 13-wide output arrays, the
//     feature-gradient rows, the dOut staging and the gW1 GEMM; the centre evaluation keeps a body
//     of its own.  The live state of the loop that does 6/7 of the work shrinks by ~50 registers.
//   * ACT is a template parameter: layer 0 walks ACT + 2 k-pairs without a branch per level (every
//     MFMA sat in its own basic block), the cache rows are ACT wide.
//   * Inside an offset evaluation the two point halves are software-pipelined: the matrix pipe's
//     long phases run beside the OTHER half's VALU work —
//         L0(half 1) MFMAs            ||  Softplus(half 0)
//         gW0(half 0) MFMAs (32 x 64) ||  Softplus, dPre, sigmoid, bf16 split of half 1
//         gW0(half 1) MFMAs           ||  dIn stores, column-0 sums of gW1
//     The interleave is written out in the source — a piece of VALU work behind every MFMA (pair),
//     fenced with __builtin_amdgcn_sched_barrier(0): left to itself the scheduler issues a phase's
//     MFMAs back to back and the VALU work after them (sched_group_barrier patterns did not change
//     that here), which at one wave per SIMD is the serial form again.  The next evaluation's
//     inputs (cache row hand-over, positions, upstream gradient) are prepared beside the last phase.
// Every value is produced by the same operations in the same order as in sdf_fd_bwd_mfma_kernel
// <NL, true, true>, the tail sharing included.  tests/test_gpu_hashgrid.py compares it with the general
// kernel's cache-less form (no tail sharing there): MLP gradients bit-identical where the ranges are
// whole iterations, equal to float summation order elsewhere.
// gW0 of the OFFSET evaluations as bf16 x 3 on v_mfma_f32_32x32x16_bf16 (K = 16 points per MFMA): ablating
// its 64 f32 MFMAs per evaluation took 46 us off the kernel although they run beside the other half's
// VALU work (profiles/round6_pipe_ablation.txt).  Operands staged TRANSPOSED per wave:
//   dPre^T [hi|mid][64 units][PT_ROW] bf16 — the hi / mid fragments the dIn product splits anyway —,
//   In'^T  [hi|mid][32 input columns][PT_ROW] (rows nobody writes stay zero for the launch),
// PT_ROW = 40 (80-byte rows: the 16-byte fragments of 16 lanes fall on 16 different bank quads).
// 12 MFMAs of 32 clocks per point half instead of 32 of 64.  (-DDSU_PIPE_GW0_F32: the f32 form, A/B.)
constexpr int PT_ROW = 40;
constexpr int PT_IMGD = 2 * 64 * PT_ROW, PT_IMGI = 2 * 32 * PT_ROW;      // bf16 elements
constexpr int PT_IMGO = 2 * 32 * PT_ROW;                                  // dOut^T of the centre evaluation (gW1)
#ifdef DSU_PIPE_GW0_F32
constexpr int PIPE_IMG_F = 0;
#else
constexpr int PIPE_IMG_F = (PT_IMGD + PT_IMGI + PT_IMGO) / 2;             // floats per wave (5120)
#endif
constexpr int PIPE_LDS_F = W1P_F + 4 * STAGE_F + 4 * PIPE_IMG_F;

template <int NL, int ACT>
__global__ __launch_bounds__(256) void sdf_fd_bwd_pipe_kernel(
    dsu_sdf_mlp mlp, const float* __restrict__ pts, int64_t n, float radius, float eps, float eps2,
    const float* __restrict__ d_sdf, const float* __restrict__ d_grad,
    const float* __restrict__ d_feature, const float* __restrict__ d_laplace,
    float* __restrict__ partials, const __half2* __restrict__ enc, float2* __restrict__ dinbuf,
    const int32_t* __restrict__ perm) {
  static_assert(ACT >= 1 && ACT <= NL && 2 * NL + 4 <= 32, "layout");
  constexpr int KPA = ACT + 2;             // k-pairs walked by layer 0: ACT levels, xyz, (z, 1)
  constexpr int FG = (2 * ACT + 3) / 4;    // float4 groups of feature columns staged per point
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* w1perm = lds;
  float* b1s = lds + 2 * NOUT * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  float* sd = lds + W1P_F + wave * STAGE_F;
  float* sin_ = sd + 32 * SD_ROW;
  float* sdo = sin_ + 32 * SIN_ROW;

  // the scatter kernel's work counter (behind the dIn buffer) starts every launch at zero
  if (blockIdx.x == 0 && threadIdx.x == 0)
    *reinterpret_cast<int*>(dinbuf + (size_t)7 * (size_t)n * NL) = 0;

  // ---- weights.  k-pair tt of layer 0 is the permuted input pair t = tt (level tt) or NL + tt - ACT
  float w0a[2][KPA];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int tt = 0; tt < KPA; ++tt) {
      const int t = tt < ACT ? tt : NL + (tt - ACT);
      w0a[T][tt] = w0p<NL>(mlp, 32 * T + l31, 2 * t + h);
    }
  // Layer 0 of the recompute as bf16 x 3 when its 2 KPA inputs fit one 16-deep MFMA (ACT <= 6, the
  // optimisation's schedule): an ablation priced the 2 x KPA f32 MFMAs per point half at 51 us of the kernel.
  // A fragments: W0'[unit 32 T + l31][input 8 h + t] hi / mid (inputs in the order of `in`).
#ifdef DSU_PIPE_L0_F32
  constexpr bool L0BF = false;
#else
  constexpr bool L0BF = 2 * KPA <= 16;
#endif
  bf16x8 w0b_hi[2], w0b_mid[2];
  if constexpr (L0BF) {
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      float w[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int j = 8 * h + t, tt = j >> 1;
        const bool real = j < 2 * KPA;
        const int t_ = tt < ACT ? tt : NL + (tt - ACT);
        const float v = w0p<NL>(mlp, 32 * T + l31, real ? 2 * t_ + (j & 1) : 0);
        w[t] = real ? v : 0.0f;
      }
      bf16_split8(w, w0b_hi[T], w0b_mid[T]);
    }
  }
  // the centre evaluation's dPre = W1^T . dOut (13 upstream gradients per point) as bf16 x 3 as well:
  // A fragments W1[o = 8 h + t][unit 32 T + l31] hi / mid (outputs 13..15: zero)
#ifdef DSU_PIPE_DPRE_VALU
  constexpr bool DPBF = false;
#else
  constexpr bool DPBF = true;
#endif
  bf16x8 w1t_hi[2], w1t_mid[2];
  if constexpr (DPBF) {
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      float w[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int o = 8 * h + t;
        const float v = mlp.w1[(o < NOUT ? o : 0) * 64 + 32 * T + l31];
        w[t] = o < NOUT ? v : 0.0f;
      }
      bf16_split8(w, w1t_hi[T], w1t_mid[T]);
    }
  }
  float w1o0[2][16];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) w1o0[T][r] = mlp.w1[feat_of(T, r, h)];
  bf16x8 w0t_hi[2][2], w0t_mid[2][2];
  {
    float w0t[2][16];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        w0t[T][r] = l31 < MC<NL>::KIN ? w0p<NL>(mlp, feat_of(T, r, h), l31) : 0.0f;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int g = 0; g < 2; ++g) bf16_split8(&w0t[T][8 * g], w0t_hi[T][g], w0t_mid[T][g]);
  }
  load_w1perm(w1perm, b1s, mlp);
  // In' and dOut rows: the columns nobody writes (masked levels, padding) stay zero for the launch
  for (int t = lane; t < 2 * 32 * SIN_ROW; t += 64) sin_[t] = 0.0f;
#ifndef DSU_PIPE_GW0_F32
  __bf16* imgD = reinterpret_cast<__bf16*>(lds + W1P_F + 4 * STAGE_F + wave * PIPE_IMG_F);
  __bf16* imgI = imgD + PT_IMGD;
  __bf16* imgO = imgI + PT_IMGI;
  for (int t = lane; t < (PT_IMGI + PT_IMGO) / 2; t += 64) reinterpret_cast<uint32_t*>(imgI)[t] = 0u;
#endif
  __syncthreads();

  f32x16 gw0[2], gw1[2];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) gw0[T][r] = gw1[T][r] = 0.0f;
  float gb1[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) gb1[o] = 0.0f;
  float gw1c0[2][16];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) gw1c0[T][r] = 0.0f;

  // ---- pieces shared by the two evaluation bodies
  // B operands of layer 0 for both point halves from the lane's own inputs
  auto l0_half = [&](const float (&b)[KPA], f32x16 (&acc)[2]) {
#pragma unroll
    for (int T = 0; T < 2; ++T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[T][r] = 0.0f;
#pragma unroll
      for (int tt = 0; tt < KPA; ++tt)
#if defined(DSU_PIPE_ABL) && (DSU_PIPE_ABL & 16)
        acc[T][tt & 15] += w0a[T][tt] * b[tt];        // (timing ablation: no layer-0 MFMAs)
#else
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0a[T][tt], b[tt], acc[T], 0, 0, 0);
#endif
    }
  };
  // B fragments of layer 0 for both point halves from the lane's own inputs: packed bf16 hi / mid pairs,
  // one permlane32 swap per dword (the lane of half h' holds inputs 8 h' .. 8 h' + 7 of the column's point)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  auto pack_operands = [&](const float* in, auto nc, bf16x8& bh0, bf16x8& bm0, bf16x8& bh1, bf16x8& bm1) {
    constexpr int NV = decltype(nc)::value;            // values per point (<= 16)
    uint32_t ph[8], pm[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float x0 = 2 * q < NV ? in[2 * q < NV ? 2 * q : 0] : 0.0f;
      const float x1 = 2 * q + 1 < NV ? in[2 * q + 1 < NV ? 2 * q + 1 : 0] : 0.0f;
      const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
      const __bf16 m0 = (__bf16)(x0 - (float)h0), m1 = (__bf16)(x1 - (float)h1);
      ph[q] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
      pm[q] = (uint32_t)__builtin_bit_cast(uint16_t, m0) | ((uint32_t)__builtin_bit_cast(uint16_t, m1) << 16);
    }
    u32x4 a0, a1, c0, c1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      auto r = __builtin_amdgcn_permlane32_swap(ph[q], ph[q + 4], false, false);
      a0[q] = r[0]; a1[q] = r[1];
      auto r2 = __builtin_amdgcn_permlane32_swap(pm[q], pm[q + 4], false, false);
      c0[q] = r2[0]; c1[q] = r2[1];
    }
    bh0 = __builtin_bit_cast(bf16x8, a0); bh1 = __builtin_bit_cast(bf16x8, a1);
    bm0 = __builtin_bit_cast(bf16x8, c0); bm1 = __builtin_bit_cast(bf16x8, c1);
  };
  auto l0_bf = [&](const bf16x8& bh, const bf16x8& bm, f32x16 (&acc)[2]) {
#pragma unroll
    for (int T = 0; T < 2; ++T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[T][r] = 0.0f;
      acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b_hi[T], bh, acc[T], 0, 0, 0);
      acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b_hi[T], bm, acc[T], 0, 0, 0);
      acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0b_mid[T], bh, acc[T], 0, 0, 0);
    }
  };
  auto softplus2 = [&](f32x16 (&acc)[2]) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; r += 2) softplus_inplace2(acc[T], r);
  };
  // dPre -> sigmoid factor -> dIn MFMAs of one hidden tile (bf16 x 3, as in the general kernel)
  auto din_tile = [&](int T, float (&dpre)[16], const f32x16& H, f32x16& din) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dpre[r] *= 1.0f - __builtin_amdgcn_exp2f(H[r] * -144.26950408889634f);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8 bh, bm;
      bf16_split8(&dpre[8 * g], bh, bm);
      din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bh, din, 0, 0, 0);
      din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bm, din, 0, 0, 0);
      din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_mid[T][g], bh, din, 0, 0, 0);
    }
  };
  auto stage_rows = [&](float* dst, int T, const float* v /*16*/) {   // [point][hidden] rows
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      *reinterpret_cast<float4*>(&dst[l31 * SD_ROW + 32 * T + 8 * qd + 4 * h]) =
          make_float4(v[4 * qd], v[4 * qd + 1], v[4 * qd + 2], v[4 * qd + 3]);
  };
  auto stage_in = [&](const float (&in)[2 * KPA]) {                   // own-half lanes: In' row
#pragma unroll
    for (int g = 0; g < FG; ++g) {
      float4 v;
      v.x = in[4 * g];
      v.y = in[4 * g + 1];
      v.z = 4 * g + 2 < 2 * ACT ? in[(4 * g + 2) < 2 * ACT ? 4 * g + 2 : 0] : 0.0f;
      v.w = 4 * g + 3 < 2 * ACT ? in[(4 * g + 3) < 2 * ACT ? 4 * g + 3 : 0] : 0.0f;
      *reinterpret_cast<float4*>(&sin_[l31 * SIN_ROW + 4 * g]) = v;
    }
    *reinterpret_cast<float4*>(&sin_[l31 * SIN_ROW + 2 * NL]) =
        make_float4(in[2 * ACT], in[2 * ACT + 1], in[2 * ACT + 2], in[2 * ACT + 3]);
  };
  // contraction over the 32 staged points: acc[T][feat][col] += rowsA[point][feat] * rowsB[point][col]
  auto gemm_points = [&](f32x16 (&acc)[2], const float* rowsB) {
    float q[2][12];
    auto ld = [&](int tb, float* d) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pr = 2 * (4 * tb + u) + h;
        d[3 * u + 0] = sd[pr * SD_ROW + l31];
        d[3 * u + 1] = sd[pr * SD_ROW + 32 + l31];
        d[3 * u + 2] = rowsB[pr * SIN_ROW + l31];
      }
    };
    ld(0, q[0]);
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      if (tb + 1 < 4) ld(tb + 1, q[(tb + 1) & 1]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* d = q[tb & 1] + 3 * u;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[0], d[2], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[1], d[2], acc[1], 0, 0, 0);
      }
    }
  };
  // the same contraction with a piece of other work behind the two MFMAs of every point pair
  // (filler(integral_constant<k>), k = 0..15), fenced so that the interleave stays as written: at
  // one wave per SIMD an MFMA phase issued back to back is time the wave only waits for the pipe
  auto gemm_points_with = [&](f32x16 (&acc)[2], const float* rowsB, auto&& filler) {
    float q[2][12];
    auto ld = [&](int tb, float* d) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pr = 2 * (4 * tb + u) + h;
        d[3 * u + 0] = sd[pr * SD_ROW + l31];
        d[3 * u + 1] = sd[pr * SD_ROW + 32 + l31];
        d[3 * u + 2] = rowsB[pr * SIN_ROW + l31];
      }
    };
    ld(0, q[0]);
    auto step = [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      constexpr int tb = K / 4, u = K % 4;
      if (u == 0 && tb + 1 < 4) ld(tb + 1, q[(tb + 1) & 1]);
      const float* d = q[tb & 1] + 3 * u;
#if !(defined(DSU_PIPE_ABL) && (DSU_PIPE_ABL & 8))
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[0], d[2], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[1], d[2], acc[1], 0, 0, 0);
#else
      acc[0][K & 15] += d[0] * d[2];                 // (timing ablation: the staged operands stay live)
      acc[1][K & 15] += d[1] * d[2];
#endif
      filler(kc);
      __builtin_amdgcn_sched_barrier(0);
    };
    step(std::integral_constant<int, 0>{});  step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});  step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});  step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});  step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});  step(std::integral_constant<int, 9>{});
    step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
    step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
  };
  auto store_din = [&](int e, const f32x16& din, int64_t pi, int64_t r1) {
#if defined(DSU_PIPE_ABL) && (DSU_PIPE_ABL & 2)
    if (pi < r1 && din[0] == 12345.678f) {        // (timing ablation: no dIn stores)
#else
    if (pi < r1) {
#endif
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int la = ((r & 3) + 8 * (r >> 2)) / 2;      // this register pair's level for h = 0
        const int lev = la + 2 * h;
        if (la < ACT) {                                   // (else: no lane holds an active level here)
          if (lev < ACT) dinbuf[((size_t)e * ACT + lev) * n + pi] = make_float2(din[r], din[r + 1]);
        }
      }
    }
  };

#ifndef DSU_PIPE_GW0_F32
  // hi / mid fragments of dPre (registers 8 g + t of tile T = units 32 T + (t & 3) + 8 (2 g + (t >> 2)) + 4 h
  // of the point in column l31) -> the transposed image, two-byte stores at compile-time offsets
  auto stage_dpre_T = [&](int T, int g, const bf16x8& bh, const bf16x8& bm) {
    __bf16* base = imgD + 4 * h * PT_ROW + l31;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int u = 32 * T + (t & 3) + 8 * (2 * g + (t >> 2));
      base[u * PT_ROW] = bh[t];
      base[(64 + u) * PT_ROW] = bm[t];
    }
  };
  // the lane's own inputs -> rows (input columns) of In'^T: features at their column, (x, y, z, 1) at 2 NL ..
  auto stage_in_T = [&](const float (&in)[2 * KPA]) {
    __bf16* base = imgI + l31;
#pragma unroll
    for (int j = 0; j < 2 * KPA; ++j) {
      const int row = j < 2 * ACT ? j : 2 * NL + (j - 2 * ACT);
      const __bf16 hi = (__bf16)in[j];
      base[row * PT_ROW] = hi;
      base[(32 + row) * PT_ROW] = (__bf16)(in[j] - (float)hi);
    }
  };
  // acc[Ti][unit][column] += dPre^T . In' over the 32 staged points; a quarter of the filler's 16 pieces
  // behind each group of three MFMAs, fenced as in gemm_points_with
  auto gemm_T_with = [&](f32x16 (&acc)[2], const __bf16* imgB, auto&& filler) {
    auto frag = [&](const __bf16* img, int row, int ks) {
      return *reinterpret_cast<const bf16x8*>(img + row * PT_ROW + 16 * ks + 8 * h);
    };
    auto step = [&](auto sc) {
      constexpr int S = decltype(sc)::value;
      constexpr int ks = S >> 1, Ti = S & 1;
      const bf16x8 bh = frag(imgB, l31, ks), bm = frag(imgB + 32 * PT_ROW, l31, ks);
      const bf16x8 ah = frag(imgD, 32 * Ti + l31, ks), am = frag(imgD + 64 * PT_ROW, 32 * Ti + l31, ks);
      acc[Ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[Ti], 0, 0, 0);
      acc[Ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[Ti], 0, 0, 0);
      acc[Ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[Ti], 0, 0, 0);
      filler(std::integral_constant<int, 4 * S + 0>{});
      filler(std::integral_constant<int, 4 * S + 1>{});
      filler(std::integral_constant<int, 4 * S + 2>{});
      filler(std::integral_constant<int, 4 * S + 3>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
  };
#endif
  const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 31) / 32 * 32;
  const int64_t r0 = blockIdx.x * per;
  const int64_t r1 = r0 + per < n ? r0 + per : n;
#if defined(DSU_PIPE_ABL) && (DSU_PIPE_ABL & 1)   // (timing ablation, variant build: prologue + epilogue only)
  for (int64_t bbase = r1; bbase < r1; bbase += blockDim.x) {
#else
  for (int64_t bbase = r0; bbase < r1; bbase += blockDim.x) {
#endif
    // (the last, partial iteration of a range is shared by the waves as in the general kernel)
    int blk = wave, e_start = 0, e_step = 1;
    {
      const int64_t left = r1 - bbase;
      if (left <= 64) { blk = 0; e_start = wave; e_step = 4; }
      else if (left <= 128) { blk = wave & 1; e_start = wave >> 1; e_step = 2; }
    }
    const int64_t wave_first = bbase + blk * 64;
    if (wave_first >= r1) continue;                          // nothing for this wave (no barriers inside)
    const int64_t i = wave_first + lane;
    const bool valid = i < r1;
    const int64_t ii = valid ? i : r1 - 1;
    const bool live1 = wave_first + 32 < r1;                 // the second point half holds points
    const float p[3] = {pts[ii * 3], pts[ii * 3 + 1], pts[ii * 3 + 2]};
    const int64_t gi = perm ? (int64_t)perm[ii] : ii;
    float ds = 0.f, dl = 0.f, dg[3] = {0.f, 0.f, 0.f};
    if (valid) {
      if (d_sdf) ds = d_sdf[gi];
      if (d_laplace) dl = d_laplace[gi];
      if (d_grad) { dg[0] = d_grad[gi * 3]; dg[1] = d_grad[gi * 3 + 1]; dg[2] = d_grad[gi * 3 + 2]; }
    }
    __half2 rw[ACT], rwn[ACT];
    {
      const __half2* row = enc + ((size_t)e_start * n + ii) * ACT;
#pragma unroll
      for (int l = 0; l < ACT; ++l) rw[l] = row[l];
#pragma unroll
      for (int l = 0; l < ACT; ++l) rwn[l] = rw[l];
    }
    // inputs of evaluation e from the row in `rw`; requests the row of evaluation e + e_step
    auto prologue = [&](int e, float (&in)[2 * KPA]) {
      float q[3];
      fd_point(p, e, eps, radius, q);
      const float cx = contract(q[0], radius), cy = contract(q[1], radius), cz = contract(q[2], radius);
#pragma unroll
      for (int l = 0; l < ACT; ++l) {
        vm_take(rw[l]);
        const float2 f = __half22float2(rw[l]);
        in[2 * l] = f.x;
        in[2 * l + 1] = f.y;
      }
      if (e + e_step < 7) {
        const __half2* row = enc + ((size_t)(e + e_step) * n + ii) * ACT;
#pragma unroll
        for (int l = 0; l < ACT; ++l) rwn[l] = row[l];
      }
      in[2 * ACT + 0] = cx * 2.0f + -1.0f;
      in[2 * ACT + 1] = cy * 2.0f + -1.0f;
      in[2 * ACT + 2] = cz * 2.0f + -1.0f;
      in[2 * ACT + 3] = 1.0f;
    };
    auto handover = [&]() {
#pragma unroll
      for (int l = 0; l < ACT; ++l) {
        vm_take(rwn[l]);
        rw[l] = rwn[l];
      }
    };

    int e = e_start;
#if defined(DSU_PIPE_ABL) && (DSU_PIPE_ABL & 32)
    if (e == 0) { if (e + e_step < 7) handover(); e += e_step; }   // (timing ablation: no centre evaluation)
#endif
    if (e == 0) {
      // ------------------------------------------------ centre evaluation: 13 upstream gradients
      float in[2 * KPA];
      prologue(0, in);
      float dout[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) dout[o] = 0.0f;
      if (valid) {
        if (d_feature) {
#pragma unroll
          for (int o = 0; o < NOUT; ++o) dout[o] = d_feature[gi * NOUT + o];
        }
        dout[0] += ds - 6.0f * dl / eps2;
      }
#pragma unroll
      for (int o = 0; o < NOUT; ++o) gb1[o] += dout[o];
      float b0[KPA], b1[KPA];
      bf16x8 cbh0, cbm0, cbh1, cbm1;
      if constexpr (L0BF) {
        pack_operands(in, std::integral_constant<int, (2 * KPA <= 16 ? 2 * KPA : 16)>{}, cbh0, cbm0, cbh1, cbm1);
      } else {
#pragma unroll
        for (int tt = 0; tt < KPA; ++tt) swap_halves(in[2 * tt], in[2 * tt + 1], b0[tt], b1[tt]);
      }
      bf16x8 dbh0, dbm0, dbh1, dbm1;           // dOut of the column's point, both halves (B fragments)
      if constexpr (DPBF) pack_operands(dout, std::integral_constant<int, NOUT>{}, dbh0, dbm0, dbh1, dbm1);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half == 1 && !live1) break;
        float d[NOUT];
        if constexpr (!DPBF) {
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
            const float other = partner32(dout[o], h);
            d[o] = (h == half) ? dout[o] : other;
          }
        }
        f32x16 Hh[2];
        if constexpr (L0BF) l0_bf(half ? cbh1 : cbh0, half ? cbm1 : cbm0, Hh);
        else l0_half(half ? b1 : b0, Hh);
        softplus2(Hh);
        f32x16 din;
#pragma unroll
        for (int r = 0; r < 16; ++r) din[r] = 0.0f;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          float dpre[16];
          if constexpr (DPBF) {
            f32x16 dacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[r] = 0.0f;
            const bf16x8 bh = half ? dbh1 : dbh0, bm = half ? dbm1 : dbm0;
            dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1t_hi[T], bh, dacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1t_hi[T], bm, dacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1t_mid[T], bh, dacc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) dpre[r] = dacc[r];
          } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) dpre[r] = w1o0[T][r] * d[0];
          const float* wp = w1perm + h * NOUT * 32 + T * 16;
#ifdef DSU_PIPE_CENTRE_UNROLL
#pragma unroll DSU_PIPE_CENTRE_UNROLL
#else
#pragma unroll       // (fully unrolled: the 48 weight reads of a tile are in flight together: -2.4 us; by 3: +7 us)
#endif
          for (int o = 1; o < NOUT; ++o) {
            const float4* w4 = reinterpret_cast<const float4*>(wp + o * 32);
            const float dv = d[o];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const float4 w = w4[qd];
              dpre[4 * qd + 0] = fmaf(w.x, dv, dpre[4 * qd + 0]);
              dpre[4 * qd + 1] = fmaf(w.y, dv, dpre[4 * qd + 1]);
              dpre[4 * qd + 2] = fmaf(w.z, dv, dpre[4 * qd + 2]);
              dpre[4 * qd + 3] = fmaf(w.w, dv, dpre[4 * qd + 3]);
            }
          }
          }
#ifdef DSU_PIPE_GW0_F32
          din_tile(T, dpre, Hh[T], din);
          stage_rows(sd, T, dpre);
#else
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dpre[r] *= 1.0f - __builtin_amdgcn_exp2f(Hh[T][r] * -144.26950408889634f);
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            bf16x8 bh, bm;
            bf16_split8(&dpre[8 * g], bh, bm);
            din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bh, din, 0, 0, 0);
            din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bm, din, 0, 0, 0);
            din = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_mid[T][g], bh, din, 0, 0, 0);
            stage_dpre_T(T, g, bh, bm);
          }
#endif
        }
#ifndef DSU_PIPE_GW0_F32
        // the centre's two contractions over the points as bf16 x 3 as well: gW0 from dPre^T / In'^T, then
        // gW1[unit][o'] from the hidden activations (split here) and dOut^T (13 rows of a 32-row image)
        if (h == half) {
          stage_in_T(in);
#pragma unroll
          for (int o = 0; o < NOUT; ++o) {
            const __bf16 hi = (__bf16)dout[o];
            imgO[o * PT_ROW + l31] = hi;
            imgO[(32 + o) * PT_ROW + l31] = (__bf16)(dout[o] - (float)hi);
          }
        }
        __builtin_amdgcn_wave_barrier();
        gemm_T_with(gw0, imgI, [](auto) {});
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            float hv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) hv[t] = Hh[T][8 * g + t];
            bf16x8 bh, bm;
            bf16_split8(hv, bh, bm);
            stage_dpre_T(T, g, bh, bm);
          }
        __builtin_amdgcn_wave_barrier();
        gemm_T_with(gw1, imgO, [](auto) {});
        __builtin_amdgcn_wave_barrier();
        store_din(0, din, wave_first + half * 32 + l31, r1);
      }
#else
        if (h == half) {
          stage_in(in);
#pragma unroll
          for (int o4 = 0; o4 < 4; ++o4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * o4 + 0 < NOUT) v.x = dout[(4 * o4 + 0) < NOUT ? 4 * o4 + 0 : 0];
            if (4 * o4 + 1 < NOUT) v.y = dout[(4 * o4 + 1) < NOUT ? 4 * o4 + 1 : 0];
            if (4 * o4 + 2 < NOUT) v.z = dout[(4 * o4 + 2) < NOUT ? 4 * o4 + 2 : 0];
            if (4 * o4 + 3 < NOUT) v.w = dout[(4 * o4 + 3) < NOUT ? 4 * o4 + 3 : 0];
            *reinterpret_cast<float4*>(&sdo[l31 * SIN_ROW + 4 * o4]) = v;
          }
        }
        __builtin_amdgcn_wave_barrier();
        gemm_points(gw0, sin_);
        __builtin_amdgcn_wave_barrier();
        // hidden activations of this half's points -> LDS, then gW1[feat][o'] += H^T . dOut
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          float hv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) hv[r] = Hh[T][r];
          stage_rows(sd, T, hv);
        }
        __builtin_amdgcn_wave_barrier();
        gemm_points(gw1, sdo);
        __builtin_amdgcn_wave_barrier();
        store_din(0, din, wave_first + half * 32 + l31, r1);
      }
#endif
      if (e + e_step < 7) handover();
      e += e_step;
    }
    // ------------------------------------------------ offset evaluations: one upstream gradient
    // loop-carried: the inputs and the upstream gradient of the evaluation about to run, prepared
    // beside the previous evaluation's last MFMA phase
    float in[2 * KPA];
    float dout0 = 0.0f;
    float q3[3] = {0.f, 0.f, 0.f};
    auto prep_unpack = [&]() {                              // rw (already handed over) -> feature inputs
#pragma unroll
      for (int l = 0; l < ACT; ++l) {
        const float2 f = __half22float2(rw[l]);
        in[2 * l] = f.x;
        in[2 * l + 1] = f.y;
      }
    };
    auto prep_point = [&](int en) {                         // position of evaluation en -> xyz inputs
      fd_point(p, en, eps, radius, q3);
      in[2 * ACT + 0] = contract(q3[0], radius) * 2.0f + -1.0f;
      in[2 * ACT + 1] = contract(q3[1], radius) * 2.0f + -1.0f;
      in[2 * ACT + 2] = contract(q3[2], radius) * 2.0f + -1.0f;
      in[2 * ACT + 3] = 1.0f;
    };
    auto prep_request = [&](int en) {                       // row of the evaluation after en (clamped)
      const int e2 = en + e_step < 7 ? en + e_step : 6;
      const __half2* row = enc + ((size_t)e2 * n + ii) * ACT;
#pragma unroll
      for (int l = 0; l < ACT; ++l) rwn[l] = row[l];
    };
    auto prep_dout = [&](int en) {
      const int ax = (en - 1) >> 1;
      const float sgn = ((en - 1) & 1) ? -1.0f : 1.0f;
      const float dga = ax == 0 ? dg[0] : (ax == 1 ? dg[1] : dg[2]);
      dout0 = valid ? sgn * 0.5f * dga / eps + dl / eps2 : 0.0f;
    };
    if (e < 7) {
#pragma unroll
      for (int l = 0; l < ACT; ++l) vm_take(rw[l]);
      prep_unpack();
      prep_point(e);
      prep_request(e);
      prep_dout(e);
    }
#pragma unroll 1
    for (; e < 7; e += e_step) {
      gb1[0] += dout0;
      const float other0 = partner32(dout0, h);
      const float d0h0 = h == 0 ? dout0 : other0;       // gradient of the point this lane column holds
      const float d0h1 = h == 1 ? dout0 : other0;       //   in half 0 / half 1
      float b0[KPA], b1[KPA];
      bf16x8 obh0, obm0, obh1, obm1;
      if constexpr (L0BF) {
        pack_operands(in, std::integral_constant<int, (2 * KPA <= 16 ? 2 * KPA : 16)>{}, obh0, obm0, obh1, obm1);
      } else {
#pragma unroll
        for (int tt = 0; tt < KPA; ++tt) swap_halves(in[2 * tt], in[2 * tt + 1], b0[tt], b1[tt]);
      }
      f32x16 H0[2], H1[2], din0, din1;
      float dp0[2][16], dp1[2][16];
      // P1: layer 0 of half 0
      if constexpr (L0BF) l0_bf(obh0, obm0, H0);
      else l0_half(b0, H0);
      __builtin_amdgcn_sched_barrier(0);
      // P2: layer 0 of half 1  ||  Softplus of half 0 (+ its column-0 sums of gW1), a few hidden units
      // behind every MFMA; the fences keep the interleave the source spells out
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) H1[T][r] = 0.0f;
      if constexpr (L0BF) {
#pragma unroll
        for (int mi = 0; mi < 6; ++mi) {
          constexpr int NM = 6;
          const int T = mi / 3, term = mi % 3;
          H1[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 2 ? w0b_mid[T] : w0b_hi[T],
                                                         term == 1 ? obm1 : obh1, H1[T], 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 32; v += 2)
            if (v * NM / 32 == mi) {
              softplus_inplace2(H0[v >> 4], v & 15);
              gw1c0[v >> 4][v & 15] = fmaf(H0[v >> 4][v & 15], d0h0, gw1c0[v >> 4][v & 15]);
              gw1c0[v >> 4][(v & 15) + 1] = fmaf(H0[v >> 4][(v & 15) + 1], d0h0, gw1c0[v >> 4][(v & 15) + 1]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int mi = 0; mi < 2 * KPA; ++mi) {
        constexpr int NM = 2 * KPA;
        const int T = mi / KPA, tt = mi % KPA;
#if defined(DSU_PIPE_ABL) && (DSU_PIPE_ABL & 16)
        H1[T][tt & 15] += w0a[T][tt] * b1[tt];
#else
        H1[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0a[T][tt], b1[tt], H1[T], 0, 0, 0);
#endif
#pragma unroll
        for (int v = 0; v < 32; v += 2)
          if (v * NM / 32 == mi) {
            softplus_inplace2(H0[v >> 4], v & 15);
            gw1c0[v >> 4][v & 15] = fmaf(H0[v >> 4][v & 15], d0h0, gw1c0[v >> 4][v & 15]);
            gw1c0[v >> 4][(v & 15) + 1] = fmaf(H0[v >> 4][(v & 15) + 1], d0h0, gw1c0[v >> 4][(v & 15) + 1]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      // P3: dPre / sigmoid / dIn of half 0, rows to LDS
#pragma unroll
      for (int r = 0; r < 16; ++r) din0[r] = 0.0f;
#pragma unroll
      for (int T = 0; T < 2; ++T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dp0[T][r] = w1o0[T][r] * d0h0;
#ifdef DSU_PIPE_GW0_F32
        din_tile(T, dp0[T], H0[T], din0);
        stage_rows(sd, T, dp0[T]);
#else
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dp0[T][r] *= 1.0f - __builtin_amdgcn_exp2f(H0[T][r] * -144.26950408889634f);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          bf16x8 bh, bm;
          bf16_split8(&dp0[T][8 * g], bh, bm);
          din0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bh, din0, 0, 0, 0);
          din0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bm, din0, 0, 0, 0);
          din0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_mid[T][g], bh, din0, 0, 0, 0);
          stage_dpre_T(T, g, bh, bm);
        }
#endif
      }
#ifdef DSU_PIPE_GW0_F32
      if (h == 0) stage_in(in);
#else
      if (h == 0) stage_in_T(in);
#endif
      __builtin_amdgcn_sched_barrier(0);
      // P4: gW0 over half 0's points  ||  Softplus, column-0 sums, dPre x sigmoid of half 1: two
      // hidden units behind the two MFMAs of every point pair
#ifdef DSU_PIPE_GW0_F32
      gemm_points_with(gw0, sin_, [&](auto kc) {
#else
      gemm_T_with(gw0, imgI, [&](auto kc) {
#endif
        constexpr int K = decltype(kc)::value;
        softplus_inplace2(H1[(2 * K) >> 4], (2 * K) & 15);
#pragma unroll
        for (int v = 2 * K; v < 2 * K + 2; ++v) {
          const int T = v >> 4, r = v & 15;
          gw1c0[T][r] = fmaf(H1[T][r], d0h1, gw1c0[T][r]);
          dp1[T][r] = (w1o0[T][r] * d0h1) *
                      (1.0f - __builtin_amdgcn_exp2f(H1[T][r] * -144.26950408889634f));
        }
      });
      // P5: dIn of half 1; half 0's dIn out; half 1's rows to LDS (behind the reads of P4)
#pragma unroll
      for (int r = 0; r < 16; ++r) din1[r] = 0.0f;
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          bf16x8 bh, bm;
          bf16_split8(&dp1[T][8 * g], bh, bm);
          din1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bh, din1, 0, 0, 0);
          din1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_hi[T][g], bm, din1, 0, 0, 0);
          din1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0t_mid[T][g], bh, din1, 0, 0, 0);
#ifndef DSU_PIPE_GW0_F32
          if (live1) stage_dpre_T(T, g, bh, bm);          // (behind P4's reads of the image: in-order LDS queue)
#endif
        }
      store_din(e, din0, wave_first + l31, r1);
      const int en = e + e_step < 7 ? e + e_step : 6;    // next evaluation (clamped: unused at the end)
      if (live1) {
#ifdef DSU_PIPE_GW0_F32
#pragma unroll
        for (int T = 0; T < 2; ++T) stage_rows(sd, T, dp1[T]);
        if (h == 1) stage_in(in);
#else
        if (h == 1) stage_in_T(in);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // P6: gW0 over half 1's points  ||  half 1's dIn out, the next evaluation's inputs
#ifdef DSU_PIPE_GW0_F32
        gemm_points_with(gw0, sin_, [&](auto kc) {
#else
        gemm_T_with(gw0, imgI, [&](auto kc) {
#endif
          constexpr int K = decltype(kc)::value;
          if (K == 1) store_din(e, din1, wave_first + 32 + l31, r1);
          if (K == 3) {
#pragma unroll
            for (int l = 0; l < ACT; ++l) {
              vm_take(rwn[l]);
              rw[l] = rwn[l];
            }
          }
          if (K == 5) prep_unpack();
          if (K == 7) prep_point(en);
          if (K == 9) prep_request(en);
          if (K == 11) prep_dout(en);
        });
      } else {
#pragma unroll
        for (int l = 0; l < ACT; ++l) {
          vm_take(rwn[l]);
          rw[l] = rwn[l];
        }
        prep_unpack();
        prep_point(en);
        prep_request(en);
        prep_dout(en);
      }
    }
  }

#if defined(DSU_PIPE_ABL) && (DSU_PIPE_ABL & 4)
  partials[(size_t)blockIdx.x * PART_STRIDE + threadIdx.x] =      // (timing ablation: no reduction)
      gw0[0][0] + gw0[1][1] + gw1[0][2] + gw1[1][3] + gw1c0[0][4] + gw1c0[1][5] + gb1[0];
  return;
#endif
  // ---- workgroup reduction of the parameter-gradient tiles -> one partial vector per workgroup
  __syncthreads();
  float* red = lds + W1P_F;                     // [4 waves][PART_STRIDE] (fits: 4*4160 floats)
  {
    float* rr = red + wave * PART_STRIDE;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int feat = feat_of(T, r, h);
        float c0 = gw1c0[T][r];                      // sum over the 32 point columns of this half
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) c0 += __shfl_xor(c0, off);
        rr[PART_GW0 + feat * 32 + l31] = gw0[T][r];
        rr[PART_GW1 + feat * 32 + l31] = gw1[T][r] + (l31 == 0 ? c0 : 0.0f);
      }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) gb1[o] += __shfl_xor(gb1[o], 32);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) gb1[o] += __shfl_xor(gb1[o], off);
    if (lane == 0) {
#pragma unroll
      for (int o = 0; o < NOUT; ++o) rr[PART_GB1 + o] = gb1[o];
    }
  }
  __syncthreads();
  float* part = partials + (size_t)blockIdx.x * PART_STRIDE;
  for (int v = threadIdx.x; v < PART_GB1 + NOUT; v += blockDim.x)
    part[v] = (red[v] + red[PART_STRIDE + v]) + (red[2 * PART_STRIDE + v] + red[3 * PART_STRIDE + v]);
}

template <int NL, int ACT>
bool launch_bwd_pipe_one(uint32_t active, int blocks, size_t shm, hipStream_t s, const dsu_sdf_mlp& mlp,
                         const float* pts, int64_t n, float radius, float eps, float eps2,
                         const float* d_sdf, const float* d_grad, const float* d_feature,
                         const float* d_laplace, float* partials, const __half2* enc, float2* dinbuf,
                         const int32_t* perm) {
  if ((int)active != ACT) return false;
  static bool lds_ok = false;
  if (!lds_ok) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sdf_fd_bwd_pipe_kernel<NL, ACT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(PIPE_LDS_F * sizeof(float))) != hipSuccess)
      return false;
    lds_ok = true;
  }
  (void)shm;
  sdf_fd_bwd_pipe_kernel<NL, ACT><<<dim3(blocks), dim3(256), PIPE_LDS_F * sizeof(float), s>>>(
      mlp, pts, n, radius, eps, eps2, d_sdf, d_grad, d_feature, d_laplace, partials, enc, dinbuf, perm);
  return true;
}

template <int NL>
bool launch_bwd_pipe(uint32_t active, int blocks, size_t shm, hipStream_t s, const dsu_sdf_mlp& mlp,
                     const float* pts, int64_t n, float radius, float eps, float eps2,
                     const float* d_sdf, const float* d_grad, const float* d_feature,
                     const float* d_laplace, float* partials, const __half2* enc, float2* dinbuf,
                     const int32_t* perm) {
#define DSU_PIPE_TRY(A)                                                                          \
  if (launch_bwd_pipe_one<NL, A>(active, blocks, shm, s, mlp, pts, n, radius, eps, eps2, d_sdf,   \
                                 d_grad, d_feature, d_laplace, partials, enc, dinbuf, perm))      \
    return true;
  DSU_PIPE_TRY(4) DSU_PIPE_TRY(5) DSU_PIPE_TRY(6) DSU_PIPE_TRY(7)
#undef DSU_PIPE_TRY
  return false;
}

// ---------------------------------------------------------------------------- K2: scatter
// Table gradients from dinbuf (SPLIT path).  One point per thread, SC_THREADS threads per
// workgroup, one LEVEL at a time for the workgroup's SC_THREADS points (so the 4096-slot cache only ever
// holds one level's entries and is flushed per level), all 7 evaluations of that level back to
// back.  Same segmented DPP merge, per-wave queue and fixed-point cache as the fused kernel.
// Sizes (variant builds override them: -DDSU_SC_THREADS / -DDSU_SC_LOG2 / -DDSU_SC_QCAP).  With the
// samples in Morton order a workgroup's points of one level touch a compact block of entries, so a
// small cache is enough and what matters is how many waves a CU holds to hide the returning LDS
// atomics: LDS bytes per workgroup = 20 * slots + 12 * QCAP * waves.
// Round 6: 256 threads, 1024 slots, 512-triple queues = 45 KB per workgroup, THREE workgroups per CU
// (768 resident) instead of one 512-thread workgroup with 150 KB: pair 0.394 -> 0.362 ms in the
// optimisation (same-box A/B of five shapes, profiles/round6_scatter_shapes_ab.txt: 512 / 2^12 / 704 /
// 256 = 0.394, 256 / 2^11 / 512 / 512 = 0.380, 256 / 2^10 / 512 / 768 = 0.362, 256 / 2^9 / 512 / 1024 =
// 0.397, 128 / 2^10 / 512 / 1024 = 0.401).  Items are 256 points x one level: their boxes of cells are
// about half as large, so the smaller dense tile still takes most items, and a CU's three workgroups
// hide each other's barriers and returning atomics (and leave LDS to other kernels when several
// drawings share the GPU).
#ifndef DSU_SC_THREADS
#define DSU_SC_THREADS 256
#endif
#ifndef DSU_SC_LOG2
#define DSU_SC_LOG2 10
#endif
#ifndef DSU_SC_QCAP
#define DSU_SC_QCAP 512
#endif
#ifndef DSU_SC_MAXBLOCKS
#define DSU_SC_MAXBLOCKS 768
#endif
constexpr int SC_MAXBLOCKS = DSU_SC_MAXBLOCKS;                // resident workgroups (256 CUs x per-CU count)
constexpr int SC_THREADS = DSU_SC_THREADS;
constexpr int SC_QCAP = DSU_SC_QCAP;                          // triples per wave queue (>= 512 + 64)
constexpr int SC_LOG2 = DSU_SC_LOG2;
constexpr int SC_SLOTS = 1 << SC_LOG2;
constexpr int SC_LDS_F = SC_SLOTS + 4 * SC_SLOTS + (SC_THREADS / 64) * 3 * SC_QCAP;
// Dense tile (see the kernel): the 512 Morton-ordered samples of one iteration touch a small box of
// cells on every level; their gradients are accumulated over that box in LDS — [corner in box][2]
// 64-bit fixed point in the cache's and the queues' bytes — and flushed once per iteration.
constexpr int SC_TILE_CAP = (SC_LDS_F * 4) / 16;              // box corners x 16 B in the cache's + queues' bytes
constexpr int SC_BBOX_OFF = SC_LDS_F;                         // 2 x 6 ints: min xyz, max xyz (two parities)
constexpr int SC_LDS_TOTAL = SC_LDS_F + 16;
static_assert(SC_QCAP % 4 == 0 && SC_QCAP >= 512, "queue: one full evaluation of a wave must fit");

__device__ __forceinline__ uint32_t sc_slot(uint32_t entry) {
  return (entry * 2654435761u) >> (32 - SC_LOG2);
}
__device__ __forceinline__ void sc_commit_from(uint32_t* keys, unsigned long long* acc,
                                               float* __restrict__ gtable, uint32_t entry,
                                               uint32_t slot, float v0, float v1) {
#pragma unroll
  for (int probe = 1; probe < 3; ++probe) {
    slot = (slot + 1) & (SC_SLOTS - 1);
    const uint32_t old = atomicCAS(&keys[slot], GC_EMPTY, entry);
    if (old == GC_EMPTY || old == entry) {
      atomicAdd(&acc[2 * slot], gc_fix(v0));
      atomicAdd(&acc[2 * slot + 1], gc_fix(v1));
      return;
    }
  }
  unsafeAtomicAdd(gtable + (size_t)entry * 2, v0);
  unsafeAtomicAdd(gtable + (size_t)entry * 2 + 1, v1);
}

#ifdef DSU_AB_SWITCHES
// variant builds: per level, workgroups that took the dense tile / the cache, and the tile cells used
__device__ unsigned long long dsu_sc_stat[3 * 16];
// per workgroup: clocks of the last launch per level [block][16] (+ [15] = whole kernel)
__device__ unsigned long long dsu_sc_clk[256 * 16];
#endif

// Sum of the MLP-part kernel's per-workgroup partial vectors into the reference-layout gradients:
// 64 elements per workgroup, 16 interleaved slices of the workgroup range, then the slices in order
// (the same order whatever the block size; `red`: 16 * 64 floats of LDS).
struct OwnReduce {
  const float* partials;
  int nblocks;
  float *g_w0, *g_b0, *g_w1, *g_b1;
};
constexpr int OWN_RED_BLOCKS = (PART_GB1 + NOUT + 63) / 64;

template <int NL>
__device__ __forceinline__ void reduce_own_block(const OwnReduce& o, int blk, float* red) {
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6, ng = blockDim.x >> 6;
  const int v = blk * 64 + e;
  const bool in_range = v < PART_GB1 + NOUT;
  for (int sl = g; sl < 16; sl += ng) {
    float acc = 0.0f;
    if (in_range)
      for (int b = sl; b < o.nblocks; b += 16) acc += o.partials[(size_t)b * PART_STRIDE + v];
    red[sl * 64 + e] = acc;
  }
  __syncthreads();
  if (g != 0 || !in_range) return;
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k * 64 + e];
  if (v < PART_GW1) {
    const int feat = v >> 5, k = v & 31;
    if (k < MC<NL>::KIN) {
      const int c = ref_col<NL>(k);
      if (c >= 0) o.g_w0[feat * MC<NL>::DIN + c] += s;
      else o.g_b0[feat] += s;
    }
  } else if (v < PART_GB1) {
    const int feat = (v - PART_GW1) >> 5, oo = (v - PART_GW1) & 31;
    if (oo < NOUT) o.g_w1[oo * HID + feat] += s;
  } else {
    o.g_b1[v - PART_GB1] += s;
  }
}

template <int NL>
__global__ void reduce_partials_mfma_kernel(OwnReduce o) {
  __shared__ float red[16 * 64];
  reduce_own_block<NL>(o, blockIdx.x, red);
}

template <int NL>
__global__ __launch_bounds__(SC_THREADS) void sdf_fd_scatter_kernel(
    GridMeta m, const float* __restrict__ pts, int64_t n, float radius, float eps, uint32_t active,
    const float2* __restrict__ dinbuf, float* __restrict__ gtable, int merge_levels,
    int centre_acc, int dense_levels, int* __restrict__ work_counter, OwnReduce own,
    dsu_partial_reduce extra) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // The first workgroups of the launch sum partial gradient vectors (the MLP part's own and, when
  // the caller passes one, another kernel's: partial_reduce.h) and leave; they are dispatched
  // first, take a few microseconds, and the scatter workgroups that inherit their slots lose
  // nothing because the scatter's work items are handed out dynamically.
  const int n_red_own = own.partials ? OWN_RED_BLOCKS : 0;
  const int n_red = n_red_own + (extra.partials ? dsu_red::blocks_of(extra) : 0);
  if ((int)blockIdx.x < n_red) {
    if ((int)blockIdx.x < n_red_own) reduce_own_block<NL>(own, blockIdx.x, lds);
    else dsu_red::reduce_block(extra, blockIdx.x - n_red_own, lds);
    return;
  }
  const int sc_block = (int)blockIdx.x - n_red, sc_grid = (int)gridDim.x - n_red;
  uint32_t* c_keys = reinterpret_cast<uint32_t*>(lds);
  unsigned long long* c_acc = reinterpret_cast<unsigned long long*>(lds + SC_SLOTS);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* qbase = lds + 5 * SC_SLOTS + wave * 3 * SC_QCAP;
  uint32_t* q_ent = reinterpret_cast<uint32_t*>(qbase);
  float* q_v0 = qbase + SC_QCAP;
  float* q_v1 = qbase + 2 * SC_QCAP;
  unsigned long long* tile = reinterpret_cast<unsigned long long*>(lds);   // tile mode: cache + queues' bytes
  int* bbox = reinterpret_cast<int*>(lds + SC_BBOX_OFF);                   // 2 x (min xyz, max xyz)
  for (int t = threadIdx.x; t < SC_SLOTS; t += blockDim.x) {
    c_keys[t] = GC_EMPTY;
    c_acc[2 * t] = 0ull;
    c_acc[2 * t + 1] = 0ull;
  }
  if (threadIdx.x < 12) bbox[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0x7fffffff : -0x7fffffff;
  int region = 0;                      // what the shared bytes are initialised for: 0 cache, 1 tile
  __syncthreads();
#ifdef DSU_AB_SWITCHES
  const unsigned long long clk_start = wall_clock64();
  if (threadIdx.x < 15 && sc_block < 256) dsu_sc_clk[sc_block * 16 + threadIdx.x] = 0ull;   // per-level item time of this launch
#endif
  // Work items = (512-point chunk of the Morton-ordered samples, level), handed out dynamically:
  // an item is self-contained (its tile is flushed at its end), items differ a lot in cost (how
  // many offsets leave their cell, tile or cache), and n / 512 chunks do not divide by the 256
  // resident workgroups — with one static range per workgroup the slowest ran 1.6x the mean
  // (tools/scatter_stats.py) and a third, almost empty iteration cost every workgroup a full one.
  // Item j = chunk j / active, level j % active.  Workgroup b starts with items b and b + G
  // (G = grid size); further ones come from `work_counter` (zeroed by the MLP-part kernel, which
  // precedes this one on the stream), fetched one item ahead so that the next item's loads can
  // be issued a whole item before they are needed.
  const int64_t n_chunks = (n + blockDim.x - 1) / blockDim.x;
  const int64_t n_items = n_chunks * (int64_t)active;
  int* next_slot = reinterpret_cast<int*>(lds + SC_BBOX_OFF) + 12;   // the item after next, via LDS
  int64_t cur = sc_block, nxt = (int64_t)sc_block + sc_grid;
  float pn[3] = {0.f, 0.f, 0.f};       // position and the seven dIn pairs of the NEXT item to process
  float2 dnx[7];
#pragma unroll
  for (int e = 0; e < 7; ++e) dnx[e] = make_float2(0.f, 0.f);
  auto request = [&](int64_t item) {   // loads of an item: unconditional, from clamped addresses
    const int64_t chunk = item / (int64_t)active;
    const uint32_t lv = (uint32_t)(item - chunk * (int64_t)active);
    int64_t idx = chunk * blockDim.x + threadIdx.x;
    idx = idx < n ? idx : n - 1;
    pn[0] = pts[idx * 3]; pn[1] = pts[idx * 3 + 1]; pn[2] = pts[idx * 3 + 2];
#pragma unroll
    for (int e = 0; e < 7; ++e) dnx[e] = dinbuf[((size_t)e * active + lv) * n + idx];
  };
  if (cur < n_items) request(cur);
  int parity = 0;                      // bounding-box words in use (double-buffered over items)
  int qn = 0;
  auto drain = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < qn; i0 += 64) {
      const int qi = i0 + lane;
      if (qi < qn) {
        const uint32_t ent = q_ent[qi];
        const float a0 = q_v0[qi], a1 = q_v1[qi];
        const uint32_t slot = sc_slot(ent);
        const uint32_t old = atomicCAS(&c_keys[slot], GC_EMPTY, ent);
        if (old == GC_EMPTY || old == ent) {
          atomicAdd(&c_acc[2 * slot], gc_fix(a0));
          atomicAdd(&c_acc[2 * slot + 1], gc_fix(a1));
        } else {
          sc_commit_from(c_keys, c_acc, gtable, ent, slot, a0, a1);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    qn = 0;
  };
  // cache -> table: one global atomic pair per touched entry, then reset (callers synchronise)
  auto flush_cache = [&]() {
    for (int t = threadIdx.x; t < SC_SLOTS; t += blockDim.x) {
      const uint32_t key = c_keys[t];
      if (key != GC_EMPTY) {
        unsafeAtomicAdd(gtable + (size_t)key * 2, gc_unfix(c_acc[2 * t]));
        unsafeAtomicAdd(gtable + (size_t)key * 2 + 1, gc_unfix(c_acc[2 * t + 1]));
        c_keys[t] = GC_EMPTY;
        c_acc[2 * t] = 0ull;
        c_acc[2 * t + 1] = 0ull;
      }
    }
  };

#pragma unroll 1
  while (cur < n_items) {              // uniform over the workgroup
    {
#ifdef DSU_AB_SWITCHES
      const unsigned long long clk_item = wall_clock64();
#endif
      const int64_t chunk = cur / (int64_t)active;
      const int lev = (int)(cur - chunk * (int64_t)active);
      const float l_scale = m.scale[lev];
      const uint32_t l_off = m.off[lev], hsize = m.off[lev + 1] - m.off[lev];
      const uint32_t l_res = m.res[lev], l_hashed = m.hashed[lev];
      const int64_t i = chunk * blockDim.x + threadIdx.x;
      const bool valid = i < n;
      const float p[3] = {pn[0], pn[1], pn[2]};
      float2 dv[7];
#pragma unroll
      for (int e = 0; e < 7; ++e) {
        dv[e].x = valid ? dnx[e].x : 0.0f;
        dv[e].y = valid ? dnx[e].y : 0.0f;
      }
      if (nxt < n_items) request(nxt);   // uniform: the next item's loads, a whole item ahead
      if (threadIdx.x == 0) *next_slot = 2 * sc_grid + atomicAdd(work_counter, 1);
      // ---- pass 1 (straight line): the centre's cell, which offsets stay in it, their sums.
      // The interpolation weight of a corner is MULTILINEAR in the position inside the cell and an
      // offset evaluation moves ONE coordinate: for an offset that stays in the centre's cell
      //     w_c(f + D e_a) = w_c(f) + D * dw_c/df_a          (exactly, in real arithmetic)
      // so the sum over the centre and those offsets of d_e * w_c(f_e) is
      //     (sum_e d_e) * w_c(f)  +  sum_a (sum_{e on a} d_e * D_e) * dw_c/df_a :
      // D0 = sum d_e and Da[a] = sum d_e * D_e are accumulated here (one contracted coordinate and
      // one floor per offset instead of a full cell and eight weights); the eight corner sums are
      // formed once, in the last emission below.  Offsets that leave the cell are emitted on their
      // own.  (Different rounding than seven separate products: the table gradient's tolerance
      // is 1e-2 relative; the tests hold it to 1e-4.)
      const CellPos ccp = cell_of(l_scale, contract(p[0], radius), contract(p[1], radius),
                                  contract(p[2], radius));
      float2 D0 = dv[0], Da[3];
      uint32_t left = 0u;
      // fd_point clamps EVERY coordinate of an offset evaluation to the box (geometry.py:170) but
      // not the centre: for a point outside the box (the perturbed random points can be) the
      // offsets' other coordinates differ from the centre's — no shared cell, all six on their own
      const bool inside = fabsf(p[0]) <= radius && fabsf(p[1]) <= radius && fabsf(p[2]) <= radius;
      int lo[3], hi[3];                  // cells this lane touches (bounding box of the iteration)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        Da[a] = make_float2(0.f, 0.f);
        lo[a] = valid ? (int)ccp.c[a] : 0x7fffffff;
        hi[a] = valid ? (int)ccp.c[a] : -0x7fffffff;
      }
#pragma unroll
      for (int e = 1; e < 7; ++e) {
        const int ax = (e - 1) >> 1;
        const float off = ((e - 1) & 1) ? -eps : eps;
        const float qa = fminf(fmaxf(p[ax] + off, -radius), radius);      // fd_point's clamp
        const float posa = fmaf(l_scale, contract(qa, radius), 0.5f);     // cell_of's position
        const float fl = floorf(posa);
        const bool same = ((uint32_t)(int)fl == ccp.c[ax]) & (centre_acc != 0) & inside;
        const float delta = posa - ((float)(int)ccp.c[ax] + ccp.f[ax]);
        D0.x += same ? dv[e].x : 0.0f;
        D0.y += same ? dv[e].y : 0.0f;
        Da[ax].x += same ? dv[e].x * delta : 0.0f;
        Da[ax].y += same ? dv[e].y * delta : 0.0f;
        left |= (valid & !same) ? (1u << e) : 0u;
        if (valid & inside) {
          lo[ax] = min(lo[ax], (int)fl);
          hi[ax] = max(hi[ax], (int)fl);
        }
      }
      if (valid & !inside) {             // rare, divergent: the general form of the six offsets
#pragma unroll 1
        for (int e = 1; e < 7; ++e) {
          float q[3];
          fd_point(p, e, eps, radius, q);
          const CellPos cq = cell_of(l_scale, contract(q[0], radius), contract(q[1], radius),
                                     contract(q[2], radius));
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], (int)cq.c[a]);
            hi[a] = max(hi[a], (int)cq.c[a]);
          }
        }
      }
      // ---- the iteration's box -> tile or cache.  The samples are in Morton order: the 512 points
      // of one iteration touch a small box of cells on EVERY level (hashed levels too — the hash
      // only decides where a corner's sum finally goes).  Their gradients are accumulated in a
      // dense LDS tile over that box, [corner in box][2] 64-bit fixed point, with one ds_add_u64
      // pair per corner and nothing else — no key compare-and-swap, no per-wave queue and drain —
      // and the tile is flushed to the table (dense index or hash of the corner) at the end of
      // the iteration.  An iteration whose box does not fit the tile (a jump of the Morton
      // curve) goes through the hashed cache as before.
      bool dense = false;
      int x0 = 0, y0 = 0, z0 = 0, ddx = 1, ddxy = 1, vol = 0;
      int64_t after_next;
      {
        int* bb = bbox + 6 * parity;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], off));
            hi[a] = max(hi[a], __shfl_xor(hi[a], off));
          }
        }
        if (lane == 0) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            atomicMin(&bb[a], lo[a]);
            atomicMax(&bb[3 + a], hi[a]);
          }
        }
        __syncthreads();                 // (A) every wave's box is in (and the item after next)
        after_next = *next_slot;
        x0 = bb[0]; y0 = bb[1]; z0 = bb[2];
        const long long ex = (long long)bb[3] - x0 + 2, ey = (long long)bb[4] - y0 + 2,
                        ez = (long long)bb[5] - z0 + 2;                  // + the far corners
        if (dense_levels && bb[3] >= bb[0] && ex * ey * ez <= (long long)SC_TILE_CAP && ex < 4096 && ey < 4096) {
          dense = true;
          ddx = (int)ex; ddxy = (int)(ex * ey); vol = (int)(ex * ey * ez);
        }
        // the OTHER parity's words are reset here for the next iteration: their readers (previous
        // iteration) are past that iteration's barrier (B), and the next iteration's atomics on
        // them come after this iteration's barrier (B), which follows this store in program order
        if (threadIdx.x < 6) bbox[6 * (parity ^ 1) + threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : -0x7fffffff;
        parity ^= 1;
      }
#ifdef DSU_AB_SWITCHES
      if (threadIdx.x == 0) {
        atomicAdd(&dsu_sc_stat[3 * lev + (dense ? 0 : 1)], 1ull);
        atomicAdd(&dsu_sc_stat[3 * lev + 2], (unsigned long long)vol);
      }
#endif
      if ((int)dense != region) {        // uniform: the shared bytes change hands
        if (region == 0) {               // cache -> tile (the cache was flushed at the end of its item)
          for (int t = threadIdx.x; t < SC_TILE_CAP * 2; t += blockDim.x) tile[t] = 0ull;
        } else {                         // tile -> cache (the tile was flushed at the end of its item)
          for (int t = threadIdx.x; t < SC_SLOTS; t += blockDim.x) {
            c_keys[t] = GC_EMPTY;
            c_acc[2 * t] = 0ull;
            c_acc[2 * t + 1] = 0ull;
          }
        }
        region = (int)dense;
        __syncthreads();
      }

      // ---- pass 2: emissions — the offsets that left the centre's cell, then (e = 7) the centre cell
#pragma unroll 1
      for (int e = 1; e < 8; ++e) {
        CellPos cp = ccp;
        float v[16];
        bool flush;
        if (e < 7) {
          flush = (left >> e) & 1u;
          if (__ballot(flush) == 0ull) continue;   // uniform: this offset stayed in the centre cell everywhere
          float2 d = dv[1];
#pragma unroll
          for (int k = 2; k < 7; ++k) {
            d.x = e == k ? dv[k].x : d.x;
            d.y = e == k ? dv[k].y : d.y;
          }
          float q[3];
          fd_point(p, e, eps, radius, q);          // the general form: every coordinate clamped
          cp = cell_of(l_scale, contract(q[0], radius), contract(q[1], radius), contract(q[2], radius));
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float w = corner_weight(cp, c);
            v[2 * c] = w * d.x;
            v[2 * c + 1] = w * d.y;
          }
        } else {
          const float X[2] = {1.0f - cp.f[0], cp.f[0]}, Y[2] = {1.0f - cp.f[1], cp.f[1]},
                      Z[2] = {1.0f - cp.f[2], cp.f[2]};
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int cx_ = c & 1, cy_ = (c >> 1) & 1, cz_ = (c >> 2) & 1;
            const float yz = Y[cy_] * Z[cz_], xz = X[cx_] * Z[cz_], xy = X[cx_] * Y[cy_];
            const float w = X[cx_] * yz;
            // dw/df_a = +- (product of the other two factors): + on the far side of axis a
            const float gx = cx_ ? yz : -yz, gy = cy_ ? xz : -xz, gz = cz_ ? xy : -xy;
            v[2 * c] = fmaf(D0.x, w, fmaf(Da[0].x, gx, fmaf(Da[1].x, gy, Da[2].x * gz)));
            v[2 * c + 1] = fmaf(D0.y, w, fmaf(Da[0].y, gx, fmaf(Da[1].y, gy, Da[2].y * gz)));
          }
          flush = valid;
        }
        // lanes with nothing to emit this round: a key no neighbour shares, zero contribution
        const int key = flush ? (int)((cp.c[0] & 1023u) | ((cp.c[1] & 1023u) << 10) | ((cp.c[2] & 1023u) << 20))
                              : (0x40000000 | lane);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = flush ? v[k] : 0.0f;
        bool lead = flush;
        // Same-cell run merge (segmented sums over the 16-lane DPP rows): worth its ~200
        // instructions only where neighbouring lanes of the Morton order DO share cells — the
        // coarse levels.  Where most emitting lanes are alone in their cell (fine levels: more
        // than three quarters of them start a run) every lane adds its own sum instead; the few
        // shared cells then take two or three atomics.
        const int key_next = dpp_i<0x101>(key), key_prev = dpp_i<0x111>(key);
        const int l15 = lane & 15;
        const bool starts = ((l15 == 0) | (key_prev != key)) & flush;
        const int n_flush = __popcll(__ballot(flush)), n_start = __popcll(__ballot(starts));
#if defined(DSU_SC_ABL) && (DSU_SC_ABL & 4)
        if (false) {                                                   // (timing ablation: no run merge)
#else
        if (lev < merge_levels && 4 * n_start <= 3 * n_flush) {      // uniform
#endif
          // neighbour keys with all lanes active (a DPP move under the EXEC mask of a
          // short-circuit reads disabled source lanes as 0)
          int ee = ((l15 == 15) | (key_next != key)) ? 1 : 0;   // run ends at this lane
#define DSU_SEG_STEP(CTRL)                                              \
          {                                                             \
            const int eo = dpp_i<CTRL>(ee);                             \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) {            \
              const float vo = dpp_f<CTRL>(v[k]);                       \
              v[k] += ee ? 0.0f : vo;                                   \
            }                                                           \
            ee |= eo;                                                   \
          }
          DSU_SEG_STEP(0x101) DSU_SEG_STEP(0x102) DSU_SEG_STEP(0x104) DSU_SEG_STEP(0x108)
#undef DSU_SEG_STEP
          // padding lanes (beyond the range) and lanes that emit nothing never lead a run
          lead = starts;
        }
        if (dense) {                     // uniform: straight into the tile
          if (lead) {
            const int base = ((int)cp.c[0] - x0) + ((int)cp.c[1] - y0) * ddx + ((int)cp.c[2] - z0) * ddxy;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const int t = base + (c & 1) + ((c >> 1) & 1) * ddx + ((c >> 2) & 1) * ddxy;
#if defined(DSU_SC_ABL) && (DSU_SC_ABL & 1)
              if (v[2 * c] == 12345.678f) tile[2 * t] = gc_fix(v[2 * c + 1]) + (unsigned long long)t;   // (timing ablation: no LDS atomics)
#else
              atomicAdd(&tile[2 * t], gc_fix(v[2 * c]));
              atomicAdd(&tile[2 * t + 1], gc_fix(v[2 * c + 1]));
#endif
            }
          }
          continue;
        }
        const unsigned long long bal = __ballot(lead);
        if (lead) {
          const int pos = qn + 8 * __popcll(bal & ((1ull << lane) - 1ull));
          uint32_t ent[8];
#pragma unroll
          for (int c = 0; c < 8; ++c)
            ent[c] = l_off + grid_index(l_hashed, hsize, l_res, cp.c[0] + (c & 1),
                                        cp.c[1] + ((c >> 1) & 1), cp.c[2] + ((c >> 2) & 1));
#pragma unroll
          for (int q4 = 0; q4 < 2; ++q4) {
            *reinterpret_cast<uint4*>(&q_ent[pos + 4 * q4]) =
                make_uint4(ent[4 * q4], ent[4 * q4 + 1], ent[4 * q4 + 2], ent[4 * q4 + 3]);
            *reinterpret_cast<float4*>(&q_v0[pos + 4 * q4]) =
                make_float4(v[8 * q4], v[8 * q4 + 2], v[8 * q4 + 4], v[8 * q4 + 6]);
            *reinterpret_cast<float4*>(&q_v1[pos + 4 * q4]) =
                make_float4(v[8 * q4 + 1], v[8 * q4 + 3], v[8 * q4 + 5], v[8 * q4 + 7]);
          }
        }
        qn += 8 * __popcll(bal);
        if (qn + 512 > SC_QCAP) drain();
      }
      if (!dense) drain();               // cache mode: the item's queued contributions
      __syncthreads();                   // (B) every wave's adds are in (and: see the box words above)
      if (dense) {
        // flush the box: one global atomic pair per touched corner, then reset
        for (int t = threadIdx.x; t < vol; t += blockDim.x) {
          const unsigned long long a0 = tile[2 * t], a1 = tile[2 * t + 1];
          if (a0 | a1) {
            const int z = t / ddxy, rr = t - z * ddxy;
            const int y = rr / ddx, x = rr - y * ddx;
            const uint32_t g = l_off + grid_index(l_hashed, hsize, l_res, (uint32_t)(x0 + x),
                                                  (uint32_t)(y0 + y), (uint32_t)(z0 + z));
#if !(defined(DSU_SC_ABL) && (DSU_SC_ABL & 2))
            unsafeAtomicAdd(gtable + (size_t)g * 2, gc_unfix(a0));
            unsafeAtomicAdd(gtable + (size_t)g * 2 + 1, gc_unfix(a1));
#else
            if (g == 0xffffffffu) gtable[0] = gc_unfix(a0) + gc_unfix(a1);      // (timing ablation: no global atomics)
#endif
            tile[2 * t] = 0ull;
            tile[2 * t + 1] = 0ull;
          }
        }
        // (the next item's adds come after its barrier (A): no barrier needed here)
      } else {
        flush_cache();                   // cache mode: this item's entries (the next item may be a tile)
      }
#ifdef DSU_AB_SWITCHES
      if (threadIdx.x == 0 && sc_block < 256 && lev < 15) dsu_sc_clk[sc_block * 16 + lev] += wall_clock64() - clk_item;
#endif
      cur = nxt;
      nxt = after_next;
    }
  }
#ifdef DSU_AB_SWITCHES
  if (threadIdx.x == 0 && sc_block < 256) dsu_sc_clk[sc_block * 16 + 15] = wall_clock64() - clk_start;
#endif
}

#ifndef DSU_BWD_MFMA_MAX_BLOCKS
#define DSU_BWD_MFMA_MAX_BLOCKS 256
#endif
constexpr int BWD_MFMA_MAX_BLOCKS = DSU_BWD_MFMA_MAX_BLOCKS;   // one workgroup per CU (458 registers: one wave per SIMD)

}  // namespace

#define DSU_DISPATCH_NL(nl, ...)                       \
  switch (nl) {                                        \
    case 10: { constexpr int NL = 10; __VA_ARGS__ } break; \
    case 12: { constexpr int NL = 12; __VA_ARGS__ } break; \
    default: return DSU_EUNSUP;                        \
  }

// VALU implementations (hashgrid.hip), kept for A/B runs: DSU_SDF_IMPL=valu
extern "C" int dsu_sdf_fwd_valu(const dsu_hashgrid_cfg*, const void*, const dsu_sdf_mlp*,
                                const float*, int64_t, float, uint32_t, uint32_t, float*, void*);
extern "C" int dsu_sdf_fd_fwd_valu(const dsu_hashgrid_cfg*, const void*, const dsu_sdf_mlp*,
                                   const float*, int64_t, float, float, uint32_t, float*, float*,
                                   float*, float*, void*, const int32_t*, void*);
extern "C" int dsu_sdf_fd_bwd_valu(const dsu_hashgrid_cfg*, const void*, const dsu_sdf_mlp*,
                                   const float*, int64_t, float, float, uint32_t, const float*,
                                   const float*, const float*, const float*, float*, float*,
                                   float*, float*, float*, void*, int64_t, void*);
extern "C" int64_t dsu_sdf_fd_bwd_workspace_bytes_valu(const dsu_hashgrid_cfg*, int64_t);

// Default mix (measured on MI355X, N = 262 144 ray-ordered samples, 4 active levels):
//   forward:  VALU 0.22 ms vs MFMA 0.30 ms  -> VALU (gather-latency bound; 2 waves/SIMD help)
//   backward: VALU 2.31 ms vs MFMA 1.60 ms  -> MFMA
// DSU_SDF_IMPL=valu|mfma forces one implementation for everything (A/B runs, tests).
// Default (measured round 2, N = 262 144 ray-ordered samples at the training step size, 4/5/6 levels):
//   fused 0.60 / 0.72 / 0.85 ms, two kernels 0.54 / 0.61 / 0.71 ms -> two kernels.
// DSU_BWD_SPLIT=0 selects the fused kernel.
static bool bwd_split() {          // read per call: the variant tests run both forms in one process
  return dsu_ab_int("DSU_BWD_SPLIT", 1) != 0;
}

static bool use_valu(bool forward) {
  static int v = -1;
  if (v < 0) v = dsu_ab_is("DSU_SDF_IMPL", "valu") ? 1 : (dsu_ab_is("DSU_SDF_IMPL", "mfma") ? 2 : 0);
  return v == 1 || (v == 0 && forward);
}

extern "C" {

#ifdef DSU_AB_SWITCHES
int dsu_debug_sc_stats(unsigned long long* out48, int reset) {
  if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(dsu_sc_stat), 48 * sizeof(unsigned long long)) != hipSuccess)
    return DSU_ELAUNCH;
  if (reset) {
    unsigned long long z[48] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(dsu_sc_stat), z, sizeof(z)) != hipSuccess) return DSU_ELAUNCH;
  }
  return DSU_OK;
}
int dsu_debug_sc_clocks(unsigned long long* out4096) {
  return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(dsu_sc_clk), 4096 * sizeof(unsigned long long)) == hipSuccess
             ? DSU_OK : DSU_ELAUNCH;
}
#endif

int dsu_sdf_fwd(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                const float* pts, int64_t n, float radius, uint32_t active_levels,
                uint32_t n_out, float* out, void* stream) {
  if (use_valu(true))
    return dsu_sdf_fwd_valu(cfg, table_f16, mlp, pts, n, radius, active_levels, n_out, out, stream);
  if (!cfg || !table_f16 || !mlp || (!pts && n) || (!out && n) || n < 0) return DSU_EINVAL;
  if (!mlp->w0 || !mlp->b0 || !mlp->w1 || !mlp->b1) return DSU_EINVAL;
  if (active_levels > cfg->n_levels) return DSU_EINVAL;
  if (n_out != 1 && n_out != NOUT) return DSU_EUNSUP;
  GridMeta m;
  int rc = make_meta(cfg, &m);
  if (rc) return rc;
  if (n == 0) return DSU_OK;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = dsu_capped_blocks(n, 256, 4096);
  DSU_DISPATCH_NL(cfg->n_levels, {
    if (n_out == 1)
      sdf_fwd_mfma_kernel<NL, 1><<<dim3(blocks), dim3(256), 0, s>>>(
          (const __half2*)table_f16, m, *mlp, pts, n, radius, active_levels, out);
    else
      sdf_fwd_mfma_kernel<NL, NOUT><<<dim3(blocks), dim3(256), 0, s>>>(
          (const __half2*)table_f16, m, *mlp, pts, n, radius, active_levels, out);
  });
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_sdf_fd_fwd_sorted(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                   const float* pts, const int32_t* perm, int64_t n, float radius, float eps,
                   uint32_t active_levels, float* sdf, float* grad, float* feature,
                   float* laplace, void* enc_cache, void* stream) {
  // the cache and the permuted write-back are features of the VALU kernel
  if (use_valu(true) || enc_cache != nullptr || perm != nullptr)
    return dsu_sdf_fd_fwd_valu(cfg, table_f16, mlp, pts, n, radius, eps, active_levels, sdf, grad,
                               feature, laplace, enc_cache, perm, stream);
  if (!cfg || !table_f16 || !mlp || (!pts && n) || (!sdf && n) || n < 0) return DSU_EINVAL;
  if (!mlp->w0 || !mlp->b0 || !mlp->w1 || !mlp->b1) return DSU_EINVAL;
  if (active_levels > cfg->n_levels || !(eps > 0.0f)) return DSU_EINVAL;
  GridMeta m;
  int rc = make_meta(cfg, &m);
  if (rc) return rc;
  if (n == 0) return DSU_OK;
  hipStream_t s = (hipStream_t)stream;
  const float eps2 = (float)((double)eps * (double)eps);
  const int blocks = dsu_capped_blocks(n, 256, 4096);
  DSU_DISPATCH_NL(cfg->n_levels, {
    sdf_fd_fwd_mfma_kernel<NL><<<dim3(blocks), dim3(256), 0, s>>>(
        (const __half2*)table_f16, m, *mlp, pts, n, radius, eps, eps2, active_levels, sdf, grad,
        feature, laplace);
  });
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_sdf_fd_fwd_cached(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                          const dsu_sdf_mlp* mlp, const float* pts, int64_t n, float radius,
                          float eps, uint32_t active_levels, float* sdf, float* grad,
                          float* feature, float* laplace, void* enc_cache, void* stream) {
  return dsu_sdf_fd_fwd_sorted(cfg, table_f16, mlp, pts, nullptr, n, radius, eps, active_levels,
                               sdf, grad, feature, laplace, enc_cache, stream);
}

int dsu_sdf_fd_fwd(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                   const float* pts, int64_t n, float radius, float eps,
                   uint32_t active_levels, float* sdf, float* grad, float* feature,
                   float* laplace, void* stream) {
  return dsu_sdf_fd_fwd_sorted(cfg, table_f16, mlp, pts, nullptr, n, radius, eps, active_levels,
                               sdf, grad, feature, laplace, nullptr, stream);
}

int64_t dsu_sdf_fd_enc_cache_bytes(int64_t n, uint32_t active_levels) {
  if (n < 0 || active_levels > DSU_MAX_LEVELS) return DSU_EINVAL;
  return (int64_t)7 * n * active_levels * 4;
}

int64_t dsu_sdf_fd_bwd_workspace_bytes(const dsu_hashgrid_cfg* cfg, int64_t n) {
  if (use_valu(false)) return dsu_sdf_fd_bwd_workspace_bytes_valu(cfg, n);
  if (!cfg || n < 0) return DSU_EINVAL;
  if (cfg->n_levels != 10 && cfg->n_levels != 12) return DSU_EUNSUP;
  const int blocks = dsu_capped_blocks(n, 256, BWD_MFMA_MAX_BLOCKS);
  int64_t bytes = (int64_t)blocks * PART_STRIDE * sizeof(float);
  // split form: + dIn of every (evaluation, point, level) as float2
  // (+ the scatter kernel's work counter behind it)
  if (bwd_split()) bytes += (int64_t)7 * n * cfg->n_levels * (int64_t)sizeof(float2) + 256;
  return bytes;
}

int dsu_sdf_fd_bwd_sorted(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                   const float* pts, const int32_t* perm, int64_t n, float radius, float eps,
                   uint32_t active_levels, const float* d_sdf, const float* d_grad,
                   const float* d_feature, const float* d_laplace, float* grad_table,
                   float* g_w0, float* g_b0, float* g_w1, float* g_b1, void* workspace,
                   int64_t workspace_bytes, const void* enc_cache, void* stream) {
  return dsu_sdf_fd_bwd_sorted_mid(cfg, table_f16, mlp, pts, perm, n, radius, eps, active_levels,
                                   d_sdf, d_grad, d_feature, d_laplace, grad_table, g_w0, g_b0, g_w1,
                                   g_b1, workspace, workspace_bytes, enc_cache, nullptr, stream);
}

int dsu_sdf_fd_bwd_sorted_mid(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                              const dsu_sdf_mlp* mlp, const float* pts, const int32_t* perm,
                              int64_t n, float radius, float eps, uint32_t active_levels,
                              const float* d_sdf, const float* d_grad, const float* d_feature,
                              const float* d_laplace, float* grad_table, float* g_w0, float* g_b0,
                              float* g_w1, float* g_b1, void* workspace, int64_t workspace_bytes,
                              const void* enc_cache, void* mid_event, void* stream) {
  return dsu_sdf_fd_bwd_sorted_fold(cfg, table_f16, mlp, pts, perm, n, radius, eps, active_levels,
                                    d_sdf, d_grad, d_feature, d_laplace, grad_table, g_w0, g_b0, g_w1,
                                    g_b1, workspace, workspace_bytes, enc_cache, mid_event, nullptr,
                                    stream);
}

int dsu_sdf_fd_bwd_sorted_fold(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                               const dsu_sdf_mlp* mlp, const float* pts, const int32_t* perm,
                               int64_t n, float radius, float eps, uint32_t active_levels,
                               const float* d_sdf, const float* d_grad, const float* d_feature,
                               const float* d_laplace, float* grad_table, float* g_w0, float* g_b0,
                               float* g_w1, float* g_b1, void* workspace, int64_t workspace_bytes,
                               const void* enc_cache, void* mid_event,
                               const dsu_partial_reduce* extra, void* stream) {
  if (extra && (!extra->partials || !extra->map || !extra->base || extra->nblocks < 0 ||
                extra->n <= 0 || extra->stride < extra->n))
    return DSU_EINVAL;
  if (extra && (use_valu(false) || !bwd_split() || n == 0)) return DSU_EUNSUP;
  if (use_valu(false) && perm) return DSU_EUNSUP;
  if (use_valu(false))
    return dsu_sdf_fd_bwd_valu(cfg, table_f16, mlp, pts, n, radius, eps, active_levels, d_sdf,
                               d_grad, d_feature, d_laplace, grad_table, g_w0, g_b0, g_w1, g_b1,
                               workspace, workspace_bytes, stream);
  if (!cfg || !table_f16 || !mlp || (!pts && n) || n < 0) return DSU_EINVAL;
  if (!mlp->w0 || !mlp->b0 || !mlp->w1 || !mlp->b1) return DSU_EINVAL;
  if (!grad_table || !g_w0 || !g_b0 || !g_w1 || !g_b1) return DSU_EINVAL;
  if (active_levels > cfg->n_levels || !(eps > 0.0f)) return DSU_EINVAL;
  GridMeta m;
  int rc = make_meta(cfg, &m);
  if (rc) return rc;
  // the same-cell run merge packs a cell's coordinates into 10 bits each
  for (uint32_t l = 0; l < active_levels; ++l)
    if (m.res[l] + 1u > 1024u) return DSU_EUNSUP;
  if (n == 0) return DSU_OK;
  const int64_t need = dsu_sdf_fd_bwd_workspace_bytes(cfg, n);
  if (need < 0) return (int)need;
  if (!workspace || workspace_bytes < need) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float eps2 = (float)((double)eps * (double)eps);
  // (the workspace is sized for one workgroup per CU; dsu_set_onewave_grid_cap may launch fewer)
  const int blocks = dsu_onewave_blocks(n, 256, BWD_MFMA_MAX_BLOCKS);
  const int ablate = dsu_ab_int("DSU_BWD_ABLATE", 0);
  if (bwd_split()) {
    const size_t shm1 = (size_t)BWD_CACHE_OFF * sizeof(float);         // no gradient cache
    const size_t shm2 = (size_t)SC_LDS_TOTAL * sizeof(float);
    float2* dinbuf = reinterpret_cast<float2*>((char*)workspace +
                                               (size_t)blocks * PART_STRIDE * sizeof(float));
    // (dsu_set_scatter_grid_cap launches fewer workgroups — the items are handed out dynamically, so
    // any count works; measured neutral with drawings in flight)
    int sblocks = dsu_capped_blocks(n, SC_THREADS, SC_MAXBLOCKS);
    if (dsu_scatter_grid_cap_value > 0 && sblocks > dsu_scatter_grid_cap_value) sblocks = dsu_scatter_grid_cap_value;
    DSU_DISPATCH_NL(cfg->n_levels, {
      auto k1 = enc_cache ? sdf_fd_bwd_mfma_kernel<NL, true, true>
                          : sdf_fd_bwd_mfma_kernel<NL, true, false>;
      auto k2 = sdf_fd_scatter_kernel<NL>;
      DSU_ENSURE_DYN_LDS((sdf_fd_bwd_mfma_kernel<NL, true, true>), shm1);
      DSU_ENSURE_DYN_LDS((sdf_fd_bwd_mfma_kernel<NL, true, false>), shm1);
      DSU_ENSURE_DYN_LDS(k2, shm2);
      // the lean / pipelined form of the MLP part (NL = 10, features from the cache, 4..7 active
      // levels: the optimisation's schedule); anything else, and DSU_BWD_PIPE=0 in variant builds,
      // takes the general kernel
      bool piped = false;
      if constexpr (NL == 10) {
        static int use_pipe = -1;
        if (use_pipe < 0) use_pipe = dsu_ab_int("DSU_BWD_PIPE", 1) != 0;
        if (use_pipe && enc_cache && ablate == 0)
          piped = launch_bwd_pipe<NL>(active_levels, blocks, shm1, s, *mlp, pts, n, radius, eps, eps2,
                                      d_sdf, d_grad, d_feature, d_laplace, (float*)workspace,
                                      (const __half2*)enc_cache, dinbuf, perm);
      }
      if (!piped)
        k1<<<dim3(blocks), dim3(256), shm1, s>>>(
            (const __half2*)table_f16, m, *mlp, pts, n, radius, eps, eps2, active_levels, d_sdf,
            d_grad, d_feature, d_laplace, grad_table, (float*)workspace, (const __half2*)enc_cache,
            dinbuf, perm, ablate);
      // the MLP part holds every SIMD with one 458-register wave; what a caller wants to run
      // beside the rest of the backward (two 96-register waves per SIMD) waits for this event
      if (mid_event && hipEventRecord((hipEvent_t)mid_event, s) != hipSuccess) return DSU_ELAUNCH;
      // same-cell run merge (DPP segmented sums over 16 lanes) only where neighbouring lanes of the
      // Morton order can share a cell: DSU_SC_MERGE_LEVELS (default: all levels)
      static int merge_lv = -1;
      if (merge_lv < 0) merge_lv = dsu_ab_int("DSU_SC_MERGE_LEVELS", 64);
      static int centre_acc = -1;      // DSU_SC_CENTRE=0: every evaluation emitted on its own (A/B)
      if (centre_acc < 0) centre_acc = dsu_ab_int("DSU_SC_CENTRE", 1) != 0;
      static int dense_lv = -1;        // DSU_SC_DENSE=0 (variant builds): every level through the cache
      if (dense_lv < 0) dense_lv = dsu_ab_int("DSU_SC_DENSE", 1) != 0;
      int* work_counter = reinterpret_cast<int*>(dinbuf + (size_t)7 * (size_t)n * cfg->n_levels);
      // the partial sums ride in the scatter launch (its first workgroups): no launches of their own
      const OwnReduce own{(const float*)workspace, blocks, g_w0, g_b0, g_w1, g_b1};
      dsu_partial_reduce ext{};
      if (extra) ext = *extra;
      const int n_red = OWN_RED_BLOCKS + (extra ? dsu_red::blocks_of(ext) : 0);
      k2<<<dim3(n_red + sblocks), dim3(SC_THREADS), shm2, s>>>(m, pts, n, radius, eps, active_levels,
                                                              dinbuf, grad_table, merge_lv, centre_acc,
                                                              dense_lv, work_counter, own, ext);
    });
    DSU_CHECK_LAUNCH();
    return DSU_OK;
  }
  if (mid_event && hipEventRecord((hipEvent_t)mid_event, s) != hipSuccess) return DSU_ELAUNCH;
  const size_t shm = (size_t)BWD_LDS_F * sizeof(float);
  DSU_DISPATCH_NL(cfg->n_levels, {
    auto k0 = enc_cache ? sdf_fd_bwd_mfma_kernel<NL, false, true>
                        : sdf_fd_bwd_mfma_kernel<NL, false, false>;
    DSU_ENSURE_DYN_LDS((sdf_fd_bwd_mfma_kernel<NL, false, true>), shm);
    DSU_ENSURE_DYN_LDS((sdf_fd_bwd_mfma_kernel<NL, false, false>), shm);
    k0<<<dim3(blocks), dim3(256), shm, s>>>(
        (const __half2*)table_f16, m, *mlp, pts, n, radius, eps, eps2, active_levels, d_sdf,
        d_grad, d_feature, d_laplace, grad_table, (float*)workspace, (const __half2*)enc_cache,
        nullptr, perm, ablate);
    reduce_partials_mfma_kernel<NL><<<dim3(OWN_RED_BLOCKS), dim3(1024), 0, s>>>(
        OwnReduce{(const float*)workspace, blocks, g_w0, g_b0, g_w1, g_b1});
  });
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_sdf_fd_bwd_cached(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                          const dsu_sdf_mlp* mlp, const float* pts, int64_t n, float radius,
                          float eps, uint32_t active_levels, const float* d_sdf,
                          const float* d_grad, const float* d_feature, const float* d_laplace,
                          float* grad_table, float* g_w0, float* g_b0, float* g_w1, float* g_b1,
                          void* workspace, int64_t workspace_bytes, const void* enc_cache,
                          void* stream) {
  return dsu_sdf_fd_bwd_sorted(cfg, table_f16, mlp, pts, nullptr, n, radius, eps, active_levels,
                               d_sdf, d_grad, d_feature, d_laplace, grad_table, g_w0, g_b0, g_w1,
                               g_b1, workspace, workspace_bytes, enc_cache, stream);
}

int dsu_sdf_fd_bwd(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                   const float* pts, int64_t n, float radius, float eps,
                   uint32_t active_levels, const float* d_sdf, const float* d_grad,
                   const float* d_feature, const float* d_laplace, float* grad_table,
                   float* g_w0, float* g_b0, float* g_w1, float* g_b1, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  return dsu_sdf_fd_bwd_sorted(cfg, table_f16, mlp, pts, nullptr, n, radius, eps, active_levels,
                               d_sdf, d_grad, d_feature, d_laplace, grad_table, g_w0, g_b0, g_w1,
                               g_b1, workspace, workspace_bytes, nullptr, stream);
}

#ifdef DSU_BWD_PROF
int dsu_debug_bwd_prof(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(dsu_bwd_prof), 16 * sizeof(unsigned long long)) != hipSuccess)
    return DSU_ELAUNCH;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(dsu_bwd_prof), z, sizeof(z)) != hipSuccess) return DSU_ELAUNCH;
  }
  return DSU_OK;
}
#endif

}  // extern "C"
