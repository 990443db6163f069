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
//
// The same kernel is the GEMM of the UNet's linear layers (a 1x1 "convolution" over M rows):
// dsu_gemm_f16_fwd (to_q / to_k / to_out, FeedForward output, proj_in / proj_out, the embedding
// MLPs; mvdiffusion/models/transformer_mv2d.py:447-483, unet_mv2d_condition.py:313-319,374),
// with two extra epilogues selected by the MODE template parameter:
//   MODE 1  GEGLU (diffusers FeedForward's first layer): the 128 weight rows of a tile are 2 x 32
//           value rows + 2 x 32 gate rows per wave pair, so a lane holds value and gate of the
//           SAME (row, channel) in matching accumulator registers and the epilogue writes
//           (value + b_v) * gelu(gate + b_g): the (M, 8C) intermediate never exists.
//   transposed output (to_v): out[(image, channel, token)] so that the attention kernel reads V^T
//           rows directly.
#include "common.h"

#include <stdlib.h>

namespace {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM_BIG = 128;   // output channels per workgroup (2 x 2 form)
constexpr int TN_BIG = 128;   // pixels per workgroup
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
  int split_k;          // > 1: blockIdx.z owns a contiguous range of K chunks, f32 partials -> ws
  float* ws;            // (split_k, npix, O) f32 when split_k > 1
  int out_t;            // 1: out[(n * O + o) * (OH*OW) + pixel-in-image]  (V^T for the attention kernel)
  int* counters;        // split_k > 1: one zeroed int per output tile -> the LAST workgroup of a
                        // tile sums the partials and runs the epilogue itself (no reduce launch)
};

// WI x WJ: 32x32 MFMA tiles per wave along output channels x pixels.  2 x 2 (the 128 x 128
// workgroup tile) for layers that fill the chip; 1 x 1 (64 x 64 tile, 37 KB of LDS: four workgroups
// = four waves per SIMD on a CU) for the many layers of the UNet whose 128 x 128 tiling leaves
// most CUs with one wave per SIMD or none (M = 768 ... 3072 rows).
template <int MODE, int WI, int WJ>
__global__ __launch_bounds__(256) void conv_f16_kernel(CArgs a) {
  constexpr bool GEGLU = MODE == 1;
  static_assert(!GEGLU || (WI == 2 && WJ == 2), "the GEGLU row pairing is written for 2 x 2");
  constexpr int TM = 64 * WI, TN = 64 * WJ;       // workgroup tile
  constexpr int RA = 2 * WI, RB = 2 * WJ;         // 16-byte staging rows per thread per chunk
  __shared__ __attribute__((aligned(16))) f16 sA[2][TM * ROW];
  __shared__ __attribute__((aligned(16))) f16 sB[2][TN * ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;      // 2x2 waves, (32 WI) x (32 WJ) each
  // GEGLU: a tile covers 64 output channels (value rows + gate rows of the 2*O-row weight)
  const int o0 = blockIdx.y * (GEGLU ? TM / 2 : TM);
  const int64_t npix = (int64_t)a.B * a.OH * a.OW;
  const int64_t p0 = (int64_t)blockIdx.x * TN;
  const int C8 = a.C >> 3;                      // 8-channel groups per tap
  const int IH = a.up2 ? a.H * 2 : a.H, IW = a.up2 ? a.W * 2 : a.W;  // logical input size

  // staging assignment: 4 A rows + 4 B rows (16 bytes each) per thread per chunk.  All four rows
  // of a thread sit at the same k offset (256 % 8 == 0), so (tap, channel) is ONE running state
  // per thread; everything that does not change along K is folded into per-row constants, and
  // offsets are 32-bit (the host rejects tensors of 2^31 elements or more).  The first version
  // recomputed two integer divisions, bounds and 64-bit addresses per row per chunk:
  // ~400 VALU instructions per 16 MFMAs (PMC: SQ_INSTS_VALU / SQ_INSTS_MFMA = 25, 55 % of wave
  // cycles in VALU issue) — the kernel was VALU-bound, not MFMA-, LDS- or latency-bound.
  int a_row[RA], b_row[RB];
  const int grp = tid & 7;
  const int kofs = grp * 8;
  uint32_t w_off[RA], pix_base[RB];
  int iy0[RB], ix0[RB];
  bool a_ok[RA], b_ok[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int idx = tid + 256 * i;
    a_row[i] = idx >> 3;
    int o = o0 + a_row[i];
    int wrow = o;
    if (GEGLU) {
      // tile row r = wm*64 + i*32 + l: i = 0 value rows, i = 1 gate rows of channel o0 + wm*32 + l
      const int r = a_row[i];
      o = o0 + (r >> 6) * 32 + (r & 31);
      wrow = ((r >> 5) & 1) ? a.O + o : o;
    }
    a_ok[i] = o < a.O;
    w_off[i] = (uint32_t)(a_ok[i] ? wrow : 0) * (uint32_t)a.Ktot + (uint32_t)kofs;
  }
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int idx = tid + 256 * i;
    b_row[i] = idx >> 3;
    const int64_t p = p0 + b_row[i];
    b_ok[i] = p < npix;
    const int64_t pp = b_ok[i] ? p : 0;
    const int n = (int)(pp / (a.OH * a.OW));
    const int rem = (int)(pp % (a.OH * a.OW));
    iy0[i] = (rem / a.OW) * a.stride - a.pad;
    ix0[i] = (rem % a.OW) * a.stride - a.pad;
    pix_base[i] = (uint32_t)n * (uint32_t)(a.H * a.W) * (uint32_t)a.C;
  }
  const bool kfast = (a.C % BK) == 0;             // every UNet/VAE layer but conv_in (C = 8)
  int st_tap = 0, st_c = 0, st_chunk = -2;

#ifdef DSU_CONV_PREFETCH1
  f16x8 ra0[RA], rb0[RB];
  uint32_t ok0 = 0;
#else
  f16x8 ra0[RA], rb0[RB], ra1[RA], rb1[RB];
  uint32_t ok0 = 0, ok1 = 0;
#endif
  // Every load is issued unconditionally (rows / taps outside the problem read element 0 and are
  // zeroed when the chunk is written to LDS): with a branch per load the compiler cannot count how
  // many loads follow a given one and waits for ALL of them (vmcnt(0)) in front of the first LDS
  // write — including the chunk that was requested a moment ago.
  auto load_regs = [&](int chunk, f16x8 (&ra)[RA], f16x8 (&rb)[RB], uint32_t& okmask) {
    if (kfast && chunk == st_chunk + 1) {
      st_c += BK;
      if (st_c >= a.C) { st_c -= a.C; ++st_tap; }
    } else {
      const int kg = (chunk * BK + kofs) >> 3;
      st_tap = kg / C8;
      st_c = (kg - st_tap * C8) << 3;
    }
    st_chunk = chunk;
    const bool kin = chunk * BK + kofs < a.Ktot;
    int ty, tx;
    if (a.KS == 3) { ty = (st_tap * 11) >> 5; tx = st_tap - 3 * ty; }       // tap < 9
    else if (a.KS == 1) { ty = 0; tx = 0; }
    else { ty = st_tap / a.KS; tx = st_tap - ty * a.KS; }
    const uint32_t kadv = (uint32_t)(chunk * BK);
    uint32_t okm = 0;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const bool oka = kin && a_ok[i];
      ra[i] = *reinterpret_cast<const f16x8*>(a.w + (oka ? w_off[i] + kadv : 0u));
      okm |= (oka ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int iy = iy0[i] + ty, ix = ix0[i] + tx;
      const bool okb = kin && b_ok[i] && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
      if (a.up2) { iy >>= 1; ix >>= 1; }
      const uint32_t off = pix_base[i] + (uint32_t)((iy * a.W + ix) * a.C + st_c);
      rb[i] = *reinterpret_cast<const f16x8*>(a.in + (okb ? off : 0u));
      okm |= (okb ? 1u : 0u) << (4 + i);
    }
    okmask = okm;
  };
  auto store_lds = [&](int buf, const f16x8 (&ra)[RA], const f16x8 (&rb)[RB], uint32_t okmask) {
    f16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (f16)0.0f;
#pragma unroll
    for (int i = 0; i < RA; ++i)
      *reinterpret_cast<f16x8*>(&sA[buf][a_row[i] * ROW + kofs]) = (okmask >> i) & 1u ? ra[i] : z;
#pragma unroll
    for (int i = 0; i < RB; ++i)
      *reinterpret_cast<f16x8*>(&sB[buf][b_row[i] * ROW + kofs]) = (okmask >> (4 + i)) & 1u ? rb[i] : z;
  };

  f32x16 acc[WI][WJ];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // K range of this workgroup (split-K: small-spatial levels have too few output tiles to fill
  // 256 CUs, so the 9*C reduction is spread over blockIdx.z and summed by conv_f16_reduce_kernel)
  const int total_chunks = (a.Ktot + BK - 1) / BK;
  const int per = (total_chunks + a.split_k - 1) / a.split_k;
  const int ch0 = blockIdx.z * per;
  const int nchunks = min(total_chunks, ch0 + per);
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f16x8 af[WI], bf[WJ];
#pragma unroll
      for (int i = 0; i < WI; ++i)
        af[i] = *reinterpret_cast<const f16x8*>(
            &sA[buf][(wm * 32 * WI + i * 32 + l31) * ROW + ks * 16 + hh * 8]);
#pragma unroll
      for (int j = 0; j < WJ; ++j)
        bf[j] = *reinterpret_cast<const f16x8*>(
            &sB[buf][(wn * 32 * WJ + j * 32 + l31) * ROW + ks * 16 + hh * 8]);
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
#ifdef DSU_CONV_PREFETCH1
  // (A/B variant: the round-2 pipeline — one chunk ahead)
  load_regs(ch0, ra0, rb0, ok0);
  store_lds(ch0 & 1, ra0, rb0, ok0);
  __syncthreads();
  for (int ch = ch0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) load_regs(ch + 1, ra0, rb0, ok0);
    compute(buf);
    if (ch + 1 < nchunks) store_lds(buf ^ 1, ra0, rb0, ok0);
    __syncthreads();
  }
#else
  // Two chunks ahead: the global loads of chunk k + 2 are issued before chunk k's MFMAs and written
  // to LDS behind chunk k + 1's.  One chunk is 16 MFMAs per wave (~0.5 k clocks); one chunk ahead,
  // every iteration ended on the tail of an L2 / HBM round trip (~1-2 k clocks).  Chunk k (counted
  // from this workgroup's first) lives in LDS buffer k & 1 and register set k & 1, hence the loop
  // over pairs with the sets named at compile time (an exit between the two halves made the
  // compiler keep two copies of the accumulators; a set chosen by a branch, copies of the sets).
  const int nk = nchunks - ch0;
  if (nk > 0) {
    load_regs(ch0, ra0, rb0, ok0);
    store_lds(0, ra0, rb0, ok0);
    if (nk > 1) load_regs(ch0 + 1, ra1, rb1, ok1);
  }
  __syncthreads();
  int k = 0;
  for (; k + 1 < nk; k += 2) {                       // whole pairs: chunk k in buffer / set 0
    // (requested even past this workgroup's last chunk — never written to LDS then: a branch around
    // the loads would again hide from the compiler how many are in flight)
    load_regs(ch0 + k + 2, ra0, rb0, ok0);
    compute(0);
    store_lds(1, ra1, rb1, ok1);
    __syncthreads();
    load_regs(ch0 + k + 3, ra1, rb1, ok1);
    compute(1);
    if (k + 2 < nk) store_lds(0, ra0, rb0, ok0);
    __syncthreads();
  }
  if (k < nk) compute(0);                            // odd count: the last chunk is already in buffer 0
#endif

  // epilogue: lane -> pixel (wn*32*WJ + j*32 + l31); register quad r4 -> channels
  //   o = o0 + wm*32*WI + i*32 + 8*r4 + 4*hh + {0..3}
  if (a.split_k > 1) {
    float* ws = a.ws + (size_t)blockIdx.z * npix * a.O;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int64_t p = p0 + wn * 32 * WJ + j * 32 + l31;
      if (p >= npix) continue;
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int o = o0 + wm * 32 * WI + i * 32 + 8 * r4 + 4 * hh;
          if (o + 3 < a.O) {
            *reinterpret_cast<float4*>(ws + (size_t)p * a.O + o) =
                make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2],
                            acc[i][j][4 * r4 + 3]);
          } else {
            for (int e = 0; e < 4 && o + e < a.O; ++e) ws[(size_t)p * a.O + o + e] = acc[i][j][4 * r4 + e];
          }
        }
    }
    if (a.counters == nullptr) return;              // conv_f16_reduce_kernel finishes the job
    // Fix-up by the last arrival: every workgroup of a tile publishes its partial sums (device-scope
    // release: the eight XCDs have private L2s), takes a ticket, and the one that draws the last
    // ticket adds the split_k partials in z order (the order does not depend on who is last) and
    // runs the epilogue.  The UNet forward issued ~180 reduce launches (7 us + a launch gap each,
    // 1.2 ms of 13) for its split-K layers.
    __shared__ int s_ticket;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
#ifdef DSU_CONV_THREADFENCE
    // (rounds 3-5, A/B variant: a device-scope fence by every thread on both sides)
    __threadfence();
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&a.counters[tile], 1);
    __syncthreads();
    if (s_ticket != a.split_k - 1) return;
    __threadfence();
    if (tid == 0) a.counters[tile] = 0;             // ready for the next launch on this stream
#else
    // publish: every wave's slab stores have left the wave (vmcnt), ONE agent-scope release (the
    // write-back of this XCD's L2 is not per wave), then the relaxed ticket; the last arriver does
    // ONE agent-scope acquire for the workgroup.  A __threadfence() per thread on both sides is the
    // same ordering at several times the cost (a write-back + invalidate per wave).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      s_ticket = __hip_atomic_fetch_add(&a.counters[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_ticket != a.split_k - 1) return;
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      a.counters[tile] = 0;                         // ready for the next launch on this stream
    }
    __syncthreads();
#endif
    const int hw = a.OH * a.OW;
    for (int idx = tid; idx < TN * (TM / 4); idx += 256) {
      const int64_t p = p0 + idx / (TM / 4);
      const int o = o0 + (idx % (TM / 4)) * 4;
      if (p >= npix || o >= a.O) continue;
      const int n = (int)(p / hw);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const bool full = o + 3 < a.O && (a.O & 3) == 0;
      for (int z = 0; z < a.split_k; ++z) {
        const float* w = a.ws + ((size_t)z * npix + p) * a.O + o;
        if (full) {
          const float4 q = *reinterpret_cast<const float4*>(w);
          v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
        } else {
          for (int e = 0; e < 4 && o + e < a.O; ++e) v[e] += w[e];
        }
      }
      for (int e = 0; e < 4 && o + e < a.O; ++e) {
        float tv = v[e];
        if (a.bias) tv += (float)a.bias[o + e];
        if (a.addvec) tv += (float)a.addvec[(size_t)n * a.O + o + e];
        if (a.residual) tv += (float)a.residual[(size_t)p * a.O + o + e];
        a.out[(size_t)p * a.O + o + e] = (f16)tv;
      }
    }
    return;
  }
  if constexpr (GEGLU) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t p = p0 + wn * 32 * WJ + j * 32 + l31;
      if (p >= npix) continue;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int o = o0 + wm * 32 + 8 * r4 + 4 * hh;
        if (o >= a.O) continue;                                  // O % 4 == 0 (checked by the host)
        f16x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float val = acc[0][j][4 * r4 + e], gate = acc[1][j][4 * r4 + e];
          if (a.bias) { val += (float)a.bias[o + e]; gate += (float)a.bias[a.O + o + e]; }
          // the f16 rounding of the projection the unfused path stored is kept
          val = (float)(f16)val; gate = (float)(f16)gate;
          ov[e] = (f16)(val * (0.5f * gate * (1.0f + erff(gate * 0.70710678118654752f))));
        }
        *reinterpret_cast<f16x4*>(a.out + (size_t)p * a.O + o) = ov;
      }
    }
    return;
  }
  if (a.out_t) {
    const int hw = a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int64_t p = p0 + wn * 32 * WJ + j * 32 + l31;
      if (p >= npix) continue;
      const int n = (int)(p / hw);
      const int t = (int)(p - (int64_t)n * hw);
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + wm * 32 * WI + i * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
          if (o >= a.O) continue;
          float v = acc[i][j][r];
          if (a.bias) v += (float)a.bias[o];
          a.out[((size_t)n * a.O + o) * hw + t] = (f16)v;        // 32 lanes = 32 consecutive tokens
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int64_t p = p0 + wn * 32 * WJ + j * 32 + l31;
    if (p >= npix) continue;
    const int n = (int)(p / (a.OH * a.OW));
#pragma unroll
    for (int i = 0; i < WI; ++i) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int o = o0 + wm * 32 * WI + i * 32 + 8 * r4 + 4 * hh;
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

// out[p][o] = f16(sum_z ws[z][p][o] + bias[o] + addvec[n][o] + residual[p][o]); 4 channels per thread
__global__ __launch_bounds__(256) void conv_f16_reduce_kernel(CArgs a, int64_t npix) {
  const int O4 = (a.O + 3) >> 2;
  const int64_t total = npix * O4;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / O4;
    const int o = (int)(t - p * O4) * 4;
    const int n = (int)(p / (a.OH * a.OW));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = o + 3 < a.O && (a.O & 3) == 0;
    for (int z = 0; z < a.split_k; ++z) {
      const float* w = a.ws + ((size_t)z * npix + p) * a.O + o;
      if (full) {
        const float4 q = *reinterpret_cast<const float4*>(w);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      } else {
        for (int e = 0; e < 4 && o + e < a.O; ++e) v[e] += w[e];
      }
    }
    for (int e = 0; e < 4 && o + e < a.O; ++e) {
      float tv = v[e];
      if (a.bias) tv += (float)a.bias[o + e];
      if (a.addvec) tv += (float)a.addvec[(size_t)n * a.O + o + e];
      if (a.residual) tv += (float)a.residual[(size_t)p * a.O + o + e];
      a.out[(size_t)p * a.O + o + e] = (f16)tv;
    }
  }
}

int conv_out_dim(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

// 64 x 64 workgroup tiles when the 128 x 128 tiling would leave the chip under-filled
// (DSU_CONV_SMALL_MAX_TILES: A/B switch, 0 = always 128 x 128)
bool small_tiles(int64_t npix, int64_t O) {
  static const int64_t max_big = dsu_ab_int("DSU_CONV_SMALL_MAX_TILES", 512);
  const int64_t big = ((npix + TN_BIG - 1) / TN_BIG) * ((O + TM_BIG - 1) / TM_BIG);
  return big < max_big;
}

template <int WI, int WJ>
void launch_conv(const CArgs& a, int64_t npix, hipStream_t s) {
  constexpr int TM = 64 * WI, TN = 64 * WJ;
  dim3 grid((unsigned)((npix + TN - 1) / TN), (unsigned)((a.O + TM - 1) / TM), (unsigned)a.split_k);
  conv_f16_kernel<0, WI, WJ><<<grid, 256, 0, s>>>(a);
}

}  // namespace

extern "C" {

int32_t dsu_conv2d_nhwc_f16_split_k(int32_t B, int32_t H, int32_t W, int32_t C, int32_t O,
                                    int32_t k, int32_t stride, int32_t pad, int32_t upsample2x) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || O <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return 1;
  const int IH = upsample2x ? 2 * H : H, IW = upsample2x ? 2 * W : W;
  const int OH = conv_out_dim(IH, k, stride, pad), OW = conv_out_dim(IW, k, stride, pad);
  if (OH <= 0 || OW <= 0) return 1;
  const int64_t npix = (int64_t)B * OH * OW;
  const int tile = small_tiles(npix, O) ? 64 : 128;
  const int64_t tiles = ((npix + tile - 1) / tile) * ((O + tile - 1) / tile);
  const int chunks = (k * k * C + BK - 1) / BK;
  const int64_t fill = tile == 64 ? 512 : 256;       // 64-tiles: four workgroups fit a CU
  if (tiles >= fill || chunks < 8) return 1;         // the output tiles already fill the chip
  static const int64_t target_pct = dsu_ab_int("DSU_CONV_SPLIT_TARGET", 50);   // A/B: percent of fill
  if (target_pct <= 0) return 1;
  // (measured on the UNet forward: 12.2 ms without split-K, 11.1 / 10.8 / 10.6 ms aiming at 2 x / 1 x /
  // 0.5 x fill — the f32 partials of a split cost a write and a read of the whole output each)
  int64_t s = (fill * target_pct / 100 + tiles - 1) / tiles;
  if (s > chunks / 4) s = chunks / 4;                // keep >= 4 chunks (256 k) per workgroup
  if (s > 16) s = 16;
  return s < 2 ? 1 : (int32_t)s;
}

int64_t dsu_conv2d_nhwc_f16_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t O, int32_t k,
                                            int32_t stride, int32_t pad, int32_t upsample2x,
                                            int32_t split_k) {
  if (split_k <= 1) return 0;
  const int IH = upsample2x ? 2 * H : H, IW = upsample2x ? 2 * W : W;
  const int OH = conv_out_dim(IH, k, stride, pad), OW = conv_out_dim(IW, k, stride, pad);
  if (OH <= 0 || OW <= 0) return DSU_EINVAL;
  return (int64_t)split_k * B * OH * OW * O * (int64_t)sizeof(float);
}

int dsu_conv2d_nhwc_f16_fwd_ws(const void* input, const void* weight_okc, const void* bias,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t O, int32_t k,
                               int32_t stride, int32_t pad, int32_t upsample2x,
                               const void* addvec, const void* residual, void* out,
                               int32_t split_k, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  return dsu_conv2d_nhwc_f16_fwd_fx(input, weight_okc, bias, B, H, W, C, O, k, stride, pad, upsample2x,
                                    addvec, residual, out, split_k, workspace, workspace_bytes,
                                    nullptr, 0, stream);
}

int dsu_conv2d_nhwc_f16_fwd_fx(const void* input, const void* weight_okc, const void* bias,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t O, int32_t k,
                               int32_t stride, int32_t pad, int32_t upsample2x,
                               const void* addvec, const void* residual, void* out,
                               int32_t split_k, void* workspace, int64_t workspace_bytes,
                               int32_t* tile_counters, int64_t n_counters, void* stream) {
  if (!input || !weight_okc || !out) return DSU_EINVAL;
  if (split_k < 1 || split_k > 64) return DSU_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || O <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return DSU_EINVAL;
  if (C % 8 != 0) return DSU_EUNSUP;           // 16-byte channel groups
  if ((int64_t)B * H * W * C >= (int64_t)1 << 31 || (int64_t)O * k * k * C >= (int64_t)1 << 31)
    return DSU_EUNSUP;                         // 32-bit element offsets inside the kernel
  CArgs a;
  a.in = (const f16*)input; a.w = (const f16*)weight_okc; a.bias = (const f16*)bias;
  a.addvec = (const f16*)addvec; a.residual = (const f16*)residual; a.out = (f16*)out;
  a.B = B; a.H = H; a.W = W; a.C = C; a.O = O; a.KS = k; a.stride = stride; a.pad = pad;
  a.up2 = upsample2x ? 1 : 0;
  a.out_t = 0; a.counters = nullptr;
  const int IH = a.up2 ? 2 * H : H, IW = a.up2 ? 2 * W : W;
  a.OH = (IH + 2 * pad - k) / stride + 1;
  a.OW = (IW + 2 * pad - k) / stride + 1;
  if (a.OH <= 0 || a.OW <= 0) return DSU_EINVAL;
  a.Ktot = k * k * C;
  const int64_t npix = (int64_t)B * a.OH * a.OW;
  const int chunks = (a.Ktot + BK - 1) / BK;
  if (split_k > chunks) split_k = chunks;
  a.split_k = split_k;
  a.ws = (float*)workspace;
  if (split_k > 1) {
    // every z must own at least one chunk: shrink split_k until the last range is non-empty
    while (split_k > 1 && (split_k - 1) * ((chunks + split_k - 1) / split_k) >= chunks) --split_k;
    a.split_k = split_k;
  }
  if (a.split_k > 1 &&
      (!workspace || workspace_bytes < (int64_t)a.split_k * npix * O * (int64_t)sizeof(float)))
    return DSU_EINVAL;
  const bool small = small_tiles(npix, O);
  const int tile = small ? 64 : 128;
  const int64_t n_tiles = ((npix + tile - 1) / tile) * ((O + tile - 1) / tile);
  a.counters = (a.split_k > 1 && tile_counters && n_tiles <= n_counters) ? tile_counters : nullptr;
  if (small) launch_conv<1, 1>(a, npix, (hipStream_t)stream);
  else launch_conv<2, 2>(a, npix, (hipStream_t)stream);
  if (a.split_k > 1 && !a.counters) {
    const int64_t total = npix * ((O + 3) / 4);
    conv_f16_reduce_kernel<<<dsu_capped_blocks(total, 256, 2048), 256, 0, (hipStream_t)stream>>>(
        a, npix);
  }
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

// ---- GEMM entry points (the UNet's linear layers): x (M, K) f16 row-major, w (N, K) f16 (the
// nn.Linear layout), f32 accumulation.
static int gemm_args(CArgs& a, const void* x, const void* w, const void* bias, int64_t M, int32_t K,
                     int32_t N, const void* residual, void* out) {
  if (!x || !w || !out) return DSU_EINVAL;
  if (M <= 0 || K <= 0 || N <= 0) return DSU_EINVAL;
  if (K % 8 != 0) return DSU_EUNSUP;
  if (M * (int64_t)K >= (int64_t)1 << 31 || (int64_t)N * K >= (int64_t)1 << 30 ||
      M * (int64_t)N >= (int64_t)1 << 31)
    return DSU_EUNSUP;
  a.in = (const f16*)x; a.w = (const f16*)w; a.bias = (const f16*)bias; a.addvec = nullptr;
  a.residual = (const f16*)residual; a.out = (f16*)out;
  a.B = 1; a.H = (int)M; a.W = 1; a.C = K; a.O = N; a.OH = (int)M; a.OW = 1;
  a.KS = 1; a.stride = 1; a.pad = 0; a.up2 = 0; a.Ktot = K; a.split_k = 1; a.ws = nullptr;
  a.out_t = 0; a.counters = nullptr;
  return DSU_OK;
}

int32_t dsu_gemm_f16_split_k(int64_t M, int32_t K, int32_t N) {
  if (M <= 0 || M >= ((int64_t)1 << 31) || K <= 0 || N <= 0) return 1;
  return dsu_conv2d_nhwc_f16_split_k(1, (int32_t)M, 1, K, N, 1, 1, 0, 0);
}

int64_t dsu_gemm_f16_workspace_bytes(int64_t M, int32_t N, int32_t split_k) {
  return split_k <= 1 ? 0 : (int64_t)split_k * M * N * (int64_t)sizeof(float);
}

/* out (M, N) = x w^T (+ bias) (+ residual (M, N)).  tokens_per_image > 0: the output is written
 * TRANSPOSED per image, out[(m / tokens) * N * tokens + n * tokens + m % tokens]  (to_v -> V^T). */
int dsu_gemm_f16_fwd(const void* x, const void* w, const void* bias, int64_t M, int32_t K, int32_t N,
                     const void* residual, void* out, int32_t tokens_per_image, int32_t split_k,
                     void* workspace, int64_t workspace_bytes, void* stream) {
  return dsu_gemm_f16_fwd_fx(x, w, bias, M, K, N, residual, out, tokens_per_image, split_k, workspace,
                             workspace_bytes, nullptr, 0, stream);
}

int dsu_gemm_f16_fwd_fx(const void* x, const void* w, const void* bias, int64_t M, int32_t K,
                        int32_t N, const void* residual, void* out, int32_t tokens_per_image,
                        int32_t split_k, void* workspace, int64_t workspace_bytes,
                        int32_t* tile_counters, int64_t n_counters, void* stream) {
  CArgs a;
  int rc = gemm_args(a, x, w, bias, M, K, N, residual, out);
  if (rc) return rc;
  if (tokens_per_image > 0) {
    if (M % tokens_per_image != 0 || residual || split_k > 1) return DSU_EINVAL;
    a.out_t = 1;
    a.B = (int)(M / tokens_per_image); a.H = tokens_per_image; a.OH = tokens_per_image;
  }
  const int chunks = (K + BK - 1) / BK;
  if (split_k < 1 || split_k > 64) return DSU_EINVAL;
  if (split_k > chunks) split_k = chunks;
  while (split_k > 1 && (split_k - 1) * ((chunks + split_k - 1) / split_k) >= chunks) --split_k;
  a.split_k = split_k;
  a.ws = (float*)workspace;
  if (split_k > 1 && (!workspace || workspace_bytes < (int64_t)split_k * M * N * (int64_t)sizeof(float)))
    return DSU_EINVAL;
  const bool small = small_tiles(M, N);
  const int tile = small ? 64 : 128;
  const int64_t n_tiles = ((M + tile - 1) / tile) * ((N + tile - 1) / tile);
  a.counters = (split_k > 1 && tile_counters && n_tiles <= n_counters) ? tile_counters : nullptr;
  if (small) launch_conv<1, 1>(a, M, (hipStream_t)stream);
  else launch_conv<2, 2>(a, M, (hipStream_t)stream);
  if (split_k > 1 && !a.counters) {
    const int64_t total = M * ((N + 3) / 4);
    conv_f16_reduce_kernel<<<dsu_capped_blocks(total, 256, 2048), 256, 0, (hipStream_t)stream>>>(a, M);
  }
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

/* diffusers FeedForward first layer + GEGLU in one pass: w (2N, K), bias (2N) or null;
 * out (M, N) = (x w[:N]^T + b[:N]) * gelu(x w[N:]^T + b[N:])   (exact erf GELU). */
int dsu_gemm_geglu_fwd(const void* x, const void* w, const void* bias, int64_t M, int32_t K,
                       int32_t N, void* out, void* stream) {
  CArgs a;
  int rc = gemm_args(a, x, w, bias, M, K, N, nullptr, out);
  if (rc) return rc;
  if (N % 4 != 0 || (int64_t)2 * N * K >= (int64_t)1 << 31) return DSU_EUNSUP;
  dim3 grid((unsigned)((M + TN_BIG - 1) / TN_BIG), (unsigned)((N + TM_BIG / 2 - 1) / (TM_BIG / 2)), 1);
  conv_f16_kernel<1, 2, 2><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

int dsu_conv2d_nhwc_f16_fwd(const void* input, const void* weight_okc, const void* bias,
                            int32_t B, int32_t H, int32_t W, int32_t C, int32_t O, int32_t k,
                            int32_t stride, int32_t pad, int32_t upsample2x, const void* addvec,
                            const void* residual, void* out, void* stream) {
  return dsu_conv2d_nhwc_f16_fwd_ws(input, weight_okc, bias, B, H, W, C, O, k, stride, pad,
                                    upsample2x, addvec, residual, out, 1, nullptr, 0, stream);
}

}  // extern "C"
