// Multi-view / cross-domain attention forward for gfx950 (f16 in, f32 accumulate, f16 out).
//
// Replaces xformers.ops.memory_efficient_attention (xformers==0.0.17, un-vendored) at its two
// call sites in the reference:
//   2_charactor_reconstructor/mvdiffusion/models/transformer_mv2d.py:802  (XFormersMVAttnProcessor)
//   2_charactor_reconstructor/mvdiffusion/models/transformer_mv2d.py:890  (XFormersJointAttnProcessor)
// The reference materialises K/V for every query batch (my_repeat x6 at :785-786, cat x2 at
// :878-883).  Here the key/value sequence of a query batch is described as S SEGMENTS of
// `seg_len` tokens, each living at a (batch) index given by a small table, so the shared K/V
// are read in place (6x / 2x less K/V traffic) — same arithmetic, no copies.
//
// Layouts (element strides, f16):
//   Q, O : [batch][token][head][d]          (the to_q output / to_out input layout)
//   K    : [batch][token][head][d]
//   Vt   : [batch][head][d][token]          (V produced already transposed per head, so that
//                                            the P.V MFMA operand reads 4 consecutive keys)
// Work split: one workgroup (4 waves) = 128 queries of one (batch, head); each wave owns 32
// queries.  S^T = K.Q^T is computed with mfma_f32_32x32x16_f16 so that every lane holds ONE
// query column (its 16+16 key scores per 32-key tile are in-lane; the row max/sum needs a single
// exchange with lane^32).  O^T = V^T.P^T reuses those registers directly as the B operand: the
// k-index <-> key permutation implied by the accumulator layout is applied to the V^T reads
// instead of shuffling P.
#include "common.h"

namespace {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QBLK = 32;     // queries per wave
constexpr int WAVES = 4;
constexpr int KVBLK = 64;    // keys per LDS tile

struct AttnArgs {
  const f16* q;
  const f16* k;
  const f16* vt;
  f16* o;
  const int32_t* seg_batch;  // [Bq][S] source batch of each K/V segment
  int Bq, H, Nq, S, seg_len;
  int64_t q_bs, q_ts, q_hs;    // strides in elements
  int64_t k_bs, k_ts, k_hs;
  int64_t vt_bs, vt_hs, vt_ds; // Vt[batch][head][d][token]: token stride 1
  int64_t o_bs, o_ts, o_hs;
  float scale_log2e;           // softmax scale * log2(e)
};

// D = head dim, DP = D padded to a multiple of 16 (QK^T k-chunks), DT = ceil(D/32) output tiles
template <int D>
struct Cfg {
  static constexpr int DP = (D + 15) / 16 * 16;
  static constexpr int KCH = DP / 16;
  static constexpr int DT = (D + 31) / 32;
};

// K tile rows are padded so that 16 consecutive rows fall on 16 distinct 16-byte LDS slots
// (ds_read_b128 is serviced in 16-lane groups over a 256-byte bank row).
constexpr int krow_bytes(int dp) {
  // smallest stride >= dp*2 bytes with stride % 256 in {16, 48, 80, 112, ...} (16 * odd)
  int s = dp * 2;
  while (true) {
    int m = s % 256;
    if (m % 32 == 16) return s;
    s += 16;
  }
}

template <int D>
__global__ __launch_bounds__(256) void mv_attention_kernel(AttnArgs a) {
  constexpr int DP = Cfg<D>::DP, KCH = Cfg<D>::KCH, DT = Cfg<D>::DT;
  constexpr int KROW = krow_bytes(DP) / 2;         // halfs per K-tile row
  constexpr int VROW = KVBLK + 4;                  // halfs per V^T-tile row (68: bank-spread)
  __shared__ __attribute__((aligned(16))) f16 sK[KVBLK * KROW];
  __shared__ __attribute__((aligned(16))) f16 sV[DT * 32 * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / a.H, h = bh % a.H;
  const int q0 = blockIdx.x * (QBLK * WAVES) + wave * QBLK;
  const int qi = q0 + l31;
  const bool qv = qi < a.Nq;

  // ---- Q^T fragments (B operand of K.Q^T): lane holds Q[qi][16c + 8*hh + 0..7]
  f16x8 qf[KCH];
  {
    const f16* qp = a.q + b * a.q_bs + (int64_t)(qv ? qi : 0) * a.q_ts + h * a.q_hs;
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
      const int d = 16 * c + 8 * hh;   // D % 8 == 0: an 8-chunk is all data or all padding
      f16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (f16)0.0f;
      qf[c] = (qv && d < D) ? *reinterpret_cast<const f16x8*>(qp + d) : z;
    }
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int n = 0; n < DT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.0f;
  float m_run = -1e30f, l_run = 0.0f;

  // staging roles are fixed per thread (only the tile origin moves): the row/column split of
  // every staged 16-/8-byte group is computed once here, not per tile (the kernel issues ~45
  // VALU instructions per MFMA at head dim 40 - softmax and staging arithmetic, not the matrix
  // pipe, bound it)
  constexpr int NK = (KVBLK * (DP / 8) + 255) / 256;
  constexpr int NV = (DT * 32 * (KVBLK / 4) + 255) / 256;
  int k_key[NK], k_lds[NK], v_k4[NV], v_lds[NV];
  int64_t k_goff[NK], v_goff[NV];
  bool k_on[NK], v_on[NV];
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const int idx = tid + 256 * j;
    const int key = idx / (DP / 8), c8 = idx % (DP / 8);
    k_key[j] = key;
    k_on[j] = idx < KVBLK * (DP / 8) && c8 * 8 < D;
    k_lds[j] = idx < KVBLK * (DP / 8) ? key * KROW + c8 * 8 : -1;
    k_goff[j] = (int64_t)key * a.k_ts + c8 * 8;
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = tid + 256 * j;
    const int d = idx / (KVBLK / 4), k4 = idx % (KVBLK / 4);
    v_k4[j] = k4 * 4;
    v_on[j] = idx < DT * 32 * (KVBLK / 4) && d < D;
    v_lds[j] = idx < DT * 32 * (KVBLK / 4) ? d * VROW + k4 * 4 : -1;
    v_goff[j] = (int64_t)d * a.vt_ds + k4 * 4;
  }

  // The K / V^T rows of tile t + 1 are requested (into registers) before tile t's MFMAs and written
  // to LDS behind them: one LDS buffer, the loads a whole tile of work ahead.  Every load is issued
  // unconditionally (rows outside the segment or the head dimension read the tile's first element
  // and are zeroed on the way to LDS) so that the compiler can count the loads in flight.
  const int tiles_per_seg = (a.seg_len + KVBLK - 1) / KVBLK;
  const int n_tiles = a.S * tiles_per_seg;
  f16x8 kreg[NK];
  f16x4 vreg[NV];
  uint32_t okk = 0, okv = 0;                         // validity bits of the staged groups
  auto request = [&](int tile) {
    const int s = tile / tiles_per_seg, t = tile - s * tiles_per_seg;
    const int kb = a.seg_batch[b * a.S + (s < a.S ? s : a.S - 1)];
    const f16* kbase = a.k + kb * a.k_bs + h * a.k_hs;
    const f16* vbase = a.vt + kb * a.vt_bs + h * a.vt_hs;
    const int key0 = t * KVBLK;
    uint32_t ok = 0, ov = 0;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const bool on = tile < n_tiles && k_on[j] && key0 + k_key[j] < a.seg_len;
      kreg[j] = *reinterpret_cast<const f16x8*>(kbase + (on ? (int64_t)key0 * a.k_ts + k_goff[j] : 0));
      ok |= (on ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool on = tile < n_tiles && v_on[j] && key0 + v_k4[j] < a.seg_len;   // seg_len % 4 == 0
      vreg[j] = *reinterpret_cast<const f16x4*>(vbase + (on ? key0 + v_goff[j] : 0));
      ov |= (on ? 1u : 0u) << j;
    }
    okk = ok;
    okv = ov;
  };
  auto publish = [&]() {
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      if (k_lds[j] < 0) continue;
      f16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (f16)0.0f;
      *reinterpret_cast<f16x8*>(&sK[k_lds[j]]) = (okk >> j) & 1u ? kreg[j] : z;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (v_lds[j] < 0) continue;
      f16x4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) z[e] = (f16)0.0f;
      *reinterpret_cast<f16x4*>(&sV[v_lds[j]]) = (okv >> j) & 1u ? vreg[j] : z;
    }
  };
  static_assert(NK <= 32 && NV <= 32, "one validity bit per staged group");
  request(0);
  {
    for (int tile = 0; tile < n_tiles; ++tile) {
      const int key0 = (tile % tiles_per_seg) * KVBLK;
      __syncthreads();   // previous tile fully consumed
      publish();
      __syncthreads();
      request(tile + 1);

      // ---- S^T = K.Q^T for the two 32-key tiles: st[kt][r] = score(key = 32kt + row(r), qi)
      f32x16 st[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] = 0.0f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
          const f16x8 kf = *reinterpret_cast<const f16x8*>(
              &sK[(kt * 32 + l31) * KROW + 16 * c + 8 * hh]);
          st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[c], st[kt], 0, 0, 0);
        }
      }
      // ---- online softmax over this tile's 64 keys (in log2 domain).  Keys beyond the segment
      // exist only in a segment's last tile; the scale is positive, so the running maximum is
      // taken over the raw scores and scaled once, and each probability is ONE fma + v_exp_f32
      // (arguments are <= 0: the raw hardware exp2 needs no range handling).
      if (key0 + KVBLK > a.seg_len) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= a.seg_len) st[kt][r] = -1e30f;
          }
      }
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      mx = fmaxf(mx * a.scale_log2e, -1e30f);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.0f;
      f16x8 pf[4];   // B operand of V^T.P^T per 16-key group g = 2kt + (r>>3)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], a.scale_log2e, -m_new));
          psum += p;
          pf[2 * kt + (r >> 3)][r & 7] = (f16)p;
        }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int n = 0; n < DT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[n][r] *= alpha;
      // ---- O^T += V^T.P^T : A operand lane (d = 32n + l31) reads keys
      //      {16g + 4hh + 0..3, 16g + 8 + 4hh + 0..3}, matching pf[g]'s k order
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int n = 0; n < DT; ++n) {
          const f16* vr = &sV[(n * 32 + l31) * VROW + 16 * g + 4 * hh];
          const f16x4 lo = *reinterpret_cast<const f16x4*>(vr);
          const f16x4 hi = *reinterpret_cast<const f16x4*>(vr + 8);
          f16x8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
          oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[g], oacc[n], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: O[qi][d] = oacc / l ; lane holds d rows (r&3) + 8(r>>2) + 4hh of tile n
  if (qv) {
    const float inv_l = 1.0f / l_run;
    f16* op = a.o + b * a.o_bs + (int64_t)qi * a.o_ts + h * a.o_hs;
#pragma unroll
    for (int n = 0; n < DT; ++n)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = n * 32 + 8 * r4 + 4 * hh;
        if (d0 + 3 < D) {
          f16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (f16)(oacc[n][4 * r4 + e] * inv_l);
          *reinterpret_cast<f16x4*>(op + d0) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (d0 + e < D) op[d0 + e] = (f16)(oacc[n][4 * r4 + e] * inv_l);
        }
      }
  }
}

}  // namespace

extern "C" {

int dsu_mv_attention_fwd(const void* q, const void* k, const void* vt, void* out,
                         const int32_t* seg_batch, int32_t Bq, int32_t H, int32_t Nq, int32_t d,
                         int32_t S, int32_t seg_len, const int64_t* q_strides,
                         const int64_t* k_strides, const int64_t* vt_strides,
                         const int64_t* o_strides, float scale, void* stream) {
  if (!q || !k || !vt || !out || !seg_batch || !q_strides || !k_strides || !vt_strides ||
      !o_strides)
    return DSU_EINVAL;
  if (Bq <= 0 || H <= 0 || Nq <= 0 || d <= 0 || S <= 0 || seg_len <= 0) return DSU_EINVAL;
  if (seg_len % 4 != 0) return DSU_EUNSUP;   // V^T rows are staged 4 keys (8 bytes) at a time
  // 16-byte Q/K row chunks and 8-byte V^T chunks must be naturally aligned
  if ((q_strides[0] | q_strides[1] | q_strides[2] | k_strides[0] | k_strides[1] | k_strides[2]) % 8)
    return DSU_EUNSUP;
  if ((vt_strides[0] | vt_strides[1] | vt_strides[2] | o_strides[0] | o_strides[1] |
       o_strides[2]) % 4)
    return DSU_EUNSUP;
  if (((uintptr_t)q | (uintptr_t)k) % 16 || ((uintptr_t)vt | (uintptr_t)out) % 8)
    return DSU_EUNSUP;
  AttnArgs a;
  a.q = (const f16*)q; a.k = (const f16*)k; a.vt = (const f16*)vt; a.o = (f16*)out;
  a.seg_batch = seg_batch;
  a.Bq = Bq; a.H = H; a.Nq = Nq; a.S = S; a.seg_len = seg_len;
  a.q_bs = q_strides[0]; a.q_ts = q_strides[1]; a.q_hs = q_strides[2];
  a.k_bs = k_strides[0]; a.k_ts = k_strides[1]; a.k_hs = k_strides[2];
  a.vt_bs = vt_strides[0]; a.vt_hs = vt_strides[1]; a.vt_ds = vt_strides[2];
  a.o_bs = o_strides[0]; a.o_ts = o_strides[1]; a.o_hs = o_strides[2];
  a.scale_log2e = scale * 1.4426950408889634f;
  dim3 grid((Nq + QBLK * WAVES - 1) / (QBLK * WAVES), Bq * H);
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 40: mv_attention_kernel<40><<<grid, 256, 0, s>>>(a); break;
    case 64: mv_attention_kernel<64><<<grid, 256, 0, s>>>(a); break;
    case 80: mv_attention_kernel<80><<<grid, 256, 0, s>>>(a); break;
    case 160: mv_attention_kernel<160><<<grid, 256, 0, s>>>(a); break;
    default: return DSU_EUNSUP;
  }
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
