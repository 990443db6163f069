// Occupancy-grid refresh of the NSR optimisation (every 16th step) as one stream-ordered sequence
// without host round trips.
//
// Replaces the Python of nerfacc 0.3.3 OccupancyGrid._update (grid.py, un-vendored; restated in
// nsr/render.py) with NeuSModel.update_step's occ_eval_fn
// (2_charactor_reconstructor/instant_nsr/models/neus.py:61-83, called from :85-88):
//   cells   = all (warm-up, step < 256) or  N/4 uniform draws  +  the occupied cells
//             (a uniform subsample of N/4 of them, with replacement, when there are more)
//   x       = (cell coords + U[0,1)^3) / res  mapped to the aabb
//   occ     = clip((sigmoid(prev inv_s) - sigmoid(next inv_s) + 1e-5) / (sigmoid(prev inv_s) + 1e-5), 0, 1),
//             prev / next = sdf(x) +- render_step_size / 2
//   occs[c] = max(occs[c] * decay, occ);   binary = occs > min(mean(occs), occ_thre)
// A cell drawn more than once (a uniform draw that hits an occupied cell: about a fifth of them)
// is decayed ONCE and receives the largest of its alphas: nerfacc's indexed assignment gathers
// every old value before it scatters, and which duplicate it keeps is unspecified; the largest is
// one of its outcomes and makes the result independent of the run order.
// The torch form costs ~50 launches and two host synchronisations (nonzero's size, float(mean)) —
// 0.7 ms of the 1.26 ms a refresh step adds (profiles/round4_nsr_refresh_step_timeline.txt).  Here:
// ordered selection of the occupied cells (hipCUB, the size stays on the device), one kernel for
// cells + points (launched for the 2 x N/4 capacity; unused slots carry cell -1 and a dummy point),
// dsu_sdf_fwd, one kernel for alpha (atomic max per cell into a scratch grid preset to -1), the EMA
// of the visited cells riding in the partial-sum pass, mean + threshold + binarisation.
// Draws: Philox4x32-10 keyed (seed, step), streams 8-10 (the step's own draws use 0-2).
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace {

__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Iota {   // counting "iterator" materialised once (hipCUB's select takes arrays here)
  int32_t* p;
};

__global__ void iota_kernel(int32_t* __restrict__ out, int32_t n) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

// slot i < n_uni: a uniform cell; slot n_uni + j: occupied cell j (count <= n_uni) or a uniform
// pick among the occupied cells (count > n_uni); all == 1: slot i is cell i (warm-up)
__global__ __launch_bounds__(256) void occ_points_kernel(
    uint64_t seed, int64_t step, int32_t res, int32_t n_cells, int32_t n_uni, int32_t all,
    const int32_t* __restrict__ occupied, const int32_t* __restrict__ n_occupied,
    const int32_t* __restrict__ cell_inj, const float* __restrict__ rand_inj, float lo0, float lo1, float lo2, float hi0, float hi1,
    float hi2, int32_t m, int32_t* __restrict__ cell, float* __restrict__ pts,
    int32_t* __restrict__ cells_out, float* __restrict__ rand_out) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t s0 = (uint32_t)step, s1 = (uint32_t)((uint64_t)step >> 32);
  int32_t c;
  if (cell_inj) {
    c = cell_inj[i];
  } else if (all) {
    c = i;
  } else if (i < n_uni) {
    c = (int32_t)(philox(make_uint4((uint32_t)i, 8u, s0, s1), key).x % (uint32_t)n_cells);
  } else {
    const int32_t j = i - n_uni, cnt = *n_occupied;
    if (cnt <= n_uni) c = j < cnt ? occupied[j] : -1;
    else c = occupied[philox(make_uint4((uint32_t)j, 9u, s0, s1), key).x % (uint32_t)cnt];
  }
  cell[i] = c;
  float r0, r1, r2;
  if (rand_inj) {
    r0 = rand_inj[3 * (size_t)i]; r1 = rand_inj[3 * (size_t)i + 1]; r2 = rand_inj[3 * (size_t)i + 2];
  } else {
    const uint4 r = philox(make_uint4((uint32_t)i, 10u, s0, s1), key);
    r0 = u01(r.x); r1 = u01(r.y); r2 = u01(r.z);
  }
  if (cells_out) cells_out[i] = c;
  if (rand_out) {
    rand_out[3 * (size_t)i] = r0; rand_out[3 * (size_t)i + 1] = r1; rand_out[3 * (size_t)i + 2] = r2;
  }
  const int32_t cc = c < 0 ? 0 : c;
  const int32_t ix = cc / (res * res), iy = (cc / res) % res, iz = cc % res;
  // (coords + rand) / res * (hi - lo) + lo   (grid.py _update; render.py _cell_points)
  const float fr = (float)res;
  pts[3 * (size_t)i + 0] = ((float)ix + r0) / fr * (hi0 - lo0) + lo0;
  pts[3 * (size_t)i + 1] = ((float)iy + r1) / fr * (hi1 - lo1) + lo1;
  pts[3 * (size_t)i + 2] = ((float)iz + r2) / fr * (hi2 - lo2) + lo2;
}

__global__ __launch_bounds__(256) void occ_alpha_kernel(
    const float* __restrict__ sdf, const int32_t* __restrict__ cell, int32_t m,
    const float* __restrict__ inv_s_p, float half_step, int32_t* __restrict__ amax) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float inv_s = fminf(fmaxf(*inv_s_p, 1e-6f), 1e6f);
  const float s = sdf[i];
  const float next = s - half_step, prev = s + half_step;
  const float pc = sigmoidf_(prev * inv_s), nc = sigmoidf_(next * inv_s);
  const float a = fminf(fmaxf(((pc - nc) + 1e-5f) / (pc + 1e-5f), 0.0f), 1.0f);
  const int32_t c = cell[i];
  // alpha >= 0: its bit pattern orders like the value and beats the preset -1.0f (negative as int)
  if (c >= 0) atomicMax(&amax[c], __float_as_int(a));
}

constexpr int SUM_BLOCKS = 256;

// EMA of the visited cells (amax >= 0) and the partial sums of the updated grid in one pass
__global__ __launch_bounds__(256) void occ_ema_sum_kernel(float* __restrict__ occs,
                                                          const int32_t* __restrict__ amax,
                                                          float decay, int32_t n,
                                                          double* __restrict__ partial) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float v = occs[i];
    const int32_t ab = amax[i];
    if (ab >= 0) {
      v = fmaxf(v * decay, __int_as_float(ab));
      occs[i] = v;
    }
    acc += (double)v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void occ_binarize_kernel(const float* __restrict__ occs, int32_t n,
                                                           const double* __restrict__ partial,
                                                           float occ_thre, uint8_t* __restrict__ bin,
                                                           float* __restrict__ thre_out) {
  __shared__ double red[256];
  red[threadIdx.x] = threadIdx.x < SUM_BLOCKS ? partial[threadIdx.x] : 0.0;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float thre = fminf((float)(red[0] / (double)n), occ_thre);   // clamp(occs.mean(), max=occ_thre)
  if (blockIdx.x == 0 && threadIdx.x == 0 && thre_out) *thre_out = thre;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    bin[i] = occs[i] > thre ? 1 : 0;
}

struct Carve {
  char* base;
  int64_t off = 0;
  template <typename T>
  T* take(int64_t count) {
    off = (off + 255) / 256 * 256;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * (int64_t)sizeof(T);
    return p;
  }
};

struct Ws {
  int32_t *iota, *occupied, *n_occupied, *cell, *amax;
  float *pts, *sdf;
  double* partial;
  void* cub;
  size_t cub_bytes;
  int64_t total;
};

int carve(int32_t res, char* base, Ws& w) {
  const int64_t n = (int64_t)res * res * res;
  if (res < 1 || n >= (1ll << 31)) return DSU_EINVAL;
  // scratch of the ordered selection: per-tile descriptors only (a few KB for 128^3); reserved by a
  // bound so that the size can be planned without a device, checked against the real need at run time
  const size_t b = (size_t)(n / 64 > (1 << 20) ? n / 64 : (1 << 20));
  Carve k{base};
  w.iota = k.take<int32_t>(n);
  w.occupied = k.take<int32_t>(n);
  w.n_occupied = k.take<int32_t>(4);
  w.cell = k.take<int32_t>(n);            // capacity: every cell (warm-up); 2 x n/4 afterwards
  w.pts = k.take<float>(3 * n);
  w.sdf = k.take<float>(n);
  w.amax = k.take<int32_t>(n);            // per cell: bits of the largest alpha drawn, -1.0f = not visited
  w.partial = k.take<double>(SUM_BLOCKS);
  w.cub = k.take<char>((int64_t)b);
  w.cub_bytes = b;
  w.total = (k.off + 255) / 256 * 256;
  return DSU_OK;
}

}  // namespace

extern "C" {

int64_t dsu_occgrid_refresh_workspace_bytes(int32_t res) {
  Ws w;
  if (carve(res, nullptr, w) != DSU_OK) return -1;
  return w.total;
}

int dsu_occgrid_refresh(const dsu_occgrid_refresh_args* a, void* stream) {
  if (!a || !a->occs || !a->binary || !a->grid || !a->table_img || !a->mlp || !a->inv_s ||
      !a->workspace || !a->aabb)
    return DSU_EINVAL;
  if (a->res < 1 || !(a->ema_decay >= 0.0f) || !(a->render_step_size > 0.0f)) return DSU_EINVAL;
  Ws w;
  int rc = carve(a->res, (char*)a->workspace, w);
  if (rc) return rc;
  if (a->workspace_bytes < w.total) return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int32_t n = a->res * a->res * a->res;
  const int32_t n_uni = n / 4;
  int32_t m = a->all_cells ? n : 2 * n_uni;
  if (a->inj_cells) {                     // test hook: the caller's cells instead of the draws
    if (a->inj_count < 0 || a->inj_count > n) return DSU_EINVAL;
    m = a->inj_count;
  }
  if (m == 0) return DSU_OK;
  if (!a->all_cells && !a->inj_cells) {
    // occupied cells in ascending order (= torch.nonzero), their number stays on the device
    size_t cub_need = 0;
    if (hipcub::DeviceSelect::Flagged(nullptr, cub_need, w.iota, a->binary, w.occupied, w.n_occupied,
                                      (int)n, s) != hipSuccess)
      return DSU_ELAUNCH;
    if (cub_need > w.cub_bytes) return DSU_EUNSUP;
    iota_kernel<<<dsu_blocks_for(n, 256), 256, 0, s>>>(w.iota, n);
    cub_need = w.cub_bytes;
    if (hipcub::DeviceSelect::Flagged(w.cub, cub_need, w.iota, a->binary, w.occupied, w.n_occupied,
                                      (int)n, s) != hipSuccess)
      return DSU_ELAUNCH;
  }
  occ_points_kernel<<<dsu_blocks_for(m, 256), 256, 0, s>>>(
      a->seed, a->step, a->res, n, n_uni, a->all_cells ? 1 : 0, w.occupied, w.n_occupied, a->inj_cells,
      a->inj_rand, a->aabb[0], a->aabb[1], a->aabb[2], a->aabb[3], a->aabb[4], a->aabb[5], m, w.cell,
      w.pts, a->cells_out, a->rand_out);
  DSU_CHECK_LAUNCH();
  rc = dsu_sdf_fwd(a->grid, a->table_img, a->mlp, w.pts, m, a->radius, a->active_levels, 1, w.sdf, s);
  if (rc) return rc;
  const float half_step = (float)((double)a->render_step_size * 0.5);
  if (hipMemsetD32Async((hipDeviceptr_t)w.amax, (int)0xBF800000u, (size_t)n, s) != hipSuccess)   // -1.0f
    return DSU_ELAUNCH;
  occ_alpha_kernel<<<dsu_blocks_for(m, 256), 256, 0, s>>>(w.sdf, w.cell, m, a->inv_s, half_step,
                                                         w.amax);
  occ_ema_sum_kernel<<<SUM_BLOCKS, 256, 0, s>>>(a->occs, w.amax, a->ema_decay, n, w.partial);
  occ_binarize_kernel<<<dsu_blocks_for(n, 256), 256, 0, s>>>(a->occs, n, w.partial, a->occ_thre,
                                                            a->binary, a->thre_out);
  DSU_CHECK_LAUNCH();
  return DSU_OK;
}

}  // extern "C"
