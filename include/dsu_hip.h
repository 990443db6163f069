/*
 * dsu_hip.h — C ABI of libdsu_hip.so, the MI355X (gfx950) hot path of DrawingSpinUp.
 *
 * Every entry point replaces one third-party native op that the reference reaches
 * through a Python import (the reference ships no native code of its own).  The
 * reference call site each function stands in for is cited as
 * `<path under /root/reference>:<line>`.
 *
 * Conventions
 *   - plain C, no torch types: device pointers, element counts, a hipStream_t passed
 *     as void*.  The caller owns every buffer; nothing is allocated inside.
 *   - all kernels are stream-ordered on `stream`; no internal synchronisation.
 *   - return value: 0 = ok, <0 = DSU_E* error code (dsu_strerror gives the text).
 *   - "f16" buffers hold IEEE binary16, "f32" IEEE binary32, row-major, contiguous.
 */
#ifndef DSU_HIP_H
#define DSU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSU_OK 0
#define DSU_EINVAL (-1)   /* bad argument (null pointer, unsupported size) */
#define DSU_ELAUNCH (-2)  /* hipLaunch / runtime error */
#define DSU_EUNSUP (-3)   /* configuration not supported by the kernels */

#define DSU_MAX_LEVELS 16

const char* dsu_strerror(int code);
/* 1 when the library was built with the A/B environment switches (variant builds for tools/), 0
 * for the product library: it reads no DSU_* environment variable. */
int dsu_ab_switches(void);
/* ABI version of this header; bumped on any signature change. */
int dsu_abi_version(void);
/* Workgroups of the two one-wave-per-SIMD kernels of the NSR step (the MLP part of the geometry
 * backward, the texture backward: 450-510 registers per lane, nothing else fits beside them on a
 * SIMD).  0 / 256 = one workgroup per CU (a single optimisation alone on the GPU); with several
 * drawings in flight on one GPU fewer workgroups leave CUs to the other drawings' kernels while such
 * a kernel runs (bench.py --inflight: 192).  Process-wide; workspaces stay sized for 256. */
int dsu_set_onewave_grid_cap(int32_t workgroups);
/* The same for the table-gradient scatter of the geometry backward (0 = its resident count: three
 * 256-thread workgroups of 45 KB LDS per CU). */
int dsu_set_scatter_grid_cap(int32_t workgroups);
/* Priority of the side stream on which the NSR step driver marches and packs the next step's samples
 * (dsu_nsr_driver_create reads it): 1 = high (default: what several drawings in flight measured best
 * with), 2 = normal (one drawing at a time: streams of non-default priority put every fourth and later
 * drawing of a process into a 0.8 s slower mode, profiles/round6_side_stream_priority.txt), 0 = low.
 * Process-wide. */
int dsu_set_nsr_side_stream_priority(int32_t level);
/* 1: a destroyed step driver hands its (drained) side stream to the next driver created on the same
 * device with the same priority, so that a process creates as many side streams as it has drawings
 * in flight instead of one per drawing.  0 (default): one stream per driver.  Prepared at the end of
 * round 6: six of six back-to-back reconstructions fast at high priority with it (un-pooled: slow from
 * the fourth on); the bench line with drawings in flight is not measured with it yet.  Process-wide. */
int dsu_set_nsr_side_stream_pooling(int32_t on);

/* ------------------------------------------------------------------------------------
 * Multi-resolution hash grid (replaces tiny-cuda-nn `tcnn.Encoding(3, {otype:HashGrid})`
 * constructed at 2_charactor_reconstructor/instant_nsr/models/network_utils.py:46 and
 * called at :55; hyper-parameters from configs/neuralangelo-ortho-wmask.yaml:52-62).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  uint32_t n_levels;           /* 10 */
  uint32_t n_features;         /* 2 (the kernels are specialised for 2) */
  uint32_t log2_hashmap_size;  /* 19 */
  uint32_t base_resolution;    /* 32 */
  double per_level_scale;      /* 1.3195079107728942 */
} dsu_hashgrid_cfg;

/* Per-level derived quantities.  offsets[l] = first table ENTRY of level l (an entry is
 * n_features values); offsets[n_levels] = total entries.  scale/resolution follow
 * tcnn's grid_scale()/grid_resolution(); hashed[l]=1 when the level is addressed by the
 * coherent-prime hash instead of the dense x+y*res+z*res^2 index. */
typedef struct {
  uint32_t offsets[DSU_MAX_LEVELS + 1];
  uint32_t resolution[DSU_MAX_LEVELS];
  float scale[DSU_MAX_LEVELS];
  uint32_t hashed[DSU_MAX_LEVELS];
} dsu_hashgrid_levels;

/* Host-only: fill `out` for `cfg`.  Pure function; safe without a GPU. */
int dsu_hashgrid_make_levels(const dsu_hashgrid_cfg* cfg, dsu_hashgrid_levels* out);

/* tcnn.Encoding.__call__ (network_utils.py:55) fused with the progressive level mask
 * (network_utils.py:56): out[n, 2*l..2*l+1] = trilinear lookup for l < active_levels,
 * 0 for the masked levels.  x: (n,3) f32 in [0,1]; table: (total_entries,2) f16;
 * out: (n, 2*n_levels) f16. */
int dsu_hashgrid_encode_fwd(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                            const float* x, int64_t n, uint32_t active_levels,
                            void* out_f16, void* stream);

/* Backward of the above w.r.t. the table only (the reference never needs dL/dx:
 * grad_type=finite_difference, neuralangelo-ortho-wmask.yaml:42).  dout: (n, 2*n_levels)
 * f32; grad_table: (total_entries,2) f32, ACCUMULATED into (caller zeroes it). */
int dsu_hashgrid_encode_bwd(const dsu_hashgrid_cfg* cfg, const float* x, const float* dout,
                            int64_t n, uint32_t active_levels, float* grad_table,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * Fused SDF network: contract -> hash grid -> cat(xyz*2-1, enc*mask) -> Linear(23,64) ->
 * Softplus(beta=100) -> Linear(64,13).  Replaces VolumeSDF.forward_level
 * (instant_nsr/models/geometry.py:189-194) and the per-point body of VolumeSDF.forward
 * (geometry.py:135-187): N1-N5 of SURVEY.md §8(a).
 * Weights are the EFFECTIVE (weight-norm already applied) row-major nn.Linear matrices.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* w0; /* (64, 23) */
  const float* b0; /* (64) */
  const float* w1; /* (13, 64) */
  const float* b1; /* (13) */
} dsu_sdf_mlp;

/* pts: (n,3) f32 world coordinates in [-radius, radius]; out: (n, n_out) f32 where
 * n_out = 1 (sdf only: forward_level / occupancy / export) or 13 (sdf + feature). */
int dsu_sdf_fwd(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                const float* pts, int64_t n, float radius, uint32_t active_levels,
                uint32_t n_out, float* out, void* stream);
/* The same network on the export's lattice (BaseImplicitGeometry.isosurface_, geometry.py:83-106:
 * `scale_anything(meshgrid(linspace(0,1,res)), (0,1), (vmin,vmax))` -> forward_level, chunk by
 * chunk): the points are formed in the kernel, x-slabs [x0, x0 + nx) of the res^3 lattice,
 * p = lin[i] * span + lo per axis (two rounded f32 operations, as the reference's tensor
 * expression).  lin: (res) f32 on the device (the caller's linspace); out: (nx * res * res) f32,
 * x-major / y / z-minor. */
int dsu_sdf_fwd_lattice(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                        const float* lin, int32_t res, int32_t x0, int32_t nx, const float* lo3,
                        const float* span3, float radius, uint32_t active_levels, float* out,
                        void* stream);

/* VolumeSDF.forward(points, with_grad, with_feature, with_laplace) with
 * grad_type=finite_difference (geometry.py:158-176): 7 network evaluations per point
 * (centre + 6 clamped +-eps offsets).  Outputs (any may be NULL except sdf):
 *   sdf (n), grad (n,3), feature (n,13), laplace (n). */
int dsu_sdf_fd_fwd(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                   const float* pts, int64_t n, float radius, float eps,
                   uint32_t active_levels, float* sdf, float* grad, float* feature,
                   float* laplace, void* stream);

/* Backward of dsu_sdf_fd_fwd w.r.t. table and MLP parameters.  Upstream gradients
 * (NULL = zero): d_sdf (n), d_grad (n,3), d_feature (n,13), d_laplace (n).
 * Accumulates into grad_table (entries,2) f32 and g_w0 (64,23), g_b0 (64), g_w1 (13,64),
 * g_b1 (13) f32 — caller zeroes them.  `workspace` is caller-owned device scratch of at
 * least dsu_sdf_fd_bwd_workspace_bytes(cfg, n) bytes (contents undefined on return). */
int dsu_sdf_fd_bwd(const dsu_hashgrid_cfg* cfg, const void* table_f16, const dsu_sdf_mlp* mlp,
                   const float* pts, int64_t n, float radius, float eps,
                   uint32_t active_levels, const float* d_sdf, const float* d_grad,
                   const float* d_feature, const float* d_laplace, float* grad_table,
                   float* g_w0, float* g_b0, float* g_w1, float* g_b1, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* Same two calls with a feature cache between them: the forward pass stores the interpolated
 * f16 hash-grid features of all 7 evaluations ([eval][point][active level] half2,
 * dsu_sdf_fd_enc_cache_bytes(n, active_levels) bytes) and the backward pass reads them back
 * instead of repeating the 7 x active_levels x 8 table gathers per point.  Same results bit
 * for bit (the cached values ARE the forward's features); enc_cache NULL = the plain calls. */
int dsu_sdf_fd_fwd_cached(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                          const dsu_sdf_mlp* mlp, const float* pts, int64_t n, float radius,
                          float eps, uint32_t active_levels, float* sdf, float* grad,
                          float* feature, float* laplace, void* enc_cache, void* stream);
int dsu_sdf_fd_bwd_cached(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                          const dsu_sdf_mlp* mlp, const float* pts, int64_t n, float radius,
                          float eps, uint32_t active_levels, const float* d_sdf,
                          const float* d_grad, const float* d_feature, const float* d_laplace,
                          float* grad_table, float* g_w0, float* g_b0, float* g_w1, float* g_b1,
                          void* workspace, int64_t workspace_bytes, const void* enc_cache,
                          void* stream);
int64_t dsu_sdf_fd_enc_cache_bytes(int64_t n, uint32_t active_levels);

/* Spatially sorted evaluation order for one optimisation step.  VolumeSDF.forward is pointwise
 * (geometry.py:135-187), but its callers hand it the samples ray by ray (neus.py:119-129 ->
 * nerfacc.ray_marching order) and the rays are random pixels of six views, so neighbouring
 * lanes touch unrelated table neighbourhoods.  dsu_spatial_sort bins the points on a
 * 2^bits-per-axis lattice of the [-radius, radius]^3 cube in Morton order (counting sort;
 * bits 4..7): pts_sorted[i] = pts[perm[i]].  The *_sorted calls take the sorted points plus
 * perm and read / write every per-point array (sdf, grad, feature, laplace, d_*) in the
 * CALLER'S row order through it; the feature cache is kept in sorted order.  perm NULL = the
 * plain calls.  Same per-point arithmetic, bit for bit. */
int64_t dsu_spatial_sort_workspace_bytes(int64_t n, int32_t bits);
int dsu_spatial_sort(const float* pts, int64_t n, float radius, int32_t bits, int32_t* perm,
                     float* pts_sorted, void* workspace, int64_t workspace_bytes, void* stream);
/* Prefetch-path variants: the sample total of the next step is still on the device when the side
 * stream packs and sorts its points.  dsu_ray_compact_points_cap = dsu_ray_compact_points into
 * fixed-capacity outputs (rows beyond out_capacity are dropped: the caller checks the total
 * later); dsu_points_tail writes the 2 x n_random random / perturbed points of neus.py:155-162
 * behind the ray samples (row total_dev[0]); dsu_spatial_sort_dev sorts n_dev[0] + n_add points
 * (at most n_capacity). */
int dsu_points_tail(float* points, int64_t capacity_rows, const int32_t* total_dev,
                    const float* pts_random, const float* perturb, int64_t n_random, float alpha,
                    void* stream);
int dsu_spatial_sort_dev(const float* pts, int64_t n_capacity, const int32_t* n_dev, int64_t n_add,
                         float radius, int32_t bits, int32_t* perm, float* pts_sorted,
                         void* workspace, int64_t workspace_bytes, void* stream);
int dsu_sdf_fd_fwd_sorted(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                          const dsu_sdf_mlp* mlp, const float* pts_sorted, const int32_t* perm,
                          int64_t n, float radius, float eps, uint32_t active_levels, float* sdf,
                          float* grad, float* feature, float* laplace, void* enc_cache,
                          void* stream);
int dsu_sdf_fd_bwd_sorted(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                          const dsu_sdf_mlp* mlp, const float* pts_sorted, const int32_t* perm,
                          int64_t n, float radius, float eps, uint32_t active_levels,
                          const float* d_sdf, const float* d_grad, const float* d_feature,
                          const float* d_laplace, float* grad_table, float* g_w0, float* g_b0,
                          float* g_w1, float* g_b1, void* workspace, int64_t workspace_bytes,
                          const void* enc_cache, void* stream);

/* A deferred sum of per-workgroup partial vectors: element v of the result is
 * sum_b partials[b * stride + v] (b < nblocks, v < n), added to base[map[v]] (map[v] < 0: a padding
 * element, dropped).  The partial reductions of the backward kernels are carried out by extra
 * workgroups of the scatter launch of dsu_sdf_fd_bwd_sorted_fold instead of launches of their own. */
typedef struct dsu_partial_reduce {
  const float* partials;
  const int32_t* map;   /* device, n entries */
  float* base;
  int32_t nblocks, stride, n;
} dsu_partial_reduce;

/* dsu_sdf_fd_bwd_sorted with a hipEvent_t (as void*, may be NULL) recorded on `stream` between the
 * two kernels of the backward: behind the MLP part, which occupies every SIMD with one
 * 458-register wave, and in front of the table-gradient scatter, which leaves room.  Work a caller
 * wants to overlap with the backward (the next step's ray march) waits for it. */
int dsu_sdf_fd_bwd_sorted_mid(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                              const dsu_sdf_mlp* mlp, const float* pts_sorted, const int32_t* perm,
                              int64_t n, float radius, float eps, uint32_t active_levels,
                              const float* d_sdf, const float* d_grad, const float* d_feature,
                              const float* d_laplace, float* grad_table, float* g_w0, float* g_b0,
                              float* g_w1, float* g_b1, void* workspace, int64_t workspace_bytes,
                              const void* enc_cache, void* mid_event, void* stream);

/* dsu_sdf_fd_bwd_sorted_mid whose scatter launch also carries out `extra` (may be NULL), a deferred
 * partial sum of ANOTHER kernel that precedes this call on `stream` (the native NSR step passes the
 * texture backward's: dsu_texture_bwd_shaded_partials).  The backward's own partial sums always
 * ride in that launch.  DSU_EUNSUP with `extra` on the fused / VALU variants. */
int dsu_sdf_fd_bwd_sorted_fold(const dsu_hashgrid_cfg* cfg, const void* table_f16,
                               const dsu_sdf_mlp* mlp, const float* pts_sorted, const int32_t* perm,
                               int64_t n, float radius, float eps, uint32_t active_levels,
                               const float* d_sdf, const float* d_grad, const float* d_feature,
                               const float* d_laplace, float* grad_table, float* g_w0, float* g_b0,
                               float* g_w1, float* g_b1, void* workspace, int64_t workspace_bytes,
                               const void* enc_cache, void* mid_event,
                               const dsu_partial_reduce* extra, void* stream);

/* Bytes of device scratch dsu_sdf_fd_bwd needs for n points (per-workgroup partial MLP
 * gradients, summed by a second kernel: deterministic, no same-address atomics).  <0 = error. */
int64_t dsu_sdf_fd_bwd_workspace_bytes(const dsu_hashgrid_cfg* cfg, int64_t n);

/* ------------------------------------------------------------------------------------
 * nerfacc 0.3.3 replacements (call sites instant_nsr/models/neus.py:53-57,84,119-129,
 * 147-152).
 * ---------------------------------------------------------------------------------- */

/* ray_aabb_intersect + optional stratified jitter: t_min += jitter*step (jitter may be
 * NULL).  rays_o/rays_d (n,3) f32; aabb 6 floats (host values). */
int dsu_ray_aabb(const float* rays_o, const float* rays_d, int64_t n_rays, const float* aabb6,
                 const float* jitter, float step, float* t_min, float* t_max, void* stream);

/* Pass 1 of ray_marching: count samples per ray through the binary occupancy grid
 * (res^3 uint8, index x*res*res + y*res + z).  num_steps: (n_rays) int32. */
int dsu_ray_march_count(const float* rays_o, const float* rays_d, const float* t_min,
                        const float* t_max, int64_t n_rays, const float* aabb6,
                        const uint8_t* occ_binary, int32_t res, float step,
                        int32_t* num_steps, void* stream);

/* Pass 2: fill.  offsets: (n_rays) int32 exclusive prefix sum of num_steps.
 * ray_indices: (n_samples) int64; t_starts/t_ends: (n_samples) f32. */
int dsu_ray_march_fill(const float* rays_o, const float* rays_d, const float* t_min,
                       const float* t_max, int64_t n_rays, const float* aabb6,
                       const uint8_t* occ_binary, int32_t res, float step,
                       const int32_t* offsets, int64_t* ray_indices, float* t_starts,
                       float* t_ends, void* stream);

/* Single-pass variant of the two calls above: the serial march runs ONCE, writing each ray's
 * samples into a fixed-capacity scratch row (ray r at scratch[r*capacity ...]) together with
 * its count; dsu_ray_compact then packs the rows at the exclusive prefix sum of the counts.
 * Identical results (same stepping loop); a ray with more than `capacity` samples is
 * reported through num_steps (caller checks max(num_steps) <= capacity). */
int dsu_ray_march_scratch(const float* rays_o, const float* rays_d, const float* t_min,
                          const float* t_max, int64_t n_rays, const float* aabb6,
                          const uint8_t* occ_binary, int32_t res, float step, int32_t capacity,
                          int32_t* num_steps, float* scratch_t_starts, float* scratch_t_ends,
                          void* stream);
int dsu_ray_compact(const float* scratch_t_starts, const float* scratch_t_ends, int32_t capacity,
                    const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                    int64_t* ray_indices, float* t_starts, float* t_ends, void* stream);

/* dsu_ray_compact that also writes the sample positions
 * rays_o[r] + rays_d[r] * ((t_start + t_end) / 2)  (neus.py:131-134: t_origins, t_dirs,
 * midpoints, positions) and no ray_indices — the fused step addresses samples by (offsets,
 * counts) only. */
int dsu_ray_compact_points(const float* scratch_t_starts, const float* scratch_t_ends,
                           int32_t capacity, const int32_t* offsets, const int32_t* counts,
                           int64_t n_rays, const float* rays_o, const float* rays_d,
                           float* t_starts, float* t_ends, float* positions, void* stream);
int dsu_ray_compact_points_cap(const float* scratch_t_starts, const float* scratch_t_ends,
                           int32_t capacity, const int32_t* offsets, const int32_t* counts,
                           int64_t n_rays, const float* rays_o, const float* rays_d,
                           float* t_starts, float* t_ends, float* positions, int64_t out_capacity,
                               void* stream);
/* offsets = exclusive prefix sum of counts; stats[0] = total, stats[1] = max(counts)
 * (nerfacc's packed_info construction; one launch, read back with ONE host copy). */
int dsu_ray_offsets(const int32_t* counts, int64_t n_rays, int32_t* offsets, int32_t* stats,
                    void* stream);

/* render_weight_from_alpha (neus.py:147): per ray segment [offsets[r], offsets[r]+cnt[r])
 * w_i = alpha_i * prod_{j<i}(1-alpha_j).  One ray per thread. */
int dsu_weights_from_alpha_fwd(const float* alpha, const int32_t* offsets, const int32_t* counts,
                               int64_t n_rays, float* weights, void* stream);
int dsu_weights_from_alpha_bwd(const float* alpha, const float* weights, const float* d_weights,
                               const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                               float* d_alpha, void* stream);

/* accumulate_along_rays (neus.py:148-152): out[r, c] = sum_i w_i * v[i, c]  (v NULL -> 1). */
int dsu_accumulate_fwd(const float* weights, const float* values, int32_t channels,
                       const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                       float* out, void* stream);

/* OccupancyGrid._update tail (nerfacc 0.3.3 grid.py): occs[idx] = max(occs[idx]*decay, occ).
 * idx may be NULL (= all cells, warm-up; scratch unused).  With idx, every old value is read
 * before any is written (the indexed assignment gathers first) and a cell listed several times
 * keeps the largest of its candidates: scratch = n floats of the caller.  occ are opacities (>= 0):
 * a negative candidate is stored as 0. */
int dsu_occgrid_ema(float* occs, const int64_t* idx, const float* occ, int64_t n, float decay,
                    float* scratch, void* stream);
/* binary = occs > thre  (thre = min(mean(occs), occ_thre) computed by the caller). */
int dsu_occgrid_binarize(const float* occs, int64_t n_cells, float thre, uint8_t* binary,
                         void* stream);

/* The whole refresh (nerfacc 0.3.3 OccupancyGrid._update with NeuSModel's occ_eval_fn,
 * instant_nsr/models/neus.py:61-88) as one stream-ordered sequence without host round trips:
 * cells (all of them while all_cells = 1, i.e. step < warm-up; otherwise res^3/4 uniform draws + the
 * occupied cells, subsampled to res^3/4 with replacement when there are more) -> points
 * (cell + U[0,1)^3) / res in the aabb -> dsu_sdf_fwd -> occ = clip((sigmoid(prev inv_s) -
 * sigmoid(next inv_s) + 1e-5) / (sigmoid(prev inv_s) + 1e-5)), prev / next = sdf +- render_step_size/2
 * -> occs[c] = max(occs[c] ema_decay, occ) -> binary = occs > min(mean(occs), occ_thre).
 * A cell drawn several times is decayed once and keeps the largest of its occ values (one of the
 * outcomes of nerfacc's indexed assignment, whose winner among duplicates is unspecified; the old
 * values are all gathered before any is written, as there).
 * Draws: Philox4x32-10 keyed (seed, step), streams 8-10.  inj_cells / inj_rand (device, inj_count
 * cells and inj_count x 3 uniforms; tests) replace the draws.  thre_out (device, may be NULL)
 * receives the threshold.  cells_out / rand_out (device, may be NULL; capacity res^3 cells while
 * all_cells, else 2 (res^3 / 4), x 3 uniforms): the cells (-1 = unused slot) and uniforms the call
 * evaluated, so that a caller can replay the update.  workspace:
 * dsu_occgrid_refresh_workspace_bytes(res) (< 0: invalid). */
typedef struct dsu_occgrid_refresh_args {
  float* occs;               /* (res^3) f32, updated in place */
  uint8_t* binary;           /* (res^3) u8: read (occupied cells), then rewritten */
  int32_t res, all_cells;
  const float* aabb;         /* HOST: min xyz, max xyz */
  uint64_t seed;
  int64_t step;
  const dsu_hashgrid_cfg* grid;
  const void* table_img;
  const dsu_sdf_mlp* mlp;    /* effective (weight-normed) weights */
  const float* inv_s;        /* device scalar exp(10 variance) */
  float radius, render_step_size, ema_decay, occ_thre;
  uint32_t active_levels;
  int32_t inj_count;
  const int32_t* inj_cells;
  const float* inj_rand;
  float* thre_out;
  void* workspace;
  int64_t workspace_bytes;
  int32_t* cells_out;
  float* rand_out;
} dsu_occgrid_refresh_args;
int64_t dsu_occgrid_refresh_workspace_bytes(int32_t res);
int dsu_occgrid_refresh(const dsu_occgrid_refresh_args* args, void* stream);

/* Fused NeuS shading + compositing of one ray batch (neus.py:90-112 get_alpha, :143-153
 * render_weight_from_alpha + the four accumulate_along_rays): per sample
 *   alpha = clip((sigmoid(prev*inv_s) - sigmoid(next*inv_s) + 1e-5) / (sigmoid(prev*inv_s) + 1e-5))
 * with the cos-anneal of neus.py:97-104, w = alpha * prod_{j<i}(1-alpha_j); per ray
 * comp[r] = {opacity, depth, rgb(3), sum w*normal (3)}.  sdf (n), normal (n,3), rgb (n,3),
 * rays_d (n_rays,3) unit directions, t_starts/t_ends (n); alpha/weights (n) are written for
 * the backward pass.  inv_s: DEVICE pointer to the scalar exp(10*variance) (clipped to
 * [1e-6,1e6] inside, neus.py:91), so no host synchronisation is needed. */
int dsu_neus_composite_fwd(const float* sdf, const float* normal, const float* rgb,
                           const float* rays_d, const float* t_starts, const float* t_ends,
                           const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                           const float* inv_s, float cos_anneal_ratio, float* alpha,
                           float* weights, float* comp, void* stream);
/* Backward: d_comp (n_rays,8), optional d_weights (n) -> d_sdf (n), d_normal (n,3), d_rgb (n,3);
 * d_inv_s (1 float, caller zeroes) accumulates the gradient of the variance scalar. */
int dsu_neus_composite_bwd(const float* sdf, const float* normal, const float* rgb,
                           const float* rays_d, const float* t_starts, const float* t_ends,
                           const int32_t* offsets, const int32_t* counts, int64_t n_rays,
                           const float* inv_s, float cos_anneal_ratio, const float* alpha,
                           const float* weights, const float* d_comp, const float* d_weights,
                           float* d_sdf, float* d_normal, float* d_rgb, float* d_inv_s,
                           void* stream);
/* normal = F.normalize(sdf_grad) and the texture-MLP input cat(feature, normal) (neus.py:143,
 * texture.py:22) in one pass; and its backward. */
int dsu_shade_prep_fwd(const float* grad, const float* feature, int64_t n, float* normal,
                       float* tex_in, void* stream);
int dsu_shade_prep_bwd(const float* grad, const float* d_normal, const float* d_tex_in, int64_t n,
                       float* d_grad, float* d_feature, void* stream);
/* The same, and rows [n, n + tail_rows) of d_feature set to zero: the regulariser points that
 * share the geometry launch with the n ray samples (neus.py:155-162) receive no feature gradient. */
int dsu_shade_prep_bwd_tail(const float* grad, const float* d_normal, const float* d_tex_in,
                            int64_t n, int64_t tail_rows, float* d_grad, float* d_feature,
                            void* stream);

/* VolumeRadiance (2_charactor_reconstructor/instant_nsr/models/texture.py:9-30): VanillaMLP
 * 16 -> 64 -> 64 -> 3, ReLU, no weight norm (models/network_utils.py:94-138), then sigmoid.
 * Weights row-major (out, in) f32 as nn.Linear stores them (texture.network.layers.{0,2,4}).
 *   dsu_texture_fwd: rgb (n,3) = sigmoid(MLP(tex_in (n,16)))
 *   dsu_texture_bwd: d_tex_in (n,16) written; g_* accumulated (+=) with the parameter
 *     gradients; `rgb` is the forward output (sigmoid' = rgb (1 - rgb)); hidden activations are
 *     recomputed.  workspace: dsu_texture_bwd_workspace_bytes(n) bytes of device memory. */
typedef struct dsu_tex_mlp {
  const float *w0, *b0, *w1, *b1, *w2, *b2;
} dsu_tex_mlp;
int dsu_texture_fwd(const dsu_tex_mlp* mlp, const float* tex_in, int64_t n, float* rgb,
                    void* stream);
int64_t dsu_texture_bwd_workspace_bytes(int64_t n);
int dsu_texture_bwd(const dsu_tex_mlp* mlp, const float* tex_in, const float* rgb,
                    const float* d_rgb, int64_t n, float* d_tex_in, float* g_w0, float* g_b0,
                    float* g_w1, float* g_b1, float* g_w2, float* g_b2, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* The same two calls with the shading glue of neus.py:143 / texture.py:22 inside: the input row is
 * cat(feature (n,13), normalize(sdf_grad (n,3))) built in the kernel (`normal` (n,3) is written for
 * the compositing kernels), and the backward pulls the row's gradient back through cat / normalize
 * itself: d_feature (n + tail_rows, 13) (the tail rows zeroed), d_grad (n,3) =
 * (dn - n (n.dn)) / |sdf_grad| with dn = d_row[13:16] + d_normal (d_normal may be NULL).  Replaces
 * dsu_shade_prep_fwd + dsu_texture_fwd and dsu_texture_bwd + dsu_shade_prep_bwd. */
int dsu_texture_fwd_shaded(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                           int64_t n, float* normal, float* rgb, void* stream);
int dsu_texture_bwd_shaded(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                           const float* rgb, const float* d_rgb, const float* d_normal, int64_t n,
                           int64_t tail_rows, float* d_grad, float* d_feature, float* g_w0,
                           float* g_b0, float* g_w1, float* g_b1, float* g_w2, float* g_b2,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* dsu_texture_bwd_shaded without its final launch: the per-workgroup partial parameter gradients
 * stay in `workspace` and `red` receives partials / nblocks / stride / n of the deferred sum; the
 * caller sets red->map (device copy of dsu_texture_partial_map) and red->base (a contiguous
 * [w0 | b0 | w1 | b1 | w2 | b2] gradient block, accumulated +=) and hands the record to a launch that
 * carries it out (dsu_sdf_fd_bwd_sorted_fold).  dsu_texture_partial_map fills map_host (may be NULL)
 * and returns the number of entries. */
int32_t dsu_texture_partial_map(int32_t* map_host);
int dsu_texture_bwd_shaded_partials(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                                    const float* rgb, const float* d_rgb, const float* d_normal,
                                    int64_t n, int64_t tail_rows, float* d_grad, float* d_feature,
                                    void* workspace, int64_t workspace_bytes,
                                    dsu_partial_reduce* red, void* stream);
/* The pair with the ReLU pattern of hidden layer 1 handed from the forward to the backward
 * (h1_mask: (n, 2) uint32, word h of a sample = its hidden units 32 T + (r & 3) + 8 (r >> 2) + 4 h,
 * bit 16 T + r): the backward then recomputes layer 1 with three bf16 products per f32 product
 * (2^-16 relative on the activations) while the masks stay the forward's, bit for bit.  NULL mask =
 * the calls above (exact f32 recompute). */
int dsu_texture_fwd_shaded_m(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                             int64_t n, float* normal, float* rgb, uint32_t* h1_mask, void* stream);
int dsu_texture_bwd_shaded_partials_m(const dsu_tex_mlp* mlp, const float* feature, const float* grad,
                                      const float* rgb, const float* d_rgb, const float* d_normal,
                                      int64_t n, int64_t tail_rows, float* d_grad, float* d_feature,
                                      const uint32_t* h1_mask, void* workspace, int64_t workspace_bytes,
                                      dsu_partial_reduce* red, void* stream);

/* OrthoNeuSSystem.preprocess_data (systems/neus_ortho.py:26-82) for n sampled (view, y, x)
 * triples (int64, drawn by the caller): c2w gather, get_ortho_rays (models/ray_utils.py:36-58),
 * colour / normal / mask / view-weight gathers, cosines = cosine_similarity(rays_d, normal,
 * eps=1e-6), rays = [rays_o | normalize(rays_d)].  Dataset tensors are the resident
 * (V,H,W,*) f32 arrays of OrthoDatasetBase.setup (datasets/ortho.py:99-151). */
int dsu_ortho_ray_batch(const int64_t* index, const int64_t* x, const int64_t* y, int64_t n,
                        const float* c2w, const float* origins, const float* directions,
                        const float* images, int32_t image_channels, const float* normals,
                        const float* masks, const float* view_weights, int32_t H, int32_t W,
                        float* rays, float* rgb, float* normal, float* mask, float* cosines,
                        float* vw, void* stream);

/* The same launch also writing rays[:, :3] and rays[:, 3:] as two contiguous (n,3) arrays (both or
 * neither may be NULL). */
int dsu_ortho_ray_batch_split(const int64_t* index, const int64_t* x, const int64_t* y, int64_t n,
                              const float* c2w, const float* origins, const float* directions,
                              const float* images, int32_t image_channels, const float* normals,
                              const float* masks, const float* view_weights, int32_t H, int32_t W,
                              float* rays, float* rgb, float* normal, float* mask, float* cosines,
                              float* vw, float* rays_o, float* rays_d, void* stream);

/* Ray-level loss terms of OrthoNeuSSystem.training_step
 * (2_charactor_reconstructor/instant_nsr/systems/neus_ortho.py:94-133 with
 * systems/criterions.py:4-27) and their gradient w.r.t. the raw composite
 * comp (R,8) = [opacity, depth, rgb(3), sum w*normal(3)], in ONE launch:
 *   terms[0] rgb_mse  = ranking(sum_c (rgb-gt)^2 | fg, rgb_p_ratio, mean) * lambda_rgb_mse
 *   terms[1] rgb_l1   = ranking(sum_c |rgb-gt|   | fg, rgb_p_ratio, mean) * lambda_rgb_l1
 *   terms[2] normal   = ranking(1-cos(normalize(n), gt) [* exp|cos'| / sum exp|cos'| if
 *                       geo_aware] | fg, normal_p_ratio, view_weights, sum|mean) * lambda_normal
 *   terms[3] mask     = ranking(BCE(clamp(opacity,1e-3,1-1e-3), mask) | all, mask_p_ratio,
 *                       view_weights, mean) * lambda_mask
 * fg = mask > 0 and cosines < -0.1.  ranking() reproduces criterions.py:16-27 exactly,
 * including its selection rule (sorted errors indexed with the original positions of the k
 * smallest; k = int(ratio * count)).  n_rays <= DSU_RAY_LOSS_MAX_RAYS. */
#define DSU_RAY_LOSS_MAX_RAYS 8192
typedef struct dsu_ray_loss_cfg {
  double rgb_p_ratio, normal_p_ratio, mask_p_ratio;
  float lambda_rgb_mse, lambda_rgb_l1, lambda_normal, lambda_mask;
  int32_t geo_aware;
  int32_t reserved;
} dsu_ray_loss_cfg;
int dsu_ray_losses(const float* comp, const float* rgb, const float* normal, const float* mask,
                   const float* cosines, const float* view_weights, int32_t n_rays,
                   const dsu_ray_loss_cfg* cfg, float* terms, float* d_comp, void* stream);

/* Sample-level terms (neus_ortho.py:118-151) over the concatenated evaluation
 * [n_samples ray samples | n_random random points | n_random perturbed copies]:
 *   terms[0] eikonal       = mean_i (|grad_i| - 1)^2            * lambda_eikonal   (samples)
 *   terms[1] sparsity      = mean_r exp(-scale |sdf_r|)         * lambda_sparsity  (random)
 *   terms[2] normal_smooth = mean_{r,c} |grad_r - grad_perturbed_r| * lambda_smooth
 * and their gradients: d_grad_all[:n_samples] += eikonal part when accumulate_prefix != 0
 * (the compositing/shading backward has already written it; d_sdf_all[:n_samples] is left
 * untouched) or = when 0 (then d_sdf_all[:n_samples] = 0); the random / perturbed rows of
 * d_sdf_all and d_grad_all are written.  accumulate_prefix bit 1 (value 2): `terms` was zeroed by
 * the caller (otherwise this call zeroes it first). */
int dsu_sample_losses(const float* sdf_all, const float* grad_all, int64_t n_samples,
                      int64_t n_random, float lambda_eikonal, float lambda_sparsity,
                      float sparsity_scale, float lambda_smooth, int32_t accumulate_prefix,
                      float* d_sdf_all, float* d_grad_all, float* terms, void* stream);

/* ------------------------------------------------------------------------------------
 * Style translator (3_style_translator/training/models.py).
 * ---------------------------------------------------------------------------------- */

/* generate_coordinates (models.py:551-604): the 18-channel fixed offset map for an (H,W)
 * image, written as (18,H,W) f32 (batch broadcast is the caller's view). */
int dsu_ric_offsets(int32_t H, int32_t W, float* offsets, void* stream);

/* torchvision.ops.deform_conv2d(input, offset, weight, padding=(1,1)) for a 3x3 kernel,
 * stride 1, dilation 1, one offset group, no bias, no mask (call sites models.py:302-351).
 * input (B,C,H,W) f32; offset (18,H,W) f32 shared over the batch (offset_batch_stride=0)
 * or per image; weight (O,C,3,3); out (B,O,H,W).
 * Optional fused epilogue (scale/shift per output channel = folded eval BatchNorm, then
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 tanh), optional residual added last.
 * in_relu != 0 applies ReLU to the input as it is read (the resnet blocks' leading
 * nonlinearity_0, models.py:321) without a separate pass. */
int dsu_deform_conv3x3_fwd(const float* input, const float* offset, int64_t offset_batch_stride,
                           const float* weight, int32_t B, int32_t C, int32_t H, int32_t W,
                           int32_t O, int32_t in_relu, const float* ep_scale,
                           const float* ep_shift, int32_t act, const float* residual, float* out,
                           void* stream);

/* nn.Conv2d forward, NCHW f32, square kernel k in {1,3,7} with stride in {1,2} (GeneratorJ,
 * models.py:41-129; PerceptualVGG19 features 0/2/5, models.py:536-544) or k = 4 with stride in
 * {1,2} (DiscriminatorN_IN, models.py:441-462), zero padding, same fused epilogue.  bias may be
 * NULL. */
int dsu_conv2d_fwd(const float* input, const float* weight, const float* bias, int32_t B,
                   int32_t C, int32_t H, int32_t W, int32_t O, int32_t k, int32_t stride,
                   int32_t pad, int32_t in_relu, const float* ep_scale, const float* ep_shift,
                   int32_t act, const float* residual, float* out, void* stream);

/* Evaluation-time variants of the two convolutions above on the bf16 MFMA with "bf16 x 3"
 * operands (x = x_hi + x_mid + O(2^-16 |x|); products a_hi b_hi + a_hi b_mid + a_mid b_hi with
 * f32 accumulation: relative error ~2^-15 per product — finer than the TF32 arithmetic
 * torch.backends.cudnn.allow_tf32 = True gives the reference's nn.Conv2d by default).  Same
 * tensors, epilogue and argument meaning; the weight comes packed:
 *   dsu_conv_x3_packed_elems(O, C, k)   number of uint16 elements of EACH of w_hi / w_mid,
 *   dsu_conv_x3_pack_weights            (O,C,k,k) f32 -> (roundup(O,32), ceil(C/16), k*k, 16) bf16
 *                                       hi and mid parts, zero padded; 16-byte aligned buffers.
 * k in {1,3,7} (stride 2 for k = 3 only); others DSU_EUNSUP.  Used by GeneratorJ / GeneratorJ_RIC
 * in eval mode (models.py:41-129, :302-351); training keeps the exact-f32 kernels. */
int64_t dsu_conv_x3_packed_elems(int32_t O, int32_t C, int32_t k);
int dsu_conv_x3_pack_weights(const float* weight, int32_t O, int32_t C, int32_t k, uint16_t* w_hi,
                             uint16_t* w_mid, void* stream);
int dsu_deform_conv3x3_fwd_x3(const float* input, const float* offset, int64_t offset_batch_stride,
                              const uint16_t* w_hi, const uint16_t* w_mid, int32_t B, int32_t C,
                              int32_t H, int32_t W, int32_t O, int32_t in_relu,
                              const float* ep_scale, const float* ep_shift, int32_t act,
                              const float* residual, float* out, void* stream);
int dsu_conv2d_fwd_x3(const float* input, const uint16_t* w_hi, const uint16_t* w_mid,
                      const float* bias, int32_t B, int32_t C, int32_t H, int32_t W, int32_t O,
                      int32_t k, int32_t stride, int32_t pad, int32_t in_relu,
                      const float* ep_scale, const float* ep_shift, int32_t act,
                      const float* residual, float* out, void* stream);

/* Exact-f32 evaluation variants on the same im2col-in-registers structure (f32 MFMA,
 * v_mfma_f32_32x32x2_f32: every product and sum an f32 fmaf, the arithmetic of
 * torchvision.ops.deform_conv2d's im2col + f32 addmm with TF32 off, which is PyTorch's default
 * for matmuls: models.py:302-351 via test_stage1.py:42-71; and of the f32 ONNX session behind
 * mv.py:17-18,134-150).  Same tensors, epilogue and argument meaning as dsu_deform_conv3x3_fwd /
 * dsu_conv2d_fwd; the weight comes packed as (roundup(O,32), ceil(C/16), k*k, 16) f32, zero padded
 * (dsu_conv_x3_packed_elems(O, C, k) floats, 16-byte aligned).  C % 8 == 0 (callers pad with zero
 * channels); k in {1,3} (stride 2 for k = 3 only), others DSU_EUNSUP -> dsu_conv2d_fwd. */
int dsu_conv_f32p_pack_weights(const float* weight, int32_t O, int32_t C, int32_t k, float* w_packed,
                               void* stream);
int dsu_deform_conv3x3_fwd_f32p(const float* input, const float* offset, int64_t offset_batch_stride,
                                const float* w_packed, int32_t B, int32_t C, int32_t H, int32_t W,
                                int32_t O, int32_t in_relu, const float* ep_scale,
                                const float* ep_shift, int32_t act, const float* residual, float* out,
                                void* stream);
int dsu_conv2d_fwd_f32p(const float* input, const float* w_packed, const float* bias, int32_t B,
                        int32_t C, int32_t H, int32_t W, int32_t O, int32_t k, int32_t stride,
                        int32_t pad, int32_t in_relu, const float* ep_scale, const float* ep_shift,
                        int32_t act, const float* residual, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Style translator, per-character TRAINING (3_style_translator/training/trainers.py:140-192:
 * autograd through GeneratorJ / GeneratorJ_RIC, DiscriminatorN_IN and PerceptualVGG19 on
 * batch_size x C x 32 x 32 patches, configs/config_stage{1,2}.yaml).  SURVEY.md 8f-1.
 * The data gradient of nn.Conv2d is dsu_conv2d_fwd on dout with flipped/transposed weights.
 * ---------------------------------------------------------------------------------- */

/* Fixed sampling table of the 3x3 deformable convolution for an (18,H,W) offset map shared by
 * the batch (models.py:297-300: coords_k depend on the resolution only): 9 records per pixel
 * with the four clamped corner offsets and bilinear weights, computed with the forward
 * kernel's own arithmetic.  table: caller-owned device buffer of dsu_deform_tap_table_bytes
 * (0 = invalid arguments). */
int64_t dsu_deform_tap_table_bytes(int32_t H, int32_t W);
int dsu_deform_tap_table(const float* offset, int32_t H, int32_t W, void* table, void* stream);

/* Weight gradient of nn.Conv2d (tap_table == NULL; k in {1,3,4,7}, any stride/pad) or of
 * torchvision.ops.deform_conv2d with the fixed offsets of tap_table (k = 3, stride 1, pad 1):
 *   dweight[o][c][ty][tx] (= or +=, accumulate) sum_{b,pixel} dout[b][o][pixel] * col[b][c][tap][pixel]
 * input (B,C,H,W), dout (B,O,OH,OW), O <= 128.  workspace: per-slice partial sums, size from
 * dsu_conv2d_wgrad_workspace_bytes (same B,C,O,k and the OUTPUT size OH,OW); summed in a
 * fixed order, so results are run-to-run identical. */
int64_t dsu_conv2d_wgrad_workspace_bytes(int32_t deform, int32_t B, int32_t C, int32_t O,
                                         int32_t OH, int32_t OW, int32_t k);
int dsu_conv2d_wgrad(const float* input, const float* dout, const void* tap_table, int32_t B,
                     int32_t C, int32_t H, int32_t W, int32_t O, int32_t k, int32_t stride,
                     int32_t pad, float* workspace, float* dweight, int32_t accumulate,
                     void* stream);

/* Input gradient of the fixed-offset deformable convolution, second half: dcol (planes = B*C,
 * 9, npix) = W^T dout (a 1x1 dsu_conv2d_fwd with the (C*9, O) transposed weight) is pulled back
 * through the transposed sampling operator given as CSR over INPUT pixels: row q lists
 * src = tap * npix + output_pixel and the bilinear weight of every sample that read q.
 *   dx[plane][q] = sum_e wgt[e] * dcol[plane][src[e]],  e in [rowptr[q], rowptr[q+1]) */
int dsu_deform_conv3x3_dgrad_gather(const float* dcol, const int32_t* rowptr, const int32_t* src,
                                    const float* wgt, int64_t planes, int32_t npix, float* dx,
                                    void* stream);

/* nn.BatchNorm2d in training mode (instance = 0: statistics over (batch, hw) per channel;
 * running_mean/var, when given, take `stat_updates` momentum updates with the unbiased batch
 * variance) or nn.InstanceNorm2d without affine/running stats (instance = 1, DiscriminatorN_IN
 * models.py:436-439), followed by act (0 none, 1 ReLU, 2 LeakyReLU(0.2)).  x,y (B,C,HW) f32;
 * save_mean/save_invstd: one value per channel (instance = 0) or per (image, channel). */
typedef struct dsu_norm_cfg {
  int32_t batch, channels, hw;
  int32_t instance;
  int32_t act;
  int32_t stat_updates;
  float eps;
  float momentum;
} dsu_norm_cfg;
int dsu_norm_train_fwd(const dsu_norm_cfg* cfg, const float* x, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float* y,
                       float* save_mean, float* save_invstd, void* stream);
/* Backward of the above through the activation (derivative taken from the output y):
 * dx, and for instance = 0 dgamma/dbeta (C) when non-NULL. */
int dsu_norm_train_bwd(const dsu_norm_cfg* cfg, const float* x, const float* y, const float* dy,
                       const float* gamma, const float* save_mean, const float* save_invstd,
                       float* dx, float* dgamma, float* dbeta, void* stream);

/* out[c] = sum over (b, hw) of x[b][c][hw]: the bias gradient of a convolution. */
int dsu_channel_sum(const float* x, int32_t B, int32_t C, int32_t HW, float* out, void* stream);

/* Activations (act: 1 ReLU, 2 LeakyReLU(0.2), 3 tanh) and their backward from the OUTPUT y. */
int dsu_act_fwd(const float* x, float* y, int64_t n, int32_t act, void* stream);
int dsu_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int32_t act, void* stream);

/* nn.MaxPool2d(2,2) (GeneratorJ_RIC.maxpool models.py:214, VGG19 features[4]) on `planes`
 * images of H x W (H, W even); backward routes to the first maximum of each window. */
int dsu_maxpool2_fwd(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream);
int dsu_maxpool2_bwd(const float* x, const float* dy, float* dx, int64_t planes, int32_t H,
                     int32_t W, void* stream);

/* nn.Upsample(scale_factor=2) nearest (UpsamplingLayer, models.py:8-14): H x W -> 2H x 2W. */
int dsu_upsample2_fwd(const float* x, float* y, int64_t planes, int32_t H, int32_t W,
                      void* stream);
int dsu_upsample2_bwd(const float* dy, float* dx, int64_t planes, int32_t H, int32_t W,
                      void* stream);

/* nn.L1Loss (kind 0) / nn.MSELoss (kind 1) terms of trainers.py:97-135 before the division by
 * n: partial256[256] partial sums of |x - t| or (x - t)^2 (t = target[i], or target_const when
 * target is NULL); grad (may be NULL) = d/dx of (grad_scale * sum). */
int dsu_pair_loss(const float* x, const float* target, float target_const, int64_t n,
                  int32_t kind, float grad_scale, float* grad, float* partial256, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-view diffusion UNet (2_charactor_reconstructor/mvdiffusion/models).
 * ---------------------------------------------------------------------------------- */

/* xformers.ops.memory_efficient_attention(q, k, v, attn_bias=None) as called by
 * XFormersMVAttnProcessor (transformer_mv2d.py:802) and XFormersJointAttnProcessor (:890),
 * WITHOUT materialising the repeated / concatenated K,V (transformer_mv2d.py:785-786,
 * 878-883): the key/value sequence of query batch b is the concatenation of S segments of
 * seg_len tokens; segment s of query batch b lives at batch index seg_batch[b*S + s] of K/Vt.
 *   q, out : f16, element strides q_strides/o_strides = {batch, token, head}, d contiguous
 *   k      : f16, strides {batch, token, head}, d contiguous
 *   vt     : f16 V TRANSPOSED per head, strides {batch, head, d}, token contiguous
 *   d in {40, 64, 80, 160}; scale = softmax scale (1/sqrt(d) for xformers' default).
 * f32 accumulation, f16 output. */
int dsu_mv_attention_fwd(const void* q, const void* k, const void* vt, void* out,
                         const int32_t* seg_batch, int32_t Bq, int32_t H, int32_t Nq, int32_t d,
                         int32_t S, int32_t seg_len, const int64_t* q_strides,
                         const int64_t* k_strides, const int64_t* vt_strides,
                         const int64_t* o_strides, float scale, void* stream);

/* nn.Conv2d forward on NHWC f16 activations with f32 accumulation (diffusers ResnetBlock2D /
 * Downsample2D / Upsample2D / conv_in / conv_out convolutions, unet_mv2d_blocks.py:528,649,
 * 688,798,839; unet_mv2d_condition.py:290,623).
 *   input (B,H,W,C) f16; weight_okc (O, k*k, C) f16 = the nn.Conv2d weight (O,C,k,k)
 *   permuted to (O,k,k,C) once by the caller; bias (O) f16 or NULL.
 *   upsample2x != 0: the convolution reads the nearest-neighbour x2 upsampled input
 *   (Upsample2D = F.interpolate(scale 2, nearest) + conv) without materialising it.
 *   addvec (B,O) f16 or NULL: added per image and channel (time-embedding projection of
 *   ResnetBlock2D); residual (B,OH,OW,O) f16 or NULL: added last.  C % 8 == 0. */
int dsu_conv2d_nhwc_f16_fwd(const void* input, const void* weight_okc, const void* bias,
                            int32_t B, int32_t H, int32_t W, int32_t C, int32_t O, int32_t k,
                            int32_t stride, int32_t pad, int32_t upsample2x, const void* addvec,
                            const void* residual, void* out, void* stream);
/* Same convolution with split-K: the k*k*C reduction is divided over `split_k` workgroups per
 * output tile (f32 partials in `workspace`, summed by a second kernel that also applies bias /
 * addvec / residual).  For the 8x8 and 4x4 levels of the UNet the output tiles alone occupy
 * 20-120 of the 256 CUs.  dsu_conv2d_nhwc_f16_split_k returns the factor the library would pick
 * (1 = plain kernel), dsu_conv2d_nhwc_f16_workspace_bytes the f32 scratch it needs. */
int32_t dsu_conv2d_nhwc_f16_split_k(int32_t B, int32_t H, int32_t W, int32_t C, int32_t O,
                                    int32_t k, int32_t stride, int32_t pad, int32_t upsample2x);
int64_t dsu_conv2d_nhwc_f16_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t O, int32_t k,
                                            int32_t stride, int32_t pad, int32_t upsample2x,
                                            int32_t split_k);
int dsu_conv2d_nhwc_f16_fwd_ws(const void* input, const void* weight_okc, const void* bias,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t O, int32_t k,
                               int32_t stride, int32_t pad, int32_t upsample2x,
                               const void* addvec, const void* residual, void* out,
                               int32_t split_k, void* workspace, int64_t workspace_bytes,
                               void* stream);
/* Split-K with the reduction inside the kernel: `tile_counters` is a caller-owned, ZEROED int32
 * buffer of at least (output tiles) entries that stays allocated between calls; the last workgroup
 * of a tile to finish sums the split_k partial tiles (in z order) and runs the epilogue, then
 * resets the tile's counter — no separate reduce launch.  One buffer serves every launch issued
 * on ONE stream (launches on different streams need their own).  NULL = the reduce kernel. */
int dsu_conv2d_nhwc_f16_fwd_fx(const void* input, const void* weight_okc, const void* bias,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t O, int32_t k,
                               int32_t stride, int32_t pad, int32_t upsample2x,
                               const void* addvec, const void* residual, void* out,
                               int32_t split_k, void* workspace, int64_t workspace_bytes,
                               int32_t* tile_counters, int64_t n_counters, void* stream);
int dsu_gemm_f16_fwd_fx(const void* x, const void* w, const void* bias, int64_t M, int32_t K,
                        int32_t N, const void* residual, void* out, int32_t tokens_per_image,
                        int32_t split_k, void* workspace, int64_t workspace_bytes,
                        int32_t* tile_counters, int64_t n_counters, void* stream);

/* nn.Linear forward of the UNet's transformer blocks and embedding MLPs on the same MFMA kernel
 * (a 1x1 convolution over M rows): Attention.to_q / to_k / to_out, FeedForward.net.2,
 * TransformerMV2DModel.proj_in / proj_out, TimestepEmbedding, ResnetBlock2D.time_emb_proj
 * (transformer_mv2d.py:447-483, 304-370; unet_mv2d_condition.py:313-319,374).
 *   x (M, K) f16 row-major, w (N, K) f16 = the nn.Linear weight as stored, bias (N) or NULL,
 *   residual (M, N) or NULL (added last), out (M, N) f16, f32 accumulation.  K % 8 == 0.
 *   tokens_per_image > 0 (M a multiple of it): out is written TRANSPOSED per image,
 *   out[(m / tokens) * N * tokens + n * tokens + m % tokens] — Attention.to_v, so that
 *   dsu_mv_attention_fwd reads V^T rows in place; no residual / split-K in that form.
 *   split_k / workspace as for dsu_conv2d_nhwc_f16_fwd_ws (dsu_gemm_f16_split_k = the library's
 *   choice, dsu_gemm_f16_workspace_bytes = f32 scratch). */
int32_t dsu_gemm_f16_split_k(int64_t M, int32_t K, int32_t N);
int64_t dsu_gemm_f16_workspace_bytes(int64_t M, int32_t N, int32_t split_k);
int dsu_gemm_f16_fwd(const void* x, const void* w, const void* bias, int64_t M, int32_t K, int32_t N,
                     const void* residual, void* out, int32_t tokens_per_image, int32_t split_k,
                     void* workspace, int64_t workspace_bytes, void* stream);
/* diffusers FeedForward(activation_fn="geglu") first layer fused with its activation
 * (transformer_mv2d.py:483 -> GEGLU.forward): w (2N, K), bias (2N) or NULL;
 * out (M, N) = (x w[:N]^T + b[:N]) * gelu_erf(x w[N:]^T + b[N:]); both halves are rounded to f16
 * before the product, as the unfused nn.Linear output is.  K % 8 == 0, N % 4 == 0. */
int dsu_gemm_geglu_fwd(const void* x, const void* w, const void* bias, int64_t M, int32_t K,
                       int32_t N, void* out, void* stream);

/* nn.GroupNorm(G, C, eps) on NHWC f16 (+ optional fused SiLU): diffusers ResnetBlock2D
 * norm1/norm2 + nonlinearity, TransformerMV2DModel.norm (transformer_mv2d.py:304),
 * conv_norm_out + conv_act (unet_mv2d_condition.py:1046-1048).  x/out (B,HW,C) f16;
 * gamma/beta (C) f16; stats_ws: caller-owned f32 scratch of B*G*2 floats. */
int dsu_groupnorm_nhwc_f16(const void* x, const void* gamma, const void* beta, int32_t B,
                           int32_t HW, int32_t C, int32_t G, float eps, int32_t silu,
                           float* stats_ws, void* out, void* stream);

/* nn.LayerNorm(C) over the last dimension of (rows, C) f16 (BasicMVTransformerBlock
 * norm1 / norm_joint_mid / norm2 / norm3, transformer_mv2d.py:447-519). */
int dsu_layernorm_f16(const void* x, const void* gamma, const void* beta, int64_t rows,
                      int32_t C, float eps, void* out, void* stream);

/* diffusers GEGLU (FeedForward activation_fn="geglu", transformer_mv2d.py:483):
 * h (rows, 2*D) f16 = proj output; out[r][j] = h[r][j] * gelu_erf(h[r][D+j]). */
int dsu_geglu_f16(const void* h, int64_t rows, int32_t D, void* out, void* stream);

/* cv2.inpaint(img, mask, radius, cv2.INPAINT_TELEA) on an 8-bit 3-channel HOST image
 * (1_lama_contour_remover/predict.py:63: the predicted contour pixels and the background are
 * filled from the character's own pixels).  Host code, as in the reference: fast marching is a
 * strictly ordered front propagation.  img/out (rows, cols, 3) u8, mask (rows, cols) u8
 * (non-zero = fill); rows, cols >= 3. */
int dsu_inpaint_telea_u8c3(const uint8_t* img, const uint8_t* mask, int32_t rows, int32_t cols,
                           int32_t radius, uint8_t* out);

/* torch.optim.AdamW step on a range of the hash-table parameters (the optimizer the reference
 * configures in configs/neuralangelo-ortho-wmask.yaml:96-110 -> systems/utils.py parse_optimizer;
 * torch's update: p -= lr*wd*p; m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g;
 * p -= (lr / bias_correction1) * m / (sqrt(v) / bias_correction2_sqrt + eps)), fused with the f16
 * image rewrite (img_f16[i] = half(p[i])) and the gradient reset (g = 0).  n floats, n % 4 == 0,
 * 16-byte aligned pointers.  dsu_table_decay: p *= factor (+ image) — the accumulated
 * `1 - lr*wd` factors of steps in which a level of the progressive grid was still masked. */
int dsu_table_adamw(float* p, float* g, float* m, float* v, void* img_f16, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                    float bias_correction2_sqrt, void* stream);
int dsu_table_decay(float* p, void* img_f16, int64_t n, float factor, void* stream);
/* The same AdamW update for up to DSU_ADAMW_MAX_TENSORS small tensors in one launch (the SDF MLP,
 * texture MLP and variance parameters of the three optimizer groups, each with its group's lr and
 * its own bias corrections); `tensors` is a HOST array. */
#define DSU_ADAMW_MAX_TENSORS 24
typedef struct {
  float* p; float* g; float* m; float* v;
  int64_t n;
  float lr, bias_correction1, bias_correction2_sqrt, reserved;
} dsu_adamw_tensor;
int dsu_adamw_multi(const dsu_adamw_tensor* tensors, int32_t count, float beta1, float beta2,
                    float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------------------------
 * Native driver of ONE NSR optimisation step: OrthoNeuSSystem.training_step
 * (2_charactor_reconstructor/instant_nsr/systems/neus_ortho.py:79-169) over
 * NeuSModelTextureMLP.forward_ (instant_nsr/models/neus.py:114-196), the ray batch of
 * preprocess_data (neus_ortho.py:26-77) and the AdamW update of the SDF-MLP / texture-MLP /
 * variance groups (configs/neuralangelo-ortho-wmask.yaml:96-127) sequenced by the library
 * itself: the host-side cost of a step is ~20 kernel launches issued from C instead of ~1.4 ms
 * of interpreter work (as long as the step's device time).  The hash table's own update stays the
 * caller's bookkeeping (levels, lazy decay) with the launch itself inside the step; the occupancy-grid
 * refresh of every 16th step is dsu_nsr_driver_occ_refresh, called by the caller before that step.
 *
 * All device memory is the caller's: parameters, dataset tensors and ONE workspace of
 * dsu_nsr_driver_workspace_bytes(cfg) bytes.  The driver object owns a side stream, three
 * events and 16 bytes of pinned host memory (the next step's rays are drawn, marched, packed
 * and Morton-sorted on the side stream while this step's backward runs; the sample total
 * reaches the host through the pinned words).  Random draws: counter-based Philox4x32-10 keyed
 * by (seed, step) — reproducible, independent of the step's timing; tests inject their own.
 * ---------------------------------------------------------------------------------- */
typedef struct dsu_nsr_driver dsu_nsr_driver;
typedef struct dsu_nsr_driver_cfg {
  dsu_hashgrid_cfg grid;
  float radius;              /* model.radius (1.0) */
  float render_step_size;    /* 1.732 * 2 * radius / num_samples_per_ray (neus.py:59) */
  int32_t cap_points;        /* capacity of the packed sample buffers (2^19: 2x the target) */
  int32_t cap_rays;          /* max_train_num_rays (8192), <= DSU_RAY_LOSS_MAX_RAYS */
  int32_t n_random;          /* 2048 regulariser points (neus.py:155) */
  int32_t sort_bits;         /* Morton bits per axis for the evaluation order (0 = ray-major) */
  int32_t dynamic_ray_sampling; /* neus_ortho.py:88-92 */
  int32_t train_num_samples; /* train_num_rays * num_samples_per_ray (2^18) */
  /* resident dataset (datasets/ortho.py:99-151), f32 contiguous */
  const float *c2w, *origins, *directions, *images, *normals, *masks, *view_weights;
  int32_t V, H, W, image_channels;
  /* parameters (f32, device): SDF MLP with weight norm (network_utils.py:113-138), texture MLP,
   * variance scalar; updated in place */
  float *w0_v, *w0_g, *b0, *w1_v, *w1_g, *b1;
  float* tex[6];
  float* variance;
  /* losses and optimizer (neuralangelo-ortho-wmask.yaml:86-127) */
  dsu_ray_loss_cfg ray_loss;
  float lambda_eikonal, lambda_sparsity, sparsity_scale, lambda_smooth;
  float beta1, beta2, adam_eps, weight_decay;
  uint64_t seed;
  void* workspace;
  int64_t workspace_bytes;
} dsu_nsr_driver_cfg;

typedef struct dsu_nsr_step_args {
  int64_t step;              /* global step: RNG counter and buffer parity */
  int32_t n_rays;            /* rays of this step */
  int32_t prefetch_next;     /* 1: enqueue step+1's samples on the side stream */
  uint32_t active_levels;    /* ProgressiveBandHashGrid.current_level */
  float eps;                 /* finite-difference step (geometry.py:196-215) */
  float cos_anneal_ratio;
  float lr_geometry, lr_texture, lr_variance;
  int32_t adam_step;         /* 1-based count of small-tensor updates (bias corrections) */
  int32_t randomized;        /* stratified jitter on (model.randomized) */
  int32_t refresh_effective; /* 1: parameters were written from outside since the last step */
  int32_t occ_res;
  const uint8_t* occ_binary; /* current occupancy grid (NULL = no pruning) */
  const void* table_img;     /* f16 image of the hash table */
  float* table_grad;         /* (entries,2) f32, accumulated into */
  /* injected draws for THIS step (tests; any NULL = the driver's own draw) */
  const int64_t *inj_index, *inj_x, *inj_y;
  const float *inj_jitter, *inj_pts_random, *inj_perturb;
  /* tests: a whole ray batch as OrthoNeuSSystem.preprocess_data returns it (neus_ortho.py:26-82) in
   * place of the (index, x, y) gathers: rays (n,6) = [origin | direction], rgb (n,3), normal (n,3),
   * mask, cosines, view_weights (n); inj_rays NULL = the dataset path; with it the other five are
   * required */
  const float *inj_rays, *inj_rgb, *inj_normal, *inj_mask, *inj_cosines, *inj_view_weights;
  /* the hash table's AdamW step (dsu_table_adamw), launched by the driver behind the backward:
   * master parameters and moments, number of floats of the active levels, this step's lr and
   * bias corrections (the caller keeps the level / decay bookkeeping); table_p NULL = skip */
  float *table_p, *table_m, *table_v;
  int64_t table_n;
  float table_lr, table_bc1, table_bc2_sqrt, table_eps, table_wd;
  /* outputs (host) */
  int32_t out_n_samples, out_max_count, out_next_n_rays;
  /* optional: 8 floats of DEVICE memory that receive a copy of this step's loss terms (the
   * driver's own two sets are reused two steps later); written by the step's last kernel */
  float* terms_out;
} dsu_nsr_step_args;

/* The random draws of step `step` (neus_ortho.py:31-41: view / pixel triples of the ray batch;
 * nerfacc's stratified jitter; neus.py:155-160: n_random points in [-1,1)^3 and their N(0,1)
 * perturbation) from Philox4x32-10 with key `seed` and counter (element, stream, step): the
 * same (seed, step) gives the same draws whatever ran before.  index in [0,V), x in [0,W), y in
 * [0,H) as int64 (what dsu_ortho_ray_batch takes), jitter in [0,1). */
int dsu_nsr_draws(uint64_t seed, int64_t step, int32_t n_rays, int32_t V, int32_t H, int32_t W,
                  int64_t* index, int64_t* x, int64_t* y, float* jitter, int32_t n_random,
                  float* pts_random, float* perturb, void* stream);
int64_t dsu_nsr_driver_workspace_bytes(const dsu_nsr_driver_cfg* cfg);
int dsu_nsr_driver_create(const dsu_nsr_driver_cfg* cfg, dsu_nsr_driver** out);
void dsu_nsr_driver_destroy(dsu_nsr_driver* d);
/* One step on `main_stream`.  DSU_EUNSUP: the step produced more samples than cap_points (or a
 * ray more than the march scratch row holds); out_n_samples / out_max_count say how many. */
int dsu_nsr_driver_step(dsu_nsr_driver* d, dsu_nsr_step_args* args, void* main_stream);
/* Device pointer to two sets of 8 floats; the set of step s starts at 8 * (s & 1) and holds its 7
 * loss terms: rgb_mse, rgb_l1, normal, mask, eikonal, sparsity, normal_smooth (each already
 * multiplied by its lambda).  A set is reused (zeroed) at the end of the following step. */
const float* dsu_nsr_driver_terms(const dsu_nsr_driver* d);
/* Device pointer to the first (second = 0) or second (1) AdamW moments of the 13 small tensors,
 * 7 902 floats in the order lin0.weight_v (64x23), lin0.weight_g (64), lin0.bias (64),
 * lin1.weight_v (13x64), lin1.weight_g (13), lin1.bias (13), the texture MLP's w0 b0 w1 b1 w2 b2,
 * variance (1).  After ONE step from zeroed moments m / (1 - beta1) is that step's gradient per
 * parameter — how tests/test_gpu_nsr_native.py compares the library-sequenced step with the
 * reference's loss.backward() (systems/neus_ortho.py:79-169). */
const float* dsu_nsr_driver_adam_moments(const dsu_nsr_driver* d, int32_t second);
/* HIP-event timing of the two geometry launches of a step (family 0: dsu_sdf_fd_fwd_sorted,
 * 1: dsu_sdf_fd_bwd_sorted) on the stream they run on: enable = n > 0 times the steps whose index
 * is a multiple of n (resets; 1 = every step — the four event records cost ~28 us of main-queue time
 * per step), 0 disables; then read the number of timed launches, their summed duration and their
 * algorithmic bytes (points x (7 x active_levels x 8 x 4 + 84)). */
int dsu_nsr_driver_timing(dsu_nsr_driver* d, int32_t enable);
int dsu_nsr_driver_timing_read(dsu_nsr_driver* d, int32_t family, int64_t* launches,
                               double* total_ms, double* alg_bytes);
/* Algorithmic MLP flops of the same launches (points x 2 x (7 x 64 x (3 + 2 active_levels) + 64 x 19),
 * three times that for the backward family): these kernels are bound by the f32 arithmetic of
 * VanillaMLP (network_utils.py:107-138), not by the table traffic the contract prices them on. */
int dsu_nsr_driver_timing_flops(dsu_nsr_driver* d, int32_t family, double* mlp_flops);
/* dsu_occgrid_refresh with the driver's own effective weights, inv_s, grid, radius, step size, aabb
 * and seed filled in (args->grid / mlp / inv_s / aabb / seed / radius / render_step_size are
 * ignored): the refresh of every 16th step without leaving the native path. */
int dsu_nsr_driver_occ_refresh(dsu_nsr_driver* d, const dsu_occgrid_refresh_args* args,
                               void* main_stream);
/* Wait for the side stream and drop a pending prefetch (before the caller changes the ray count,
 * the dataset or the occupancy grid behind the driver's back). */
int dsu_nsr_driver_sync(dsu_nsr_driver* d);

/* mcubes.smooth on the export's binary volume (MarchingCubeHelper.forward,
 * instant_nsr/models/geometry.py:57-58 -> PyMCubes' constrained smoothing): the weighted-Jacobi
 * iteration on the compacted band voxels, float64.  nbr (6, nv) int32: slot of the -x,+x,-y,+y,
 * -z,+z neighbour or -1 (outside the band: folds onto the diagonal); lower / upper (nv) f64: the
 * per-voxel bounds of PyMCubes' projection (`np.maximum(x, lower)` then `np.minimum(x, upper)`;
 * +-infinity = unbounded); x (nv) in/out; y caller-owned scratch of 3*nv doubles.
 * dsu_smooth_iterate runs `iters` iterations x <- proj(w * (-D^-1 R x) + (1 - w) x);
 * dsu_smooth_energy writes dsu_smooth_energy_partials() partial sums of x . Q x (the caller adds
 * them in order and halves: the energy of the stopping test). */
int32_t dsu_smooth_energy_partials(void);
int dsu_smooth_iterate(const int32_t* nbr, int64_t nv, const double* lower, const double* upper,
                       double weight, int32_t iters, double* x, double* y, void* stream);
int dsu_smooth_energy(const int32_t* nbr, int64_t nv, const double* x, double* y,
                      double* partials, void* stream);

/* The signed distance transform mcubes.smooth starts from (PyMCubes: scipy's
 * distance_transform_edt on both classes, +-0.5 at the boundary voxels; geometry.py:57-58), in the
 * band that matters: exact wherever the other class is within R voxels (1 <= R <= 8).
 * binary (X,Y,Z) bytes (non-zero = inside), z minor.  Per voxel d2 = min(squared distance to the
 * nearest voxel of the OTHER class, (R+1)^2) as an integer; the outputs go through two caller-built
 * tables: dist[v] = value_table[(inside ? 0 : (R+1)^2 + 1) + d2] (f64: the caller applies sqrt,
 * the far-field value, the half-voxel shift and the sign on the host), band[v] = band_table[d2].
 * workspace: dsu_volume_band_distance_workspace_bytes(X, Y, Z) bytes of device scratch. */
int64_t dsu_volume_band_distance_workspace_bytes(int32_t X, int32_t Y, int32_t Z);
int dsu_volume_band_distance(const uint8_t* binary, int32_t X, int32_t Y, int32_t Z, int32_t R,
                             const double* value_table, const uint8_t* band_table, double* dist,
                             uint8_t* band, void* workspace, int64_t workspace_bytes, void* stream);
/* mcubes.marching_cubes' configuration byte per cube (PyMCubes marchingcubes.h: `if (v[m] <=
 * isovalue) cubeindex |= 1 << m`, geometry.py:59): volume (X,Y,Z) f64 -> cube (X-1,Y-1,Z-1) bytes. */
int dsu_mc_cube_index(const double* volume, int32_t X, int32_t Y, int32_t Z, double isovalue,
                      uint8_t* cube, void* stream);

/* ------------------------------------------------------------------------------------
 * Mesh post-processing of the export (save_mesh, instant_nsr/utils/mesh_utils.py:25-73): the
 * geometric queries of color_projection (utils/coloring_utils.py:91-138) and get_offset_mask
 * (utils/thinning_utils.py:96-193).  SURVEY.md 8f-2.
 * ---------------------------------------------------------------------------------- */

/* Uniform xy grid over the triangles (tris: (n_faces, 3 vertices, 3) f32): cell (cx, cy) =
 * floor((x - x0) / cell), clamped to [0, g); a triangle is listed in every cell its xy bounding
 * box overlaps.  Counting sort in two launches around the caller's exclusive prefix sum:
 * dsu_zgrid_count adds to counts (g*g int32, zeroed by the caller); dsu_zgrid_fill writes the
 * face ids into items at offsets[cell] (+ a zeroed cursor array it increments). */
int dsu_zgrid_count(const float* tris, int64_t n_faces, float x0, float y0, float cell, int32_t g,
                    int32_t* counts, void* stream);
int dsu_zgrid_fill(const float* tris, int64_t n_faces, float x0, float y0, float cell, int32_t g,
                   const int32_t* offsets, int32_t* cursor, int32_t* items, void* stream);
/* mesh_raycast.raycast(origin, (0, 0, sign), mesh=triangles) for n_rays origins (coloring_utils.py:
 * 107-130, thinning_utils.py:101-193): per ray the number of hits, the nearest and the farthest
 * hit (distance t >= 0 along the ray and face id; ties -> smaller face id; no hit: id -1, t 0).
 * A hit = (x, y) inside or on the boundary of the triangle's xy projection and not behind the
 * origin.  self_vertex (optional, with faces (n_faces,3) int32): the mesh vertex a ray starts at;
 * triangles incident to it count as hits at distance exactly 0. */
int dsu_zray_cast(const float* tris, const int32_t* faces, int64_t n_faces, float x0, float y0,
                  float cell, int32_t g, const int32_t* offsets, const int32_t* items,
                  const float* origins, int64_t n_rays, int32_t sign, const int32_t* self_vertex,
                  int32_t* hit_count, float* t_near, int32_t* face_near, float* t_far,
                  int32_t* face_far, void* stream);
/* MaskRenderer.render (coloring_utils.py:22-41; pytorch3d orthographic rasteriser, zbuf > -1):
 * silhouette of `scale` x the mesh seen from +z on a res x res image, 255 where a pixel centre
 * (x = (2 col + 1) / res - 1, y = 1 - (2 row + 1) / res) is covered; mask zeroed by the caller. */
int dsu_raster_mask(const float* tris, int64_t n_faces, float scale, int32_t res, uint8_t* mask,
                    void* stream);
/* cv2.erode(src, cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (ksize, ksize))) (coloring_utils.py:
 * 61-64), single channel uint8, default border. */
int dsu_erode_ellipse_u8(const uint8_t* src, int32_t H, int32_t W, int32_t ksize, uint8_t* dst,
                         void* stream);
/* interpolate_rgb (coloring_utils.py:43-58): for every query point the 8 nearest known points in
 * the xy plane (scipy cKDTree.query(k=8); coordinates float64), colours blended with weights
 * 1 / (d + 1e-6).  The
 * known points are binned on an xy grid by dsu_point_bin_count / _fill (same protocol as the
 * triangle grid). */
int dsu_point_bin_count(const float* xy, int64_t n, float x0, float y0, float cell, int32_t g,
                        int32_t* counts, void* stream);
int dsu_point_bin_fill(const float* xy, int64_t n, float x0, float y0, float cell, int32_t g,
                       const int32_t* offsets, int32_t* cursor, int32_t* items, void* stream);
int dsu_knn8_blend(const double* query_xy, int64_t n_query, const double* known_xy,
                   const float* known_rgb, int64_t n_known, float x0, float y0, float cell,
                   int32_t g, const int32_t* offsets, const int32_t* items, float* out_rgb,
                   void* stream);

/* remesh() (instant_nsr/utils/mesh_utils.py:10-22, called by models/geometry.py:63-64 with
 * face_count 50000): quadric edge-collapse decimation of a triangle mesh down to `target_faces`
 * triangles.  HOST function on HOST arrays (as in the reference, where trimesh hands the mesh to
 * Open3D's simplify_quadric_decimation): verts (n_verts,3) float64, faces (n_faces,3) int32.
 * out_verts / out_faces must hold n_verts / n_faces rows; the counts come back through
 * out_n_verts / out_n_faces.  boundary_weight: weight of the boundary-edge planes (Open3D: 1.0).
 * flags bit 0: skip the link-condition test (collapses may then create non-manifold edges, as
 * Open3D's can).  Collapses stop at the first face count <= target_faces (a collapse removes two
 * triangles, one on a boundary edge); more remain when no admissible collapse is left. */
int dsu_mesh_decimate_quadric(const double* verts, int64_t n_verts, const int32_t* faces,
                              int64_t n_faces, int64_t target_faces, double boundary_weight,
                              int32_t flags, double* out_verts, int64_t* out_n_verts,
                              int32_t* out_faces, int64_t* out_n_faces);

/* The same, seeded with per-vertex quadrics (n_verts,10: a00 a01 a02 a11 a12 a22 b0 b1 b2 c of
 * [A b; b^T c]) instead of computing them from the input triangles — the finishing pass after
 * dsu_mesh_decimate_parallel, whose collapses have accumulated them. */
int dsu_mesh_decimate_quadric_q(const double* verts, int64_t n_verts, const int32_t* faces,
                                int64_t n_faces, int64_t target_faces, double boundary_weight,
                                int32_t flags, const double* vertex_quadrics, double* out_verts,
                                int64_t* out_n_verts, int32_t* out_faces, int64_t* out_n_faces);

/* The bulk of the same `remesh` call (mesh_utils.py:10-22; geometry.py:63-64 hands it the 512^3
 * marching-cubes mesh, 1-3 M triangles) ON THE DEVICE: rounds of independent quadric edge collapses
 * with the serial function's admissibility rules (csrc/mesh_decimate_gpu.hip).  verts (n_verts,3)
 * f64 and faces (n_faces,3) i32 are DEVICE arrays updated in place: surviving vertices keep their
 * index (positions move), the live triangles are compacted to the first *out_n_faces rows.
 * Rounds run while the live count is above stop_faces (at most max_rounds) and a round never takes
 * the count below floor_faces (<= stop_faces); the caller finishes with
 * dsu_mesh_decimate_quadric_q(out_quadrics) which lands on the exact target.  out_quadrics:
 * (n_verts,10) f64 device or NULL.  out_stats: host int32[3] = rounds, collapses, rejections, or
 * NULL.  Synchronises the stream (one count per round is read back).  flags as above. */
int64_t dsu_mesh_decimate_parallel_workspace_bytes(int64_t n_verts, int64_t n_faces);
int dsu_mesh_decimate_parallel(double* verts, int64_t n_verts, int32_t* faces, int64_t n_faces,
                               int64_t stop_faces, int64_t floor_faces, double boundary_weight,
                               int32_t flags, int32_t max_rounds, double* out_quadrics,
                               int64_t* out_n_faces, int32_t* out_stats, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* One implicit step of trimesh.smoothing.filter_laplacian (save_mesh, mesh_utils.py:42-45):
 * solves (I + lamb (I - L)) X = rhs for the umbrella operator L given as CSR neighbour lists
 * (offsets (n_verts+1), neighbours: int32 device), rhs / x / tmp (n_verts,3) float64 device.
 * x holds the starting guess on entry (rhs itself is a good one) and the solution on return;
 * `sweeps` Jacobi sweeps (even; the error contracts by lamb / (1 + lamb) per sweep). */
int dsu_umbrella_implicit_solve(const int32_t* offsets, const int32_t* neighbours, int64_t n_verts,
                                double lamb, const double* rhs, double* x, double* tmp,
                                int32_t sweeps, void* stream);

/* Image-side host steps of thinning_processing (instant_nsr/utils/thinning_utils.py:205-218) on
 * HOST arrays (H,W) uint8, non-zero = character:
 *   cv2.distanceTransform(mask, cv2.DIST_L2, 5)            -> out (H,W) float32
 *   skimage.morphology.skeletonize(mask, method='lee')     -> out (H,W) uint8, 0 / 255 */
int dsu_distance_transform_l2_5x5(const uint8_t* mask, int32_t H, int32_t W, float* out);
int dsu_skeletonize_lee_2d(const uint8_t* img, int32_t H, int32_t W, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* DSU_HIP_H */
