"""Thin tensor-level wrappers over the C ABI (one function per exported kernel).

These only allocate outputs (torch caching allocator), pass raw device pointers and the
current torch stream, and turn non-zero return codes into DsuError.  No arithmetic
happens in Python.
"""
import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import DsuError, HashGridCfg, NormCfg, SdfMlp, check, lib, ptr, stream


@dataclass(frozen=True)
class HashGridConfig:
    """configs/neuralangelo-ortho-wmask.yaml:52-62 (xyz_encoding_config)."""
    n_levels: int = 10
    n_features_per_level: int = 2
    log2_hashmap_size: int = 19
    base_resolution: int = 32
    per_level_scale: float = 1.3195079107728942

    def c(self):
        return HashGridCfg(self.n_levels, self.n_features_per_level, self.log2_hashmap_size,
                           self.base_resolution, self.per_level_scale)

    def levels(self):
        return _lib.hashgrid_levels(self.c())

    @property
    def n_entries(self):
        return self.levels()["offsets"][self.n_levels]

    @property
    def n_params(self):
        return self.n_entries * self.n_features_per_level

    @property
    def n_output_dims(self):
        return self.n_levels * self.n_features_per_level


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


# ------------------------------------------------------------------ hash grid / SDF network
def hashgrid_encode_fwd(cfg: HashGridConfig, table_f16, x, active_levels):
    x = _f32c(x)
    n = x.shape[0]
    out = torch.empty((n, cfg.n_output_dims), dtype=torch.float16, device=x.device)
    c = cfg.c()
    check(lib().dsu_hashgrid_encode_fwd(C.byref(c), ptr(table_f16, torch.float16), ptr(x), n,
                                        int(active_levels), ptr(out), stream()),
          "dsu_hashgrid_encode_fwd")
    return out


def hashgrid_encode_bwd(cfg: HashGridConfig, x, dout, active_levels, grad_table=None):
    x = _f32c(x)
    dout = _f32c(dout)
    n = x.shape[0]
    if grad_table is None:
        grad_table = torch.zeros(cfg.n_params, dtype=torch.float32, device=x.device)
    c = cfg.c()
    check(lib().dsu_hashgrid_encode_bwd(C.byref(c), ptr(x), ptr(dout), n, int(active_levels),
                                        ptr(grad_table, torch.float32), stream()),
          "dsu_hashgrid_encode_bwd")
    return grad_table


def _mlp_struct(w0, b0, w1, b1):
    for t in (w0, b0, w1, b1):
        ptr(t, torch.float32)
    return SdfMlp(w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr())


def sdf_fwd(cfg, table_f16, mlp, pts, radius, active_levels, n_out=1):
    """mlp = (w0 (64,3+2L), b0 (64), w1 (13,64), b1 (13)) effective f32 weights."""
    pts = _f32c(pts)
    n = pts.shape[0]
    out = torch.empty((n, n_out), dtype=torch.float32, device=pts.device)
    c, m = cfg.c(), _mlp_struct(*mlp)
    check(lib().dsu_sdf_fwd(C.byref(c), ptr(table_f16, torch.float16), C.byref(m), ptr(pts), n,
                            float(radius), int(active_levels), int(n_out), ptr(out), stream()),
          "dsu_sdf_fwd")
    return out


def sdf_fwd_lattice(cfg, table_f16, mlp, lin, x0, nx, lo, span, radius, active_levels, out):
    """SDF on x-slabs [x0, x0 + nx) of the res^3 export lattice, points formed in the kernel
    (p = lin[i] * span + lo per axis); lin (res) f32 device, lo / span three host floats (f32 values);
    out: f32 view of nx * res * res elements, written in place."""
    res = lin.shape[0]
    c, m = cfg.c(), _mlp_struct(*mlp)
    lo3 = (C.c_float * 3)(*[float(v) for v in lo])
    sp3 = (C.c_float * 3)(*[float(v) for v in span])
    if out.numel() != nx * res * res:
        raise DsuError("sdf_fwd_lattice: output size")
    check(lib().dsu_sdf_fwd_lattice(C.byref(c), ptr(table_f16, torch.float16), C.byref(m),
                                    ptr(lin, torch.float32), res, int(x0), int(nx), lo3, sp3,
                                    float(radius), int(active_levels), ptr(out, torch.float32), stream()),
          "dsu_sdf_fwd_lattice")
    return out


def spatial_sort(pts, radius, bits=6):
    """Morton-bin the points of one step (csrc/spatial_sort.hip): returns (pts_sorted, perm) with
    pts_sorted[i] = pts[perm[i]].  Pass both to sdf_fd_fwd / sdf_fd_bwd (perm=...)."""
    pts = _f32c(pts)
    n = pts.shape[0]
    dev = pts.device
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    out = torch.empty_like(pts)
    wbytes = int(lib().dsu_spatial_sort_workspace_bytes(n, int(bits)))
    if wbytes < 0:
        check(wbytes, "dsu_spatial_sort_workspace_bytes")
    ws = torch.empty(max(wbytes, 4) // 4, dtype=torch.int32, device=dev)
    check(lib().dsu_spatial_sort(ptr(pts), n, float(radius), int(bits), ptr(perm), ptr(out),
                                 ptr(ws), wbytes, stream()), "dsu_spatial_sort")
    return out, perm


def sdf_fd_fwd(cfg, table_f16, mlp, pts, radius, eps, active_levels, with_grad=True,
               with_feature=True, with_laplace=True, enc_cache=None, perm=None):
    """enc_cache: True -> also return the feature cache tensor for sdf_fd_bwd (5th value).
    perm (int32, from spatial_sort): `pts` are the sorted points; outputs come back in the
    caller's original row order (row perm[i] = result of pts[i]); the cache stays in sorted order."""
    pts = _f32c(pts)
    n = pts.shape[0]
    dev = pts.device
    cache = None
    if enc_cache:
        nbytes = int(lib().dsu_sdf_fd_enc_cache_bytes(n, int(active_levels)))
        cache = torch.empty(max(nbytes, 4) // 2, dtype=torch.float16, device=dev)
    sdf = torch.empty(n, dtype=torch.float32, device=dev)
    grad = torch.empty((n, 3), dtype=torch.float32, device=dev) if with_grad else None
    feat = torch.empty((n, 13), dtype=torch.float32, device=dev) if with_feature else None
    lap = torch.empty(n, dtype=torch.float32, device=dev) if with_laplace else None
    c, m = cfg.c(), _mlp_struct(*mlp)
    check(lib().dsu_sdf_fd_fwd_sorted(C.byref(c), ptr(table_f16, torch.float16), C.byref(m),
                                      ptr(pts), ptr(perm, torch.int32), n, float(radius),
                                      float(eps), int(active_levels),
                                      ptr(sdf), ptr(grad), ptr(feat), ptr(lap), ptr(cache),
                                      stream()), "dsu_sdf_fd_fwd")
    if enc_cache:
        return sdf, grad, feat, lap, cache
    return sdf, grad, feat, lap


def sdf_fd_bwd(cfg, table_f16, mlp, pts, radius, eps, active_levels, d_sdf, d_grad, d_feature,
               d_laplace, grad_table=None, enc_cache=None, perm=None, extra=None):
    """perm: as in sdf_fd_fwd (`pts` sorted, the upstream gradients in the original row order).
    extra: a DeferredSum (texture_bwd_shaded_partials) carried out by the scatter launch."""
    pts = _f32c(pts)
    n = pts.shape[0]
    dev = pts.device
    w0, b0, w1, b1 = mlp
    if grad_table is None:
        grad_table = torch.zeros(cfg.n_params, dtype=torch.float32, device=dev)
    sizes = [t.numel() for t in (w0, b0, w1, b1)]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)      # one memset
    g = [v.view_as(t) for v, t in zip(torch.split(flat, sizes), (w0, b0, w1, b1))]
    c, m = cfg.c(), _mlp_struct(*mlp)
    d = [None if t is None else _f32c(t) for t in (d_sdf, d_grad, d_feature, d_laplace)]
    wbytes = lib().dsu_sdf_fd_bwd_workspace_bytes(C.byref(c), n)
    if wbytes < 0:
        check(int(wbytes), "dsu_sdf_fd_bwd_workspace_bytes")
    ws = torch.empty(max(int(wbytes), 4) // 4, dtype=torch.float32, device=dev)
    if enc_cache is not None:
        need = int(lib().dsu_sdf_fd_enc_cache_bytes(n, int(active_levels)))
        if enc_cache.numel() * enc_cache.element_size() < need:
            raise DsuError("feature cache smaller than dsu_sdf_fd_enc_cache_bytes")
    if extra is not None:
        # the scatter launch also carries out a deferred partial sum of another kernel
        check(lib().dsu_sdf_fd_bwd_sorted_fold(C.byref(c), ptr(table_f16, torch.float16), C.byref(m),
                                               ptr(pts), ptr(perm, torch.int32), n, float(radius),
                                               float(eps), int(active_levels),
                                               ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]),
                                               ptr(grad_table), ptr(g[0]), ptr(g[1]), ptr(g[2]),
                                               ptr(g[3]), ptr(ws), int(wbytes), ptr(enc_cache), None,
                                               C.byref(extra.record), stream()),
              "dsu_sdf_fd_bwd_sorted_fold")
        return grad_table, g
    check(lib().dsu_sdf_fd_bwd_sorted(C.byref(c), ptr(table_f16, torch.float16), C.byref(m),
                                      ptr(pts), ptr(perm, torch.int32), n, float(radius),
                                      float(eps), int(active_levels),
                                      ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(grad_table),
                                      ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]), ptr(ws),
                                      int(wbytes), ptr(enc_cache), stream()), "dsu_sdf_fd_bwd")
    return grad_table, g


# ------------------------------------------------------------------ hash-table optimizer
def table_adamw(p, g, m, v, img_f16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt):
    """One torch.optim.AdamW update of the first n floats of the table (active levels), fused
    with the f16 image rewrite and the gradient reset."""
    check(lib().dsu_table_adamw(ptr(p, torch.float32), ptr(g, torch.float32), ptr(m, torch.float32),
                                ptr(v, torch.float32), ptr(img_f16, torch.float16), int(n),
                                float(lr), float(beta1), float(beta2), float(eps),
                                float(weight_decay), float(bc1), float(bc2_sqrt), stream()),
          "dsu_table_adamw")


def adamw_multi(entries, beta1, beta2, eps, weight_decay):
    """entries: list of (p, g, m, v, lr, bc1, bc2_sqrt) f32 device tensors / floats (<= 24): one
    launch of torch.optim.AdamW's update for all of them."""
    n = len(entries)
    if n == 0:
        return
    arr = (_lib.AdamwTensor * n)()
    for k, (p, g, m, v, lr, bc1, bc2s) in enumerate(entries):
        a = arr[k]
        a.p, a.g, a.m, a.v = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
        a.n, a.lr, a.bias_correction1, a.bias_correction2_sqrt = p.numel(), lr, bc1, bc2s
    check(lib().dsu_adamw_multi(arr, n, float(beta1), float(beta2), float(eps), float(weight_decay),
                                stream()), "dsu_adamw_multi")


def table_decay(p, img_f16, start, n, factor):
    """p[start:start+n] *= factor (+ f16 image); start, n multiples of 4 floats."""
    check(lib().dsu_table_decay(C.c_void_p(p.data_ptr() + 4 * int(start)),
                                C.c_void_p(img_f16.data_ptr() + 2 * int(start)), int(n),
                                float(factor), stream()), "dsu_table_decay")


# ------------------------------------------------------------------ export: mcubes.smooth
def smooth_iterate(nbr, lower, upper, x, y, weight, iters):
    """`iters` projected weighted-Jacobi iterations on the band voxels, in place on x (f64);
    lower / upper: per-voxel bounds (f64, +-inf = none)."""
    nv = x.shape[0]
    check(lib().dsu_smooth_iterate(ptr(nbr, torch.int32), nv, ptr(lower, torch.float64),
                                   ptr(upper, torch.float64), float(weight), int(iters), ptr(x, torch.float64),
                                   ptr(y, torch.float64), stream()), "dsu_smooth_iterate")
    return x


def smooth_energy(nbr, x, y):
    """x . Q x / 2 (device scalar, f64; fixed summation order)."""
    nv = x.shape[0]
    part = torch.empty(int(lib().dsu_smooth_energy_partials()), dtype=torch.float64,
                       device=x.device)
    check(lib().dsu_smooth_energy(ptr(nbr, torch.int32), nv, ptr(x, torch.float64),
                                  ptr(y, torch.float64), ptr(part), stream()),
          "dsu_smooth_energy")
    return part.sum() / 2


def volume_band_distance(binary, R, value_table, band_table):
    """binary (X,Y,Z) bool/uint8 device volume -> (dist f64 (X,Y,Z), band bool (X,Y,Z)) through the
    caller's tables over d2 = min(squared distance to the nearest voxel of the other class, (R+1)^2):
    value_table (2, (R+1)^2 + 1) f64 (row 0: inside voxels), band_table ((R+1)^2 + 1) uint8."""
    b = binary.contiguous()
    if b.dtype == torch.bool:
        b = b.view(torch.uint8)
    X, Y, Z = b.shape
    dev = b.device
    dist = torch.empty((X, Y, Z), dtype=torch.float64, device=dev)
    band = torch.empty((X, Y, Z), dtype=torch.uint8, device=dev)
    wbytes = int(lib().dsu_volume_band_distance_workspace_bytes(X, Y, Z))
    if wbytes < 0:
        check(wbytes, "dsu_volume_band_distance_workspace_bytes")
    ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    check(lib().dsu_volume_band_distance(ptr(b, torch.uint8), X, Y, Z, int(R),
                                         ptr(value_table, torch.float64), ptr(band_table, torch.uint8),
                                         ptr(dist), ptr(band), ptr(ws), wbytes, stream()),
          "dsu_volume_band_distance")
    return dist, band.view(torch.bool)


def mc_cube_index(volume, isovalue):
    """(X,Y,Z) f64 device volume -> (X-1,Y-1,Z-1) uint8 marching-cubes configuration bytes."""
    X, Y, Z = volume.shape
    cube = torch.empty((X - 1, Y - 1, Z - 1), dtype=torch.uint8, device=volume.device)
    check(lib().dsu_mc_cube_index(ptr(volume, torch.float64), X, Y, Z, float(isovalue), ptr(cube),
                                  stream()), "dsu_mc_cube_index")
    return cube


# ------------------------------------------------------------------ nerfacc replacements
def ray_aabb(rays_o, rays_d, aabb6, jitter=None, step=0.0):
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    n = rays_o.shape[0]
    t_min = torch.empty(n, dtype=torch.float32, device=rays_o.device)
    t_max = torch.empty_like(t_min)
    a = (C.c_float * 6)(*[float(v) for v in aabb6])
    check(lib().dsu_ray_aabb(ptr(rays_o), ptr(rays_d), n, a, ptr(jitter), float(step),
                             ptr(t_min), ptr(t_max), stream()), "dsu_ray_aabb")
    return t_min, t_max


def ray_march(rays_o, rays_d, t_min, t_max, aabb6, occ_binary, res, step):
    """Two-pass marching.  Returns ray_indices (int64), t_starts, t_ends (n,), and the per-ray
    (offsets, counts) int32 packing the compositing kernels consume."""
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    n = rays_o.shape[0]
    dev = rays_o.device
    a = (C.c_float * 6)(*[float(v) for v in aabb6])
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    occp = ptr(occ_binary, torch.uint8) if occ_binary is not None else None
    check(lib().dsu_ray_march_count(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), n, a, occp,
                                    int(res), float(step), ptr(counts), stream()),
          "dsu_ray_march_count")
    csum = torch.cumsum(counts, 0, dtype=torch.int32)
    offsets = (csum - counts).contiguous()
    total = int(csum[-1].item()) if n > 0 else 0
    ray_indices = torch.empty(total, dtype=torch.int64, device=dev)
    t_starts = torch.empty(total, dtype=torch.float32, device=dev)
    t_ends = torch.empty(total, dtype=torch.float32, device=dev)
    if total > 0:
        check(lib().dsu_ray_march_fill(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), n, a,
                                       occp, int(res), float(step), ptr(offsets),
                                       ptr(ray_indices), ptr(t_starts), ptr(t_ends), stream()),
              "dsu_ray_march_fill")
    return ray_indices, t_starts, t_ends, offsets, counts


# lazily built, process-wide; keyed by stream: drawings in flight on one GPU run on their own streams
_MARCH_SCRATCH = {}


def ray_march_single_pass(rays_o, rays_d, t_min, t_max, aabb6, occ_binary, res, step):
    """Same outputs as ray_march with ONE serial march (fixed-capacity scratch rows + compaction).
    The capacity covers the longest possible chord of the box; rays are re-marched with the
    two-pass path in the (never observed) case that a count exceeds it."""
    import math
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    n = rays_o.shape[0]
    dev = rays_o.device
    a = (C.c_float * 6)(*[float(v) for v in aabb6])
    diag = math.sqrt(sum((aabb6[3 + d] - aabb6[d]) ** 2 for d in range(3)))
    cap = int(diag / step) + 8
    key = (str(dev), cap, stream().value)          # scratch rows belong to ONE stream's launches
    sc = _MARCH_SCRATCH.get(key)
    if sc is None or sc[0].shape[0] < n * cap:
        rows = max(n, 8192)
        sc = (torch.empty(rows * cap, dtype=torch.float32, device=dev),
              torch.empty(rows * cap, dtype=torch.float32, device=dev))
        _MARCH_SCRATCH[key] = sc
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    occp = ptr(occ_binary, torch.uint8) if occ_binary is not None else None
    check(lib().dsu_ray_march_scratch(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), n, a, occp,
                                      int(res), float(step), cap, ptr(counts), ptr(sc[0]),
                                      ptr(sc[1]), stream()), "dsu_ray_march_scratch")
    csum = torch.cumsum(counts, 0, dtype=torch.int32)
    offsets = (csum - counts).contiguous()
    stats = torch.stack([csum[-1], counts.max()]).tolist() if n > 0 else [0, 0]   # one host sync
    total, cmax = int(stats[0]), int(stats[1])
    if cmax > cap:
        return ray_march(rays_o, rays_d, t_min, t_max, aabb6, occ_binary, res, step)
    ray_indices = torch.empty(total, dtype=torch.int64, device=dev)
    t_starts = torch.empty(total, dtype=torch.float32, device=dev)
    t_ends = torch.empty(total, dtype=torch.float32, device=dev)
    if total > 0:
        check(lib().dsu_ray_compact(ptr(sc[0]), ptr(sc[1]), cap, ptr(offsets), ptr(counts), n,
                                    ptr(ray_indices), ptr(t_starts), ptr(t_ends), stream()),
              "dsu_ray_compact")
    return ray_indices, t_starts, t_ends, offsets, counts


class MarchHandle:
    """State between ray_march_begin (march + offsets scan, all asynchronous) and
    ray_march_finish (needs the host-side total to size the packed outputs)."""
    __slots__ = ("rays_o", "rays_d", "counts", "offsets", "stats", "scratch", "cap", "n")


def ray_march_begin(rays_o, rays_d, t_min, t_max, aabb6, occ_binary, res, step):
    """Launch the single-pass march into the scratch rows and the offsets scan.  Nothing here
    waits for the device; `handle.stats` (int32[2] = total, max count) is read by the caller
    (directly, or through an asynchronous copy when the march is prefetched on a side stream)."""
    import math
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    n = rays_o.shape[0]
    dev = rays_o.device
    a = (C.c_float * 6)(*[float(v) for v in aabb6])
    diag = math.sqrt(sum((aabb6[3 + d] - aabb6[d]) ** 2 for d in range(3)))
    cap = int(diag / step) + 8
    key = (str(dev), cap, stream().value)
    sc = _MARCH_SCRATCH.get(key)
    if sc is None or sc[0].shape[0] < n * cap:
        rows = max(n, 8192)
        sc = (torch.empty(rows * cap, dtype=torch.float32, device=dev),
              torch.empty(rows * cap, dtype=torch.float32, device=dev))
        _MARCH_SCRATCH[key] = sc
    h = MarchHandle()
    h.rays_o, h.rays_d, h.scratch, h.cap, h.n = rays_o, rays_d, sc, cap, n
    h.counts = torch.empty(n, dtype=torch.int32, device=dev)
    h.offsets = torch.empty(n, dtype=torch.int32, device=dev)
    h.stats = torch.empty(2, dtype=torch.int32, device=dev)
    occp = ptr(occ_binary, torch.uint8) if occ_binary is not None else None
    check(lib().dsu_ray_march_scratch(ptr(rays_o), ptr(rays_d), ptr(t_min), ptr(t_max), n, a, occp,
                                      int(res), float(step), cap, ptr(h.counts), ptr(sc[0]),
                                      ptr(sc[1]), stream()), "dsu_ray_march_scratch")
    check(lib().dsu_ray_offsets(ptr(h.counts), n, ptr(h.offsets), ptr(h.stats), stream()),
          "dsu_ray_offsets")
    return h


def ray_march_finish(h, total, cmax, tail_rows=0):
    """Pack the scratch rows: returns (points (total + tail_rows, 3), t_starts, t_ends)."""
    if cmax > h.cap:
        raise DsuError(f"a ray produced {cmax} samples, above the scratch capacity {h.cap}")
    dev = h.rays_o.device
    points = torch.empty(total + tail_rows, 3, dtype=torch.float32, device=dev)
    t_starts = torch.empty(total, dtype=torch.float32, device=dev)
    t_ends = torch.empty(total, dtype=torch.float32, device=dev)
    if total > 0:
        check(lib().dsu_ray_compact_points(ptr(h.scratch[0]), ptr(h.scratch[1]), h.cap,
                                           ptr(h.offsets), ptr(h.counts), h.n, ptr(h.rays_o),
                                           ptr(h.rays_d), ptr(t_starts), ptr(t_ends), ptr(points),
                                           stream()), "dsu_ray_compact_points")
    return points, t_starts, t_ends


class PackedStepBuffers:
    """Fixed-capacity outputs of the prefetch path (one set per step parity): packed sample
    positions + tail rows, t_starts / t_ends, their Morton-sorted copy + permutation, sort scratch."""

    def __init__(self, capacity, tail_rows, bits, device):
        self.capacity, self.tail_rows, self.bits = int(capacity), int(tail_rows), int(bits)
        rows = self.capacity + self.tail_rows
        f32 = dict(dtype=torch.float32, device=device)
        self.points = torch.empty(rows, 3, **f32)
        self.t_starts = torch.empty(self.capacity, **f32)
        self.t_ends = torch.empty(self.capacity, **f32)
        self.sorted = torch.empty(rows, 3, **f32) if bits else None
        self.perm = torch.empty(rows, dtype=torch.int32, device=device) if bits else None
        self.ws_bytes = int(lib().dsu_spatial_sort_workspace_bytes(rows, self.bits)) if bits else 0
        self.ws = torch.empty(max(self.ws_bytes, 4) // 4, dtype=torch.int32, device=device)


def ray_pack_prefetched(h, bufs, pts_random, perturb, radius, alpha=1e-2):
    """On the CURRENT (side) stream, without knowing the sample total on the host: compaction of
    the march scratch into bufs.points / t_starts / t_ends, the random + perturbed points behind
    the samples, and the Morton sort of all of them (count read from h.stats on the device)."""
    n_r = pts_random.shape[0]
    assert 2 * n_r == bufs.tail_rows
    check(lib().dsu_ray_compact_points_cap(ptr(h.scratch[0]), ptr(h.scratch[1]), h.cap,
                                           ptr(h.offsets), ptr(h.counts), h.n, ptr(h.rays_o),
                                           ptr(h.rays_d), ptr(bufs.t_starts), ptr(bufs.t_ends),
                                           ptr(bufs.points), bufs.capacity, stream()),
          "dsu_ray_compact_points_cap")
    check(lib().dsu_points_tail(ptr(bufs.points), bufs.capacity + bufs.tail_rows, ptr(h.stats),
                                ptr(_f32c(pts_random)), ptr(_f32c(perturb)), n_r, float(alpha),
                                stream()), "dsu_points_tail")
    if bufs.bits:
        check(lib().dsu_spatial_sort_dev(ptr(bufs.points), bufs.capacity + bufs.tail_rows,
                                         ptr(h.stats), bufs.tail_rows, float(radius), bufs.bits,
                                         ptr(bufs.perm), ptr(bufs.sorted), ptr(bufs.ws),
                                         bufs.ws_bytes, stream()), "dsu_spatial_sort_dev")


def ray_march_points(rays_o, rays_d, t_min, t_max, aabb6, occ_binary, res, step, tail_rows=0):
    """Single-pass march for the fused optimisation step: returns
    (points (total + tail_rows, 3), t_starts, t_ends, offsets, counts, total) where
    points[:total] = rays_o[r] + rays_d[r] * (t_start + t_end) / 2 and the tail rows are left for
    the caller (random / perturbed points evaluated in the same geometry launch).  One host copy
    (total, max count); no ray_indices."""
    h = ray_march_begin(rays_o, rays_d, t_min, t_max, aabb6, occ_binary, res, step)
    total, cmax = h.stats.tolist()                                # the step's one host sync
    points, t_starts, t_ends = ray_march_finish(h, total, cmax, tail_rows)
    return points, t_starts, t_ends, h.offsets, h.counts, total


def _tex_struct(params):
    from ._lib import TexMlp
    for t, shape in zip(params, ((64, 16), (64,), (64, 64), (64,), (3, 64), (3,))):
        if tuple(t.shape) != shape:
            raise DsuError(f"texture MLP parameter of shape {tuple(t.shape)}, expected {shape}")
    return TexMlp(*[ptr(t, torch.float32).value for t in params])


def texture_fwd(params, tex_in):
    """params = [w0 (64,16), b0, w1 (64,64), b1, w2 (3,64), b2] f32 contiguous;
    returns rgb (n,3) = sigmoid(MLP(tex_in))."""
    tex_in = _f32c(tex_in)
    n = tex_in.shape[0]
    assert tex_in.shape[1] == 16
    rgb = torch.empty((n, 3), dtype=torch.float32, device=tex_in.device)
    m = _tex_struct(params)
    check(lib().dsu_texture_fwd(C.byref(m), ptr(tex_in), n, ptr(rgb), stream()), "dsu_texture_fwd")
    return rgb


def texture_bwd(params, tex_in, rgb, d_rgb):
    """Returns (d_tex_in (n,16), [g_w0, g_b0, g_w1, g_b1, g_w2, g_b2])."""
    tex_in, rgb, d_rgb = _f32c(tex_in), _f32c(rgb), _f32c(d_rgb)
    n = tex_in.shape[0]
    dev = tex_in.device
    d_in = torch.empty_like(tex_in)
    sizes = [t.numel() for t in params]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    g = [v.view_as(t) for v, t in zip(torch.split(flat, sizes), params)]
    wbytes = int(lib().dsu_texture_bwd_workspace_bytes(n))
    ws = torch.empty(max(wbytes, 4) // 4, dtype=torch.float32, device=dev)
    m = _tex_struct(params)
    check(lib().dsu_texture_bwd(C.byref(m), ptr(tex_in), ptr(rgb), ptr(d_rgb), n, ptr(d_in),
                                *[ptr(t) for t in g], ptr(ws), wbytes, stream()),
          "dsu_texture_bwd")
    return d_in, g


def texture_fwd_shaded(params, feature, grad, with_mask=False):
    """shade_prep_fwd + texture_fwd in one launch: returns (normal (n,3), rgb (n,3)) and, with_mask,
    the ReLU pattern of hidden layer 1 ((n,2) int32 words) for texture_bwd_shaded_partials(h1_mask=)."""
    feature, grad = _f32c(feature), _f32c(grad)
    n = feature.shape[0]
    normal = torch.empty((n, 3), dtype=torch.float32, device=feature.device)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=feature.device)
    m = _tex_struct(params)
    if with_mask:
        mask = torch.empty((n, 2), dtype=torch.int32, device=feature.device)
        check(lib().dsu_texture_fwd_shaded_m(C.byref(m), ptr(feature), ptr(grad), n, ptr(normal), ptr(rgb),
                                             ptr(mask), stream()), "dsu_texture_fwd_shaded_m")
        return normal, rgb, mask
    check(lib().dsu_texture_fwd_shaded(C.byref(m), ptr(feature), ptr(grad), n, ptr(normal), ptr(rgb),
                                       stream()), "dsu_texture_fwd_shaded")
    return normal, rgb


def texture_bwd_shaded(params, feature, grad, rgb, d_rgb, d_normal, tail_rows=0):
    """texture_bwd + shade_prep_bwd in one launch (+ the reduction): returns
    (d_grad (n,3), d_feature (n + tail_rows, 13) with zero tail rows, [g_w0 .. g_b2])."""
    feature, grad, rgb, d_rgb = _f32c(feature), _f32c(grad), _f32c(rgb), _f32c(d_rgb)
    n = rgb.shape[0]
    dev = rgb.device
    d_grad = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_feat = torch.empty((n + tail_rows, 13), dtype=torch.float32, device=dev)
    sizes = [t.numel() for t in params]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    g = [v.view_as(t) for v, t in zip(torch.split(flat, sizes), params)]
    wbytes = int(lib().dsu_texture_bwd_workspace_bytes(n))
    ws = torch.empty(max(wbytes, 4) // 4, dtype=torch.float32, device=dev)
    m = _tex_struct(params)
    dn = None if d_normal is None else _f32c(d_normal)
    check(lib().dsu_texture_bwd_shaded(C.byref(m), ptr(feature), ptr(grad), ptr(rgb), ptr(d_rgb),
                                       ptr(dn), n, int(tail_rows), ptr(d_grad), ptr(d_feat),
                                       *[ptr(t) for t in g], ptr(ws), wbytes, stream()),
          "dsu_texture_bwd_shaded")
    return d_grad, d_feat, g


def texture_partial_map():
    """Destination of every element of the texture backward's partial vectors inside a contiguous
    [w0 | b0 | w1 | b1 | w2 | b2] gradient block (-1: padding), host int32 array."""
    import numpy as np
    n = int(lib().dsu_texture_partial_map(None))
    m = np.empty(n, dtype=np.int32)
    lib().dsu_texture_partial_map(m.ctypes.data_as(C.c_void_p))
    return m


class DeferredSum:
    """A dsu_partial_reduce record plus the tensors it points into (kept alive with it)."""

    def __init__(self, record, keep):
        self.record, self.keep = record, keep


def texture_bwd_shaded_partials(params, feature, grad, rgb, d_rgb, d_normal, tail_rows=0, h1_mask=None):
    """texture_bwd_shaded WITHOUT the final sum: returns (d_grad, d_feature, g_flat, deferred) where
    g_flat (zeros now) is the contiguous gradient block the deferred sum will be added to by the
    launch it is handed to (sdf_fd_bwd(..., extra=deferred)).  h1_mask: texture_fwd_shaded(with_mask=True)'s
    pattern (the recompute of layer 1 then runs as bf16 x 3)."""
    from ._lib import PartialReduce
    feature, grad, rgb, d_rgb = _f32c(feature), _f32c(grad), _f32c(rgb), _f32c(d_rgb)
    n = rgb.shape[0]
    dev = rgb.device
    d_grad = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_feat = torch.empty((n + tail_rows, 13), dtype=torch.float32, device=dev)
    flat = torch.zeros(sum(t.numel() for t in params), dtype=torch.float32, device=dev)
    wbytes = int(lib().dsu_texture_bwd_workspace_bytes(n))
    ws = torch.empty(max(wbytes, 4) // 4, dtype=torch.float32, device=dev)
    m = _tex_struct(params)
    dn = None if d_normal is None else _f32c(d_normal)
    rec = PartialReduce()
    check(lib().dsu_texture_bwd_shaded_partials_m(C.byref(m), ptr(feature), ptr(grad), ptr(rgb),
                                                  ptr(d_rgb), ptr(dn), n, int(tail_rows), ptr(d_grad),
                                                  ptr(d_feat), ptr(h1_mask, torch.int32), ptr(ws), wbytes,
                                                  C.byref(rec), stream()),
          "dsu_texture_bwd_shaded_partials_m")
    tmap = torch.from_numpy(texture_partial_map()).to(dev)
    rec.map, rec.base = tmap.data_ptr(), flat.data_ptr()
    return d_grad, d_feat, flat, DeferredSum(rec, (ws, tmap, flat))


def ortho_ray_batch(index, x, y, c2w, origins, directions, images, normals, masks, view_weights):
    """Fused preprocess_data: returns the batch dict (rays (n,6), rgb, normal, mask, cosines,
    view_weights) for int64 (index, x, y) draws.  Dataset tensors (V,H,W,*) f32 contiguous."""
    n = index.shape[0]
    dev = images.device
    V, H, W, ch = images.shape
    f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    out = {"rays": f(n, 6), "rgb": f(n, ch), "normal": f(n, 3), "mask": f(n), "cosines": f(n),
           "view_weights": f(n)}
    i64 = torch.int64
    check(lib().dsu_ortho_ray_batch(ptr(index, i64), ptr(x, i64), ptr(y, i64), n, ptr(_f32c(c2w)),
                                    ptr(origins, torch.float32), ptr(directions, torch.float32),
                                    ptr(images, torch.float32), ch, ptr(normals, torch.float32),
                                    ptr(masks, torch.float32), ptr(view_weights, torch.float32), H,
                                    W, ptr(out["rays"]), ptr(out["rgb"]), ptr(out["normal"]),
                                    ptr(out["mask"]), ptr(out["cosines"]), ptr(out["view_weights"]),
                                    stream()), "dsu_ortho_ray_batch")
    return out


RAY_LOSS_MAX_RAYS = 8192      # DSU_RAY_LOSS_MAX_RAYS (include/dsu_hip.h)


def ray_losses(comp, rgb, normal, mask, cosines, view_weights, cfg):
    """Fused ray-level loss terms + d/d comp.  cfg: dict with rgb_p_ratio, normal_p_ratio,
    mask_p_ratio, lambda_rgb_mse, lambda_rgb_l1, lambda_normal, lambda_mask, geo_aware.
    Returns (terms (4,) = [rgb_mse, rgb_l1, normal, mask], d_comp (R,8))."""
    from ._lib import RayLossCfg
    comp = _f32c(comp)
    r = comp.shape[0]
    c = RayLossCfg(float(cfg["rgb_p_ratio"]), float(cfg["normal_p_ratio"]),
                   float(cfg["mask_p_ratio"]), float(cfg["lambda_rgb_mse"]),
                   float(cfg["lambda_rgb_l1"]), float(cfg["lambda_normal"]),
                   float(cfg["lambda_mask"]), int(bool(cfg["geo_aware"])), 0)
    terms = torch.empty(4, dtype=torch.float32, device=comp.device)
    d_comp = torch.empty_like(comp)
    check(lib().dsu_ray_losses(ptr(comp), ptr(_f32c(rgb)), ptr(_f32c(normal)), ptr(_f32c(mask)),
                               ptr(_f32c(cosines)), ptr(_f32c(view_weights)), r, C.byref(c),
                               ptr(terms), ptr(d_comp), stream()), "dsu_ray_losses")
    return terms, d_comp


def sample_losses(sdf_all, grad_all, n_samples, n_random, lambda_eikonal, lambda_sparsity,
                  sparsity_scale, lambda_smooth, d_sdf_all=None, d_grad_all=None):
    """Fused sample-level loss terms + gradients.  With d_sdf_all / d_grad_all given, their
    first n_samples rows already hold the compositing / shading gradients and the eikonal part
    is added; otherwise fresh tensors are returned.
    Returns (terms (3,) = [eikonal, sparsity, normal_smooth], d_sdf_all, d_grad_all)."""
    sdf_all, grad_all = _f32c(sdf_all), _f32c(grad_all)
    acc = d_sdf_all is not None
    if not acc:
        d_sdf_all = torch.empty_like(sdf_all)
        d_grad_all = torch.empty_like(grad_all)
    assert sdf_all.shape[0] == n_samples + 2 * n_random
    terms = torch.empty(3, dtype=torch.float32, device=sdf_all.device)
    check(lib().dsu_sample_losses(ptr(sdf_all), ptr(grad_all), n_samples, n_random,
                                  float(lambda_eikonal), float(lambda_sparsity),
                                  float(sparsity_scale), float(lambda_smooth), int(acc),
                                  ptr(d_sdf_all, torch.float32), ptr(d_grad_all, torch.float32),
                                  ptr(terms), stream()), "dsu_sample_losses")
    return terms, d_sdf_all, d_grad_all


def weights_from_alpha_fwd(alpha, offsets, counts):
    alpha = _f32c(alpha)
    w = torch.empty_like(alpha)
    check(lib().dsu_weights_from_alpha_fwd(ptr(alpha), ptr(offsets, torch.int32),
                                           ptr(counts, torch.int32), offsets.shape[0], ptr(w),
                                           stream()), "dsu_weights_from_alpha_fwd")
    return w


def weights_from_alpha_bwd(alpha, weights, d_weights, offsets, counts):
    d_alpha = torch.empty_like(alpha)
    check(lib().dsu_weights_from_alpha_bwd(ptr(_f32c(alpha)), ptr(_f32c(weights)),
                                           ptr(_f32c(d_weights)), ptr(offsets, torch.int32),
                                           ptr(counts, torch.int32), offsets.shape[0],
                                           ptr(d_alpha), stream()), "dsu_weights_from_alpha_bwd")
    return d_alpha


def accumulate_fwd(weights, values, offsets, counts):
    weights = _f32c(weights)
    n_rays = offsets.shape[0]
    ch = 1 if values is None else values.shape[-1]
    v = None if values is None else _f32c(values)
    out = torch.empty((n_rays, ch), dtype=torch.float32, device=weights.device)
    check(lib().dsu_accumulate_fwd(ptr(weights), ptr(v), ch, ptr(offsets, torch.int32),
                                   ptr(counts, torch.int32), n_rays, ptr(out), stream()),
          "dsu_accumulate_fwd")
    return out


def occgrid_ema(occs, idx, occ, decay):
    """occs[idx] = max(occs[idx] * decay, occ): old values gathered first; a cell listed several
    times is decayed once and keeps its largest candidate."""
    scratch = torch.empty(occ.shape[0], dtype=torch.float32, device=occs.device) \
        if idx is not None else None
    check(lib().dsu_occgrid_ema(ptr(occs, torch.float32), ptr(idx, torch.int64) if idx is not None
                                else None, ptr(_f32c(occ)), occ.shape[0], float(decay),
                                ptr(scratch) if scratch is not None else None, stream()),
          "dsu_occgrid_ema")


def occgrid_binarize(occs, thre):
    out = torch.empty(occs.shape[0], dtype=torch.uint8, device=occs.device)
    check(lib().dsu_occgrid_binarize(ptr(occs, torch.float32), occs.shape[0], float(thre),
                                     ptr(out), stream()), "dsu_occgrid_binarize")
    return out


# ------------------------------------------------------------------ style translator
ACT = {"none": 0, None: 0, "relu": 1, "leaky_relu": 2, "tanh": 3}


def ric_offsets(H, W, device):
    out = torch.empty((18, H, W), dtype=torch.float32, device=device)
    check(lib().dsu_ric_offsets(H, W, ptr(out), stream()), "dsu_ric_offsets")
    return out


def deform_conv3x3(x, offset, weight, ep_scale=None, ep_shift=None, act=None, residual=None,
                   in_relu=False):
    """offset: (18,H,W) shared by the batch, or (B,18,H,W)."""
    x, weight, offset = _f32c(x), _f32c(weight), _f32c(offset)
    B, Cin, H, W = x.shape
    O = weight.shape[0]
    assert weight.shape[1:] == (Cin, 3, 3), "dsu deform conv: 3x3, groups=1 only"
    bstride = 0 if offset.dim() == 3 or offset.shape[0] == 1 else 18 * H * W
    out = torch.empty((B, O, H, W), dtype=torch.float32, device=x.device)
    check(lib().dsu_deform_conv3x3_fwd(ptr(x), ptr(offset), bstride, ptr(weight), B, Cin, H, W, O,
                                       int(in_relu), ptr(ep_scale), ptr(ep_shift), ACT[act],
                                       ptr(residual), ptr(out), stream()),
          "dsu_deform_conv3x3_fwd")
    return out


def conv2d(x, weight, bias=None, stride=1, padding=0, ep_scale=None, ep_shift=None, act=None,
           residual=None, in_relu=False):
    x, weight = _f32c(x), _f32c(weight)
    B, Cin, H, W = x.shape
    O, Cw, k, k2 = weight.shape
    assert Cw == Cin and k == k2
    OH = (H + 2 * padding - k) // stride + 1
    OW = (W + 2 * padding - k) // stride + 1
    out = torch.empty((B, O, OH, OW), dtype=torch.float32, device=x.device)
    check(lib().dsu_conv2d_fwd(ptr(x), ptr(weight), ptr(bias), B, Cin, H, W, O, k, stride,
                               padding, int(in_relu), ptr(ep_scale), ptr(ep_shift), ACT[act],
                               ptr(residual),
                               ptr(out), stream()), "dsu_conv2d_fwd")
    return out


class PackedConvWeight:
    """A convolution weight in the layout the evaluation kernels read, built on the device once per
    weight version: split into bf16 hi/mid parts (dsu_conv_x3_pack_weights; bf16 x 3 kernels) or,
    with exact=True, as f32 (dsu_conv_f32p_pack_weights; exact-f32 kernels)."""

    def __init__(self, weight, exact=False):
        weight = _f32c(weight.detach())
        self.O, self.C, self.k, k2 = weight.shape
        assert self.k == k2
        self.exact = bool(exact)
        self.C8 = (self.C + 7) & ~7      # the kernels read channels in groups of eight
        n = int(lib().dsu_conv_x3_packed_elems(self.O, self.C, self.k))
        if n <= 0:
            raise DsuError("dsu_conv_x3_packed_elems: unsupported shape")
        if self.exact:
            self.f32 = torch.empty(n, dtype=torch.float32, device=weight.device)
            check(lib().dsu_conv_f32p_pack_weights(ptr(weight), self.O, self.C, self.k,
                                                   ptr(self.f32), stream()),
                  "dsu_conv_f32p_pack_weights")
            return
        self.hi = torch.empty(n, dtype=torch.int16, device=weight.device)
        self.mid = torch.empty(n, dtype=torch.int16, device=weight.device)
        check(lib().dsu_conv_x3_pack_weights(ptr(weight), self.O, self.C, self.k, ptr(self.hi),
                                             ptr(self.mid), stream()), "dsu_conv_x3_pack_weights")


def cat_channels8(tensors):
    """torch.cat(tensors, 1) with zero channels appended up to a multiple of eight (what the
    bf16 x 3 kernels read; the packed weights of those channels are zero)."""
    c = sum(t.shape[1] for t in tensors)
    if c % 8:
        t0 = tensors[0]
        z = torch.zeros((t0.shape[0], 8 - c % 8) + tuple(t0.shape[2:]), dtype=t0.dtype,
                        device=t0.device)
        tensors = tuple(tensors) + (z,)
    return torch.cat(tensors, 1)


def _channels8(x, packed):
    if x.shape[1] == packed.C8:
        return x
    assert x.shape[1] == packed.C, (x.shape, packed.C)
    return cat_channels8((x,))


def deform_conv3x3_x3(x, offset, packed, ep_scale=None, ep_shift=None, act=None, residual=None,
                      in_relu=False):
    """deform_conv3x3 with a PackedConvWeight: bf16 x 3 products with f32 accumulation, or — for a
    weight packed with exact=True — exact f32 products (dsu_deform_conv3x3_fwd_f32p)."""
    x, offset = _channels8(_f32c(x), packed), _f32c(offset)
    B, Cin, H, W = x.shape
    assert packed.k == 3, "dsu deform conv: 3x3, groups=1 only"
    bstride = 0 if offset.dim() == 3 or offset.shape[0] == 1 else 18 * H * W
    out = torch.empty((B, packed.O, H, W), dtype=torch.float32, device=x.device)
    if packed.exact:
        check(lib().dsu_deform_conv3x3_fwd_f32p(ptr(x), ptr(offset), bstride, ptr(packed.f32), B, Cin,
                                                H, W, packed.O, int(in_relu), ptr(ep_scale),
                                                ptr(ep_shift), ACT[act], ptr(residual), ptr(out),
                                                stream()), "dsu_deform_conv3x3_fwd_f32p")
        return out
    check(lib().dsu_deform_conv3x3_fwd_x3(ptr(x), ptr(offset), bstride, ptr(packed.hi),
                                          ptr(packed.mid), B, Cin, H, W, packed.O, int(in_relu),
                                          ptr(ep_scale), ptr(ep_shift), ACT[act], ptr(residual),
                                          ptr(out), stream()), "dsu_deform_conv3x3_fwd_x3")
    return out


def conv2d_x3(x, packed, bias=None, stride=1, padding=0, ep_scale=None, ep_shift=None, act=None,
              residual=None, in_relu=False):
    """conv2d with a PackedConvWeight: bf16 x 3 products with f32 accumulation, or — for a weight
    packed with exact=True — exact f32 products (dsu_conv2d_fwd_f32p; k in {1, 3})."""
    x = _channels8(_f32c(x), packed)
    B, Cin, H, W = x.shape
    k = packed.k
    OH = (H + 2 * padding - k) // stride + 1
    OW = (W + 2 * padding - k) // stride + 1
    out = torch.empty((B, packed.O, OH, OW), dtype=torch.float32, device=x.device)
    if packed.exact:
        check(lib().dsu_conv2d_fwd_f32p(ptr(x), ptr(packed.f32), ptr(bias), B, Cin, H, W, packed.O,
                                        k, stride, padding, int(in_relu), ptr(ep_scale),
                                        ptr(ep_shift), ACT[act], ptr(residual), ptr(out), stream()),
              "dsu_conv2d_fwd_f32p")
        return out
    check(lib().dsu_conv2d_fwd_x3(ptr(x), ptr(packed.hi), ptr(packed.mid), ptr(bias), B, Cin, H, W,
                                  packed.O, k, stride, padding, int(in_relu), ptr(ep_scale),
                                  ptr(ep_shift), ACT[act], ptr(residual), ptr(out), stream()),
          "dsu_conv2d_fwd_x3")
    return out


# ------------------------------------------------------------------ style translator: training
class DeformPlan:
    """Everything the fixed-offset deformable convolution needs besides the weights, for one
    (18,H,W) offset map: the per-(pixel, tap) sampling table (dsu_deform_tap_table) and its
    transpose as CSR over input pixels (built once on the host from that table)."""

    def __init__(self, offset):
        offset = _f32c(offset)
        assert offset.dim() == 3 and offset.shape[0] == 18
        self.offset = offset
        _, H, W = offset.shape
        self.H, self.W, self.npix = H, W, H * W
        nbytes = int(lib().dsu_deform_tap_table_bytes(H, W))
        self.table = torch.empty(nbytes, dtype=torch.uint8, device=offset.device)
        check(lib().dsu_deform_tap_table(ptr(offset), H, W, ptr(self.table), stream()),
              "dsu_deform_tap_table")
        rowptr, src, wgt = transpose_tap_table(self.table.cpu().numpy(), self.npix)
        dev = offset.device
        self.rowptr = torch.from_numpy(rowptr).to(dev)
        self.src = torch.from_numpy(src).to(dev)
        self.wgt = torch.from_numpy(wgt).to(dev)


def transpose_tap_table(table_bytes, npix):
    """(npix*9 records of 4 int32 corner offsets + 4 f32 weights) -> CSR over input pixels:
    rowptr (npix+1) int32, src = tap*npix + output_pixel int32, wgt f32.  Zero-weight corners
    (outside the image) are dropped; entries of a row keep (pixel, tap, corner) order."""
    import numpy as np
    rec = np.frombuffer(table_bytes, dtype=np.int32).reshape(npix * 9, 8)
    idx = rec[:, :4].reshape(-1)                                  # (npix*9*4,)
    w = rec[:, 4:].copy().view(np.float32).reshape(-1)
    pt = np.repeat(np.arange(npix * 9, dtype=np.int64), 4)        # record index = pix*9 + tap
    pix, tap = pt // 9, pt % 9
    keep = w != 0.0
    idx, w, srcv = idx[keep].astype(np.int64), w[keep], (tap * npix + pix)[keep]
    order = np.argsort(idx, kind="stable")
    counts = np.bincount(idx, minlength=npix)
    rowptr = np.zeros(npix + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(counts)
    return rowptr, srcv[order].astype(np.int32), w[order].astype(np.float32)


_DEFORM_PLANS = {}


def deform_plan(offset):
    """Cached DeformPlan for an offset map (keyed by storage, shape and device)."""
    key = (offset.data_ptr(), tuple(offset.shape), str(offset.device))
    plan = _DEFORM_PLANS.get(key)
    if plan is None:
        plan = DeformPlan(offset)
        _lib.publish_sync(offset.device)
        _DEFORM_PLANS[key] = plan
    return plan


def conv2d_wgrad(x, dout, k, stride=1, padding=0, plan=None, out=None, accumulate=False):
    """dW of nn.Conv2d, or of the fixed-offset deformable convolution when plan is given."""
    x, dout = _f32c(x), _f32c(dout)
    B, Cin, H, W = x.shape
    O, OH, OW = dout.shape[1], dout.shape[2], dout.shape[3]
    nbytes = int(lib().dsu_conv2d_wgrad_workspace_bytes(1 if plan is not None else 0, B, Cin, O,
                                                        OH, OW, k))
    if nbytes <= 0:
        raise DsuError("dsu_conv2d_wgrad_workspace_bytes: unsupported shape")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    dw = out if out is not None else torch.empty((O, Cin, k, k), dtype=torch.float32,
                                                 device=x.device)
    check(lib().dsu_conv2d_wgrad(ptr(x), ptr(dout), ptr(plan.table) if plan is not None else None,
                                 B, Cin, H, W, O, k, stride, padding, ptr(ws), ptr(dw),
                                 int(accumulate), stream()), "dsu_conv2d_wgrad")
    return dw


def conv2d_dgrad(dout, weight, in_hw, stride=1, padding=0):
    """dX of nn.Conv2d: the forward kernel on (zero-dilated, for stride > 1) dout with the
    spatially flipped, channel-transposed weight and padding k-1-p."""
    O, Cin, k, _ = weight.shape
    H, W = in_hw
    wt = weight.flip(2, 3).transpose(0, 1).contiguous()
    dout = _f32c(dout)
    if stride != 1:
        B, _, OH, OW = dout.shape
        d = torch.zeros((B, O, H - k + 1 + 2 * padding, W - k + 1 + 2 * padding),
                        dtype=torch.float32, device=dout.device)
        d[:, :, ::stride, ::stride][:, :, :OH, :OW] = dout
        dout = d
    dx = conv2d(dout, wt, None, 1, k - 1 - padding)
    assert tuple(dx.shape[2:]) == (H, W), (dx.shape, in_hw)
    return dx


def deform_conv3x3_dgrad(dout, weight, plan):
    """dX of the fixed-offset deformable convolution: dcol = W^T dout (1x1 convolution), then
    the transposed sampling operator as a gather."""
    dout = _f32c(dout)
    B, O, H, W = dout.shape
    Cin = weight.shape[1]
    wt = weight.permute(1, 2, 3, 0).reshape(Cin * 9, O, 1, 1).contiguous()
    dcol = conv2d(dout, wt, None, 1, 0)
    dx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dout.device)
    check(lib().dsu_deform_conv3x3_dgrad_gather(ptr(dcol), ptr(plan.rowptr), ptr(plan.src),
                                                ptr(plan.wgt), B * Cin, H * W, ptr(dx), stream()),
          "dsu_deform_conv3x3_dgrad_gather")
    return dx


def _norm_cfg(x, instance, act, eps, momentum, stat_updates):
    B, Cn = x.shape[0], x.shape[1]
    return NormCfg(B, Cn, x[0, 0].numel(), int(instance), ACT[act], int(stat_updates), float(eps),
                   float(momentum))


def norm_train_fwd(x, gamma=None, beta=None, running_mean=None, running_var=None, instance=False,
                   act=None, eps=1e-5, momentum=0.1, stat_updates=1):
    """BatchNorm2d (training) / InstanceNorm2d + activation.  Returns y, save_mean, save_invstd."""
    x = _f32c(x)
    cfg = _norm_cfg(x, instance, act, eps, momentum, stat_updates)
    groups = x.shape[0] * x.shape[1] if instance else x.shape[1]
    y = torch.empty_like(x)
    stats = torch.empty((2, groups), dtype=torch.float32, device=x.device)
    check(lib().dsu_norm_train_fwd(C.byref(cfg), ptr(x), ptr(gamma), ptr(beta), ptr(running_mean),
                                   ptr(running_var), ptr(y), ptr(stats[0]), ptr(stats[1]),
                                   stream()), "dsu_norm_train_fwd")
    return y, stats[0], stats[1]


def norm_train_bwd(x, y, dy, gamma, save_mean, save_invstd, instance=False, act=None,
                   affine_grads=True):
    x, y, dy = _f32c(x), _f32c(y), _f32c(dy)
    cfg = _norm_cfg(x, instance, act, 0.0, 0.0, 0)
    dx = torch.empty_like(x)
    dgb = None
    if affine_grads and not instance:
        dgb = torch.empty((2, x.shape[1]), dtype=torch.float32, device=x.device)
    check(lib().dsu_norm_train_bwd(C.byref(cfg), ptr(x), ptr(y), ptr(dy), ptr(gamma),
                                   ptr(save_mean), ptr(save_invstd), ptr(dx),
                                   ptr(dgb[0]) if dgb is not None else None,
                                   ptr(dgb[1]) if dgb is not None else None, stream()),
          "dsu_norm_train_bwd")
    return (dx, dgb[0], dgb[1]) if dgb is not None else (dx, None, None)


def channel_sum(x):
    x = _f32c(x)
    B, Cn = x.shape[0], x.shape[1]
    out = torch.empty(Cn, dtype=torch.float32, device=x.device)
    check(lib().dsu_channel_sum(ptr(x), B, Cn, x[0, 0].numel(), ptr(out), stream()),
          "dsu_channel_sum")
    return out


def act_fwd(x, act):
    x = _f32c(x)
    y = torch.empty_like(x)
    check(lib().dsu_act_fwd(ptr(x), ptr(y), x.numel(), ACT[act], stream()), "dsu_act_fwd")
    return y


def act_bwd(dy, y, act):
    dy, y = _f32c(dy), _f32c(y)
    dx = torch.empty_like(y)
    check(lib().dsu_act_bwd(ptr(dy), ptr(y), ptr(dx), y.numel(), ACT[act], stream()), "dsu_act_bwd")
    return dx


def maxpool2_fwd(x):
    x = _f32c(x)
    B, Cn, H, W = x.shape
    y = torch.empty((B, Cn, H // 2, W // 2), dtype=torch.float32, device=x.device)
    check(lib().dsu_maxpool2_fwd(ptr(x), ptr(y), B * Cn, H, W, stream()), "dsu_maxpool2_fwd")
    return y


def maxpool2_bwd(x, dy):
    x, dy = _f32c(x), _f32c(dy)
    B, Cn, H, W = x.shape
    dx = torch.empty_like(x)
    check(lib().dsu_maxpool2_bwd(ptr(x), ptr(dy), ptr(dx), B * Cn, H, W, stream()),
          "dsu_maxpool2_bwd")
    return dx


def upsample2_fwd(x):
    x = _f32c(x)
    B, Cn, H, W = x.shape
    y = torch.empty((B, Cn, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    check(lib().dsu_upsample2_fwd(ptr(x), ptr(y), B * Cn, H, W, stream()), "dsu_upsample2_fwd")
    return y


def upsample2_bwd(dy):
    dy = _f32c(dy)
    B, Cn, H2, W2 = dy.shape
    dx = torch.empty((B, Cn, H2 // 2, W2 // 2), dtype=torch.float32, device=dy.device)
    check(lib().dsu_upsample2_bwd(ptr(dy), ptr(dx), B * Cn, H2 // 2, W2 // 2, stream()),
          "dsu_upsample2_bwd")
    return dx


def pair_loss(x, target, kind, grad_scale=None):
    """kind 'l1' | 'mse'; target: tensor like x or a float.  Returns (sum of |d| or d^2 as a
    0-dim tensor, grad or None) with grad = d(grad_scale * sum)/dx."""
    x = _f32c(x)
    t = _f32c(target) if torch.is_tensor(target) else None
    tconst = 0.0 if t is not None else float(target)
    partial = torch.empty(256, dtype=torch.float32, device=x.device)
    grad = torch.empty_like(x) if grad_scale is not None else None
    check(lib().dsu_pair_loss(ptr(x), ptr(t), tconst, x.numel(), {"l1": 0, "mse": 1}[kind],
                              float(grad_scale or 0.0), ptr(grad), ptr(partial), stream()),
          "dsu_pair_loss")
    return partial.sum(), grad


# ------------------------------------------------------------------ multi-view attention
def _i64x3(a, b, c):
    return (C.c_int64 * 3)(int(a), int(b), int(c))


def mv_attention(q, k, vt, seg_batch, heads, seg_len, scale=None):
    """q (Bq,Nq,H*d) f16; k (Bk,Nk,H*d) f16; vt (Bk,H*d,Nk) f16 (V transposed);
    seg_batch (Bq,S) int32: batch index of each seg_len-token K/V segment of query batch b.
    Returns (Bq,Nq,H*d) f16."""
    Bq, Nq, Cq = q.shape
    d = Cq // heads
    S = seg_batch.shape[1]
    assert q.dtype == torch.float16 and k.dtype == torch.float16 and vt.dtype == torch.float16
    # unit stride on the innermost axis (a size-1 axis keeps whatever stride its view had)
    assert q.stride(2) == 1 and k.stride(2) == 1 and (vt.stride(2) == 1 or vt.shape[2] == 1)
    out = torch.empty((Bq, Nq, Cq), dtype=torch.float16, device=q.device)
    if scale is None:
        scale = d ** -0.5
    check(lib().dsu_mv_attention_fwd(
        C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(vt.data_ptr()), ptr(out),
        ptr(seg_batch, torch.int32), Bq, heads, Nq, d, S, int(seg_len),
        _i64x3(q.stride(0), q.stride(1), d), _i64x3(k.stride(0), k.stride(1), d),
        _i64x3(vt.stride(0), d * vt.stride(1), vt.stride(1)),
        _i64x3(out.stride(0), out.stride(1), d), float(scale), stream()), "dsu_mv_attention_fwd")
    return out


# ------------------------------------------------------------------ f16 NHWC convolution
def conv_weight_okc(weight):
    """(O,C,k,k) nn.Conv2d weight -> (O,k*k,C) f16 contiguous (K ordered tap-major)."""
    O, Cin, k, _ = weight.shape
    return weight.detach().permute(0, 2, 3, 1).reshape(O, k * k, Cin).to(torch.float16).contiguous()


_TILE_COUNTERS = {}
_N_TILE_COUNTERS = 1 << 16
import os as _os
SPLITK_FIXUP = _os.environ.get("DSU_SPLITK_FIXUP", "0") == "1"   # in-kernel split-K fix-up instead of the separate reduce launch (A/B)


def _tile_counters(device):
    """Zeroed, persistent per (device, stream) int32 buffer for the in-kernel split-K fix-up (the
    kernel resets what it touches; launches that share a buffer must share a stream).  Only when
    SPLITK_FIXUP is set (tests / tools; off in the product): the device-scope release / acquire
    around the ticket writes back and invalidates the XCD's L2 per workgroup — the UNet forward
    took 26.6 ms with it against 12.6 ms with the separate reduce launches
    (profiles/round3_unet_splitk_fixup.txt)."""
    if not SPLITK_FIXUP:
        return None
    key = (str(device), stream().value)
    t = _TILE_COUNTERS.get(key)
    if t is None:
        t = _TILE_COUNTERS[key] = torch.zeros(_N_TILE_COUNTERS, dtype=torch.int32, device=device)
    return t


def conv2d_nhwc_f16(x_nhwc, w_okc, bias=None, k=3, stride=1, pad=1, upsample2x=False, addvec=None,
                    residual=None, split_k=None):
    """x_nhwc (B,H,W,C) f16 contiguous -> (B,OH,OW,O) f16.  split_k None = the library's choice
    (1 unless the output tiles alone leave most CUs idle), or an explicit factor."""
    B, H, W, Cin = x_nhwc.shape
    O = w_okc.shape[0]
    assert w_okc.shape[1] == k * k and w_okc.shape[2] == Cin
    IH, IW = (2 * H, 2 * W) if upsample2x else (H, W)
    OH, OW = (IH + 2 * pad - k) // stride + 1, (IW + 2 * pad - k) // stride + 1
    out = torch.empty((B, OH, OW, O), dtype=torch.float16, device=x_nhwc.device)
    f16 = torch.float16
    up = int(upsample2x)
    if split_k is None:
        split_k = int(lib().dsu_conv2d_nhwc_f16_split_k(B, H, W, Cin, O, k, stride, pad, up))
    ws, wbytes = None, 0
    if split_k > 1:
        wbytes = int(lib().dsu_conv2d_nhwc_f16_workspace_bytes(B, H, W, O, k, stride, pad, up,
                                                               split_k))
        ws = torch.empty(wbytes // 4, dtype=torch.float32, device=x_nhwc.device)
    cnt = _tile_counters(x_nhwc.device) if split_k > 1 else None
    check(lib().dsu_conv2d_nhwc_f16_fwd_fx(ptr(x_nhwc, f16), ptr(w_okc, f16), ptr(bias, f16), B, H,
                                           W, Cin, O, k, stride, pad, up, ptr(addvec, f16),
                                           ptr(residual, f16), ptr(out), int(split_k), ptr(ws),
                                           wbytes, ptr(cnt), _N_TILE_COUNTERS if cnt is not None else 0,
                                           stream()), "dsu_conv2d_nhwc_f16_fwd")
    return out


def linear_f16(x, weight, bias=None, residual=None, transposed_tokens=0, split_k=None):
    """nn.Linear on the hand-written MFMA kernel.  x (..., K) f16 contiguous, weight (N, K) f16,
    residual (..., N) added last.  transposed_tokens = T > 0: x is (B, T, K) and the result is
    (B, N, T) (V^T for mv_attention).  Returns (..., N) f16."""
    K = x.shape[-1]
    N = weight.shape[0]
    M = x.numel() // K
    f16 = torch.float16
    assert x.dtype == f16 and weight.dtype == f16 and x.is_contiguous() and weight.is_contiguous()
    if transposed_tokens:
        out = torch.empty((M // transposed_tokens, N, transposed_tokens), dtype=f16, device=x.device)
        split_k = 1
    else:
        out = torch.empty(x.shape[:-1] + (N,), dtype=f16, device=x.device)
    if split_k is None:
        split_k = int(lib().dsu_gemm_f16_split_k(M, K, N))
    ws, wbytes = None, 0
    if split_k > 1:
        wbytes = int(lib().dsu_gemm_f16_workspace_bytes(M, N, split_k))
        ws = torch.empty(wbytes // 4, dtype=torch.float32, device=x.device)
    if residual is not None:
        assert residual.is_contiguous() and residual.numel() == M * N
    cnt = _tile_counters(x.device) if split_k > 1 else None
    check(lib().dsu_gemm_f16_fwd_fx(ptr(x, f16), ptr(weight, f16), ptr(bias, f16), M, K, N,
                                    ptr(residual, f16), ptr(out), int(transposed_tokens), int(split_k),
                                    ptr(ws), wbytes, ptr(cnt),
                                    _N_TILE_COUNTERS if cnt is not None else 0, stream()),
          "dsu_gemm_f16_fwd")
    return out


def linear_geglu_f16(x, weight, bias=None):
    """FeedForward's first layer + GEGLU: x (..., K), weight (2N, K), bias (2N) -> (..., N)."""
    K = x.shape[-1]
    N = weight.shape[0] // 2
    M = x.numel() // K
    f16 = torch.float16
    assert x.dtype == f16 and weight.dtype == f16 and x.is_contiguous() and weight.is_contiguous()
    out = torch.empty(x.shape[:-1] + (N,), dtype=f16, device=x.device)
    check(lib().dsu_gemm_geglu_fwd(ptr(x, f16), ptr(weight, f16), ptr(bias, f16), M, K, N, ptr(out),
                                   stream()), "dsu_gemm_geglu_fwd")
    return out


# ------------------------------------------------------------------ norms / activations (f16)
def groupnorm_nhwc_f16(x, gamma, beta, groups, eps=1e-5, silu=False):
    """x (B,H,W,C) or (B,HW,C) f16 contiguous."""
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    out = torch.empty_like(x)
    ws = torch.empty(B * groups * 2, dtype=torch.float32, device=x.device)
    f16 = torch.float16
    check(lib().dsu_groupnorm_nhwc_f16(ptr(x, f16), ptr(gamma, f16), ptr(beta, f16), B, HW, Cc,
                                       groups, float(eps), int(silu), ptr(ws), ptr(out), stream()),
          "dsu_groupnorm_nhwc_f16")
    return out


def layernorm_f16(x, gamma, beta, eps=1e-5):
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    out = torch.empty_like(x)
    f16 = torch.float16
    check(lib().dsu_layernorm_f16(ptr(x, f16), ptr(gamma, f16), ptr(beta, f16), rows, Cc,
                                  float(eps), ptr(out), stream()), "dsu_layernorm_f16")
    return out


def geglu_f16(h):
    D = h.shape[-1] // 2
    rows = h.numel() // (2 * D)
    out = torch.empty(h.shape[:-1] + (D,), dtype=torch.float16, device=h.device)
    check(lib().dsu_geglu_f16(ptr(h, torch.float16), rows, D, ptr(out), stream()), "dsu_geglu_f16")
    return out


# ------------------------------------------------------------------ fused NeuS shading / compositing
def shade_prep_fwd(grad, feature):
    grad, feature = _f32c(grad), _f32c(feature)
    n = grad.shape[0]
    normal = torch.empty((n, 3), dtype=torch.float32, device=grad.device)
    tex_in = torch.empty((n, 16), dtype=torch.float32, device=grad.device)
    check(lib().dsu_shade_prep_fwd(ptr(grad), ptr(feature), n, ptr(normal), ptr(tex_in), stream()),
          "dsu_shade_prep_fwd")
    return normal, tex_in


def shade_prep_bwd(grad, d_normal, d_tex_in, out=None):
    """out = (d_grad (n,3), d_feat (n,13)) preallocated contiguous f32 (e.g. row prefixes of the
    buffers the geometry backward consumes)."""
    grad = _f32c(grad)
    n = grad.shape[0]
    if out is None:
        d_grad = torch.empty((n, 3), dtype=torch.float32, device=grad.device)
        d_feat = torch.empty((n, 13), dtype=torch.float32, device=grad.device)
    else:
        d_grad, d_feat = out
        assert d_grad.shape == (n, 3) and d_feat.shape == (n, 13)
    dn = None if d_normal is None else _f32c(d_normal)
    check(lib().dsu_shade_prep_bwd(ptr(grad), ptr(dn), ptr(_f32c(d_tex_in)), n, ptr(d_grad),
                                   ptr(d_feat), stream()), "dsu_shade_prep_bwd")
    return d_grad, d_feat


def neus_composite_fwd(sdf, normal, rgb, rays_d, t_starts, t_ends, offsets, counts, inv_s, car):
    n, n_rays = sdf.shape[0], rays_d.shape[0]
    dev = sdf.device
    alpha = torch.empty(n, dtype=torch.float32, device=dev)
    w = torch.empty(n, dtype=torch.float32, device=dev)
    comp = torch.empty((n_rays, 8), dtype=torch.float32, device=dev)
    check(lib().dsu_neus_composite_fwd(ptr(_f32c(sdf)), ptr(_f32c(normal)), ptr(_f32c(rgb)),
                                       ptr(_f32c(rays_d)), ptr(_f32c(t_starts)), ptr(_f32c(t_ends)),
                                       ptr(offsets, torch.int32), ptr(counts, torch.int32), n_rays,
                                       ptr(inv_s, torch.float32), float(car), ptr(alpha), ptr(w),
                                       ptr(comp), stream()), "dsu_neus_composite_fwd")
    return comp, alpha, w


def neus_composite_bwd(sdf, normal, rgb, rays_d, t_starts, t_ends, offsets, counts, inv_s, car,
                       alpha, weights, d_comp, d_weights=None, d_sdf_out=None):
    n, n_rays = sdf.shape[0], rays_d.shape[0]
    dev = sdf.device
    d_sdf = torch.empty(n, dtype=torch.float32, device=dev) if d_sdf_out is None else d_sdf_out
    assert d_sdf.shape == (n,)
    d_normal = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    d_inv = torch.zeros(1, dtype=torch.float32, device=dev)
    dw = None if d_weights is None else _f32c(d_weights)
    check(lib().dsu_neus_composite_bwd(ptr(_f32c(sdf)), ptr(_f32c(normal)), ptr(_f32c(rgb)),
                                       ptr(_f32c(rays_d)), ptr(_f32c(t_starts)), ptr(_f32c(t_ends)),
                                       ptr(offsets, torch.int32), ptr(counts, torch.int32), n_rays,
                                       ptr(inv_s, torch.float32), float(car), ptr(alpha),
                                       ptr(weights), ptr(_f32c(d_comp)), ptr(dw), ptr(d_sdf),
                                       ptr(d_normal), ptr(d_rgb), ptr(d_inv), stream()),
          "dsu_neus_composite_bwd")
    return d_sdf, d_normal, d_rgb, d_inv


# ------------------------------------------------------------------ export: mesh post-processing
class ZGrid:
    """Uniform xy grid over a triangle soup (tris (F,3,3) f32 on the device) for z-parallel ray
    queries, or over a point set (xy (n,2)) for the k-nearest-neighbour search: counting sort by
    cell (csrc/mesh_post.hip)."""

    def __init__(self, data, lo, hi, points=False, cells_per_axis=None):
        import math
        dev = data.device
        n = data.shape[0]
        g = cells_per_axis or int(min(1024, max(8, math.sqrt(max(n, 1) / 2.0))))
        span = max(float(hi[0] - lo[0]), float(hi[1] - lo[1]), 1e-9)
        self.g, self.cell = g, span / g * (1.0 + 1e-6)
        self.x0, self.y0 = float(lo[0]), float(lo[1])
        count_fn = lib().dsu_point_bin_count if points else lib().dsu_zgrid_count
        fill_fn = lib().dsu_point_bin_fill if points else lib().dsu_zgrid_fill
        counts = torch.zeros(g * g, dtype=torch.int32, device=dev)
        check(count_fn(ptr(data, torch.float32), n, self.x0, self.y0, self.cell, g, ptr(counts),
                       stream()), "grid count")
        self.offsets = torch.zeros(g * g + 1, dtype=torch.int32, device=dev)
        self.offsets[1:] = torch.cumsum(counts, 0)
        total = int(self.offsets[-1])
        self.items = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        cursor = torch.zeros(g * g, dtype=torch.int32, device=dev)
        check(fill_fn(ptr(data, torch.float32), n, self.x0, self.y0, self.cell, g, ptr(self.offsets),
                      ptr(cursor), ptr(self.items), stream()), "grid fill")
        self.data = data

    def args(self):
        return (self.x0, self.y0, self.cell, self.g, ptr(self.offsets), ptr(self.items))


def zray_cast(grid, faces_i32, origins, sign, self_vertex=None):
    """mesh_raycast.raycast(origin, (0,0,sign), mesh) for all origins (n,3) f32: returns
    (hit_count i32, t_near, face_near i32, t_far, face_far i32)."""
    origins = _f32c(origins)
    n = origins.shape[0]
    dev = origins.device
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    fn, ff = torch.empty_like(cnt), torch.empty_like(cnt)
    tn, tf = torch.empty(n, device=dev), torch.empty(n, device=dev)
    x0, y0, cell, g, off, items = grid.args()
    check(lib().dsu_zray_cast(ptr(grid.data, torch.float32), ptr(faces_i32, torch.int32),
                              grid.data.shape[0], x0, y0, cell, g, off, items, ptr(origins), n,
                              int(sign), ptr(self_vertex, torch.int32), ptr(cnt), ptr(tn), ptr(fn),
                              ptr(tf), ptr(ff), stream()), "dsu_zray_cast")
    return cnt, tn, fn, tf, ff


def raster_mask(tris, scale, res):
    mask = torch.zeros(res, res, dtype=torch.uint8, device=tris.device)
    check(lib().dsu_raster_mask(ptr(tris, torch.float32), tris.shape[0], float(scale), int(res),
                                ptr(mask), stream()), "dsu_raster_mask")
    return mask


def erode_ellipse_u8(mask, ksize):
    out = torch.empty_like(mask)
    check(lib().dsu_erode_ellipse_u8(ptr(mask, torch.uint8), mask.shape[0], mask.shape[1], int(ksize),
                                     ptr(out), stream()), "dsu_erode_ellipse_u8")
    return out


def knn8_blend(query_xy, known_xy, known_rgb):
    """interpolate_rgb: colours of the 8 nearest known points (xy distance, float64 like scipy's
    cKDTree), weights 1/(d+1e-6)."""
    q64 = query_xy.to(torch.float64).contiguous()
    k64 = known_xy.to(torch.float64).contiguous()
    known_rgb = _f32c(known_rgb)
    k32 = k64.to(torch.float32).contiguous()
    both = torch.cat([q64.to(torch.float32), k32], 0)
    grid = ZGrid(k32, both.amin(0).tolist(), both.amax(0).tolist(), points=True)
    out = torch.empty(q64.shape[0], 3, device=q64.device)
    x0, y0, cell, g, off, items = grid.args()
    check(lib().dsu_knn8_blend(ptr(q64, torch.float64), q64.shape[0], ptr(k64, torch.float64),
                               ptr(known_rgb), k64.shape[0], x0, y0, cell, g, off, items, ptr(out),
                               stream()), "dsu_knn8_blend")
    return out
