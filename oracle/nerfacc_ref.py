"""CPU restatement of the nerfacc==0.3.3 ops on the NSR path.  TEST INFRASTRUCTURE ONLY.

nerfacc is an un-vendored dependency (requirements.txt:14); its source is not under
/root/reference, so this file restates the published v0.3.3 algorithm
(nerfacc/cuda/csrc/{ray_marching.cu,intersection.cu,render_weight.cu}, nerfacc/grid.py)
and anchors on the reference's call sites:
  2_charactor_reconstructor/instant_nsr/models/neus.py:53-57   OccupancyGrid(roi, 128, AABB)
  neus.py:84        occupancy_grid.every_n_step(step, occ_eval_fn, occ_thre)
  neus.py:119-129   ray_marching(o, d, scene_aabb, grid, step, stratified, cone_angle=0)
  neus.py:147-152   render_weight_from_alpha / accumulate_along_rays
PARITY UNPINNED: the reference holds no tests or golden vectors for these ops; the
known-answer tests in tests/test_oracle_nsr.py::test_nerfacc_kats are hand-derived.
All float arithmetic is done in numpy float32 scalar ops to mirror the f32 device code.
"""
import numpy as np

f32 = np.float32


def ray_aabb_intersect(o, d, aabb):
    o = o.astype(f32); d = d.astype(f32); aabb = np.asarray(aabb, f32)
    n = o.shape[0]
    near = np.full(n, f32(1e10)); far = np.full(n, f32(1e10))
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(n):
            tmin = (aabb[0] - o[i, 0]) / d[i, 0]; tmax = (aabb[3] - o[i, 0]) / d[i, 0]
            if tmin > tmax: tmin, tmax = tmax, tmin
            tymin = (aabb[1] - o[i, 1]) / d[i, 1]; tymax = (aabb[4] - o[i, 1]) / d[i, 1]
            if tymin > tymax: tymin, tymax = tymax, tymin
            if tmin > tymax or tymin > tmax: continue
            if tymin > tmin: tmin = tymin
            if tymax < tmax: tmax = tymax
            tzmin = (aabb[2] - o[i, 2]) / d[i, 2]; tzmax = (aabb[5] - o[i, 2]) / d[i, 2]
            if tzmin > tzmax: tzmin, tzmax = tzmax, tzmin
            if tmin > tzmax or tzmin > tmax: continue
            if tzmin > tmin: tmin = tzmin
            if tzmax < tmax: tmax = tzmax
            near[i] = tmin; far[i] = tmax
    return near, far


def _occupied(p, mn, mx, occ, res):
    if occ is None:
        return True
    if (p < mn).any() or (p > mx).any():
        return False
    u = (p - mn) / (mx - mn)
    ix = np.clip((u * f32(res)).astype(np.int32), 0, res - 1)  # trunc toward zero, u>=0
    return bool(occ[(int(ix[0]) * res + int(ix[1])) * res + int(ix[2])])


def _fmin(a, b):
    if np.isnan(a): return b
    if np.isnan(b): return a
    return a if a < b else b


def _dist_to_next_voxel(p, d, inv, mn, mx, res):
    t = f32(np.inf)
    with np.errstate(invalid="ignore", over="ignore"):
        for k in range(3):
            g = ((p[k] - mn[k]) / (mx[k] - mn[k])) * f32(res)
            sg = f32(np.copysign(1.0, d[k]))
            td = ((np.floor(g + f32(0.5) + f32(0.5) * sg) - g) * inv[k]) / f32(res) * (mx[k] - mn[k])
            t = _fmin(t, f32(td))
    return f32(max(t, f32(0.0))) if not np.isnan(t) else f32(0.0)


def ray_marching(o, d, t_min, t_max, aabb, occ_binary, res, step):
    """-> ray_indices (int64), t_starts, t_ends (f32), counts (int32 per ray)."""
    o = o.astype(f32); d = d.astype(f32); aabb = np.asarray(aabb, f32)
    mn, mx = aabb[:3], aabb[3:]
    dt = f32(step)
    ri, ts, te, counts = [], [], [], []
    occ = None if occ_binary is None else np.asarray(occ_binary).reshape(-1)
    with np.errstate(divide="ignore"):
        for i in range(o.shape[0]):
            inv = (f32(1.0) / d[i]).astype(f32)
            near, far = f32(t_min[i]), f32(t_max[i])
            j = 0
            t0 = near; t1 = f32(t0 + dt); tm = f32((t0 + t1) * f32(0.5))
            while tm < far:
                p = (o[i] + tm * d[i]).astype(f32)
                if _occupied(p, mn, mx, occ, res):
                    ri.append(i); ts.append(t0); te.append(t1); j += 1
                    t0 = t1; t1 = f32(t0 + dt); tm = f32((t0 + t1) * f32(0.5))
                else:
                    target = f32(tm + _dist_to_next_voxel(p, d[i], inv, mn, mx, res))
                    target = _fmin(target, far)
                    while True:
                        tm = f32(tm + dt)
                        if not (tm < target): break
                    t0 = f32(tm - dt * f32(0.5)); t1 = f32(tm + dt * f32(0.5))
            counts.append(j)
    return (np.asarray(ri, np.int64), np.asarray(ts, f32), np.asarray(te, f32),
            np.asarray(counts, np.int32))


def render_weight_from_alpha(alpha, counts):
    alpha = alpha.astype(np.float64)
    w = np.empty_like(alpha)
    b = 0
    for c in counts:
        T = 1.0
        for j in range(b, b + c):
            w[j] = alpha[j] * T
            T *= (1.0 - alpha[j])
        b += c
    return w


def render_weight_from_alpha_bwd(alpha, counts, gw):
    alpha = alpha.astype(np.float64); gw = gw.astype(np.float64)
    w = render_weight_from_alpha(alpha, counts)
    ga = np.empty_like(alpha)
    b = 0
    for c in counts:
        accum = float((gw[b:b + c] * w[b:b + c]).sum())
        T = 1.0
        for j in range(b, b + c):
            ga[j] = (gw[j] * T - accum) / max(1.0 - alpha[j], 1e-10)
            accum -= gw[j] * w[j]
            T *= (1.0 - alpha[j])
        b += c
    return ga


def accumulate_along_rays(w, values, ray_indices, n_rays):
    w = w.astype(np.float64)
    if values is None:
        src = w[:, None]
    else:
        src = w[:, None] * values.astype(np.float64)
    out = np.zeros((n_rays, src.shape[1]), np.float64)
    np.add.at(out, ray_indices, src)
    return out


def occgrid_update(occs, indices, occ, decay=0.95, occ_thre=0.01):
    """OccupancyGrid._update tail (nerfacc/grid.py @0.3.3): EMA max + mean-clamped threshold."""
    occs = occs.astype(f32).copy()
    if indices is None:
        indices = np.arange(occs.shape[0])
    occs[indices] = np.maximum(occs[indices] * f32(decay), occ.astype(f32))
    thre = min(float(occs.mean()), occ_thre)
    return occs, occs > f32(thre)
