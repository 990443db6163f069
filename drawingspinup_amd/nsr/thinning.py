"""thinning_processing of the NSR export (2_charactor_reconstructor/instant_nsr/utils/
thinning_utils.py:199-247): thin limbs of the character (small distance-map values along the mask's
skeleton) are squeezed in z by a smooth deformation field,

    distance map + skeleton of the character mask     cv2.distanceTransform / skimage skeletonize
    -> fixed vertices   (distance >= theta_1 at the vertex' pixel)
    -> thin pixels      (skeleton where distance <= theta_2, minus discs around its loose ends)
    -> offsets          get_offset_mask: z-rays through the mesh at the thin pixels   (device)
    -> d = igl.harmonic(v, f, b, d_bc, 2);  v += d    bi-harmonic field with those Dirichlet data

The image steps are host functions of the library (csrc/thinning_host.hip), the offsets are the
device kernels of nsr/mesh_post.py, the bi-harmonic solve is a sparse host solve (scipy), as the
reference's is (libigl).  OpenCV, scikit-image and libigl are absent here: PARITY UNPINNED for
those packages' steps — restated from their published algorithms (kernel file header; below).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib, ops


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def distance_transform(mask_u8):
    """cv2.distanceTransform(mask, cv2.DIST_L2, 5): (H,W) uint8 -> (H,W) float32."""
    m = np.ascontiguousarray(np.asarray(mask_u8, dtype=np.uint8))
    out = np.empty(m.shape, np.float32)
    ops.check(_lib.lib().dsu_distance_transform_l2_5x5(_p(m), m.shape[0], m.shape[1], _p(out)),
              "dsu_distance_transform_l2_5x5")
    return out


def skeletonize(mask_u8):
    """skimage.morphology.skeletonize(mask, method='lee') as the reference uses its result
    (uint8, 0 / 255)."""
    m = np.ascontiguousarray(np.asarray(mask_u8, dtype=np.uint8))
    out = np.empty(m.shape, np.uint8)
    ops.check(_lib.lib().dsu_skeletonize_lee_2d(_p(m), m.shape[0], m.shape[1], _p(out)),
              "dsu_skeletonize_lee_2d")
    return out


def get_end_points(skeleton):
    """thinning_utils.py:11-26: skeleton pixels whose 3x3 neighbourhood holds exactly two object
    pixels (itself and one neighbour), as (col, row) tuples in raster order."""
    s = (np.asarray(skeleton) > 0).astype(np.int32)
    pad = np.pad(s, 1)
    cnt = sum(pad[1 + dr:pad.shape[0] - 1 + dr, 1 + dc:pad.shape[1] - 1 + dc]
              for dr in (-1, 0, 1) for dc in (-1, 0, 1))
    ends = (s > 0) & (cnt == 2)
    # the reference slices skeleton[row-1:row+2, col-1:col+2]: at row 0 or column 0 the start index
    # -1 wraps and the slice is EMPTY (count 0), so pixels there are never end points; the last row
    # / column only lose the neighbours outside the image
    ends[0, :] = False
    ends[:, 0] = False
    rows, cols = np.nonzero(ends)
    return [(int(c), int(r)) for r, c in zip(rows, cols)]


def _fill_circle(img, center, radius, color):
    """cv2.circle(img, center, radius, color, -1): the filled disc, rows |dy| <= r with the span
    floor(sqrt(r^2 - dy^2)) (OpenCV's midpoint rasteriser can differ by one pixel on a few rows:
    unpinned)."""
    cx, cy = center
    H, W = img.shape[:2]
    for dy in range(-radius, radius + 1):
        y = cy + dy
        if y < 0 or y >= H:
            continue
        half = int(np.floor(np.sqrt(radius * radius - dy * dy)))
        img[y, max(cx - half, 0):min(cx + half, W - 1) + 1] = color


def remove_intersection(thin_mask, skeleton, r, color=0):
    """thinning_utils.py:29-35: the thin mask's loose ends that are not ends of the skeleton are
    where a thin limb joins a thick part — a disc of radius r is cleared around each."""
    ep2 = set(get_end_points(skeleton))
    for point in get_end_points(thin_mask):
        if point not in ep2:
            _fill_circle(thin_mask, point, r, color)
    return thin_mask


def get_thin_coords(thin_mask, res):
    """thinning_utils.py:49-57: 3x3-cross dilation, pixel (row, col) -> (x, y) = (col, -row) /
    (res - 1) -+ 0.5."""
    m = np.asarray(thin_mask) > 0
    pad = np.pad(m, 1)
    d = pad[1:-1, 1:-1] | pad[:-2, 1:-1] | pad[2:, 1:-1] | pad[1:-1, :-2] | pad[1:-1, 2:]
    coords = np.argwhere(d).astype(np.float32) / (res - 1) - 0.5
    out = np.zeros(coords.shape)
    out[:, 0] = coords[:, 1]
    out[:, 1] = -coords[:, 0]
    return out


def bilinear_interpolation(image, xy):
    """thinning_utils.py:69-93."""
    h, w = image.shape[:2]
    x_int = np.floor(xy[:, 0]).astype(int)
    y_int = np.floor(xy[:, 1]).astype(int)
    x_frac, y_frac = xy[:, 0] - x_int, xy[:, 1] - y_int
    x_int = np.clip(x_int, 0, w - 2)
    y_int = np.clip(y_int, 0, h - 2)
    v1, v2 = image[y_int, x_int], image[y_int, x_int + 1]
    v3, v4 = image[y_int + 1, x_int], image[y_int + 1, x_int + 1]
    return (1 - x_frac) * (1 - y_frac) * v1 + x_frac * (1 - y_frac) * v2 + \
        (1 - x_frac) * y_frac * v3 + x_frac * y_frac * v4


def get_coord_dist(xy, dist_map, res):
    """thinning_utils.py:60-66."""
    tmp = np.array(xy, dtype=np.float64, copy=True)
    tmp[:, 1] *= -1
    tmp = (tmp + 0.5) * (res - 1)
    return bilinear_interpolation(dist_map, tmp)


# ------------------------------------------------------------------------------------------------
# igl.harmonic(V, F, b, bc, k)
# ------------------------------------------------------------------------------------------------
def cotmatrix(v, f):
    """igl::cotmatrix: L_ij = (cot a_ij + cot b_ij) / 2 over the two angles opposite edge ij,
    L_ii = -sum_j L_ij (negative semi-definite)."""
    import scipy.sparse as sp
    v = np.asarray(v, np.float64)
    f = np.asarray(f, np.int64)
    n = v.shape[0]
    rows, cols, vals = [], [], []
    for k in range(3):
        i, j, o = f[:, (k + 1) % 3], f[:, (k + 2) % 3], f[:, k]      # edge ij opposite corner o
        a, b = v[i] - v[o], v[j] - v[o]
        cot = (a * b).sum(1) / np.maximum(np.linalg.norm(np.cross(a, b), axis=1), 1e-300)
        rows += [i, j]; cols += [j, i]; vals += [0.5 * cot, 0.5 * cot]
    L = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(n, n)).tocsr()
    return L - sp.diags(np.asarray(L.sum(1)).ravel())


def massmatrix_voronoi(v, f):
    """igl::massmatrix(V, F, MASSMATRIX_TYPE_VORONOI) (the default for triangles): per corner the
    Voronoi cell inside the triangle — area x (w_j + w_k) / 2 with w the circumcentre's barycentric
    coordinates — and for an obtuse triangle half the area at the obtuse corner, a quarter at the
    other two.  Returned as the diagonal (n,)."""
    v = np.asarray(v, np.float64)
    f = np.asarray(f, np.int64)
    l2 = np.stack([((v[f[:, (k + 1) % 3]] - v[f[:, (k + 2) % 3]]) ** 2).sum(1) for k in range(3)], 1)
    area = 0.5 * np.linalg.norm(np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]), axis=1)
    # w_i ~ l_i^2 (l_j^2 + l_k^2 - l_i^2); the bracket is 2 l_j l_k cos(angle_i)
    cosb = np.stack([l2[:, (k + 1) % 3] + l2[:, (k + 2) % 3] - l2[:, k] for k in range(3)], 1)
    w = l2 * cosb
    ws = w.sum(1, keepdims=True)
    w = np.divide(w, ws, out=np.full_like(w, 1.0 / 3.0), where=np.abs(ws) > 0)
    part = np.stack([area * (w[:, (k + 1) % 3] + w[:, (k + 2) % 3]) * 0.5 for k in range(3)], 1)
    obt = cosb < 0
    any_obt = obt.any(1)
    part[any_obt] = np.where(obt[any_obt], 0.5, 0.25) * area[any_obt, None]
    m = np.zeros(v.shape[0])
    np.add.at(m, f.ravel(), part.ravel())
    return m


def _weld(v, f, tol=1e-9):
    """Marching cubes puts several vertices on one lattice corner whenever the volume is exactly at
    the iso value there (zero-length edges, zero-area triangles): the cotangent weights of such
    triangles are 0 / 0.  -> (representative index per vertex, faces over the representatives
    without the degenerate ones)."""
    v = np.asarray(v, np.float64)
    f = np.asarray(f, np.int64)
    scale = max(float(np.abs(v).max()) if len(v) else 1.0, 1e-30)
    key = np.round(v / (tol * scale)).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    rep = first[inv.ravel()]                                     # lowest index of each cluster
    g = rep[f]
    ok = (g[:, 0] != g[:, 1]) & (g[:, 1] != g[:, 2]) & (g[:, 0] != g[:, 2])
    g = g[ok]
    if len(g):
        area = 0.5 * np.linalg.norm(np.cross(v[g[:, 1]] - v[g[:, 0]], v[g[:, 2]] - v[g[:, 0]]), axis=1)
        g = g[area > 1e-10 * np.median(area)]
    return rep, g


def harmonic(v, f, b, bc, k=2, weld=True):
    """igl.harmonic(V, F, b, bc, k): W minimising the k-harmonic energy trace(W^T Q W), Q = L for
    k = 1 and (L M^-1 L) for k = 2 ..., subject to W[b] = bc (igl::harmonic -> min_quad_with_fixed):
    Q_uu W_u = -Q_ub bc, sparse LU.  weld: coincident vertices are solved as one and zero-area
    triangles left out (`_weld`; libigl would return NaN on a raw marching-cubes mesh — the
    reference only ever hands it the decimated one); a vertex left without any triangle keeps 0."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import splu
    v = np.asarray(v, np.float64)
    b = np.asarray(b, np.int64).ravel()
    bc = np.asarray(bc, np.float64).reshape(len(b), -1)
    n = v.shape[0]
    if weld:
        rep, g = _weld(v, f)
        used = np.zeros(n, bool)
        used[g.ravel()] = True
        if not (np.array_equal(rep, np.arange(n)) and used.all() and len(g) == len(f)):
            # compact to the representatives that carry triangles, constraints on their clusters
            idx = np.flatnonzero(used)
            new = np.full(n, -1, np.int64)
            new[idx] = np.arange(len(idx))
            bb = new[rep[b]]
            keep = bb >= 0
            _, firstb = np.unique(bb[keep], return_index=True)      # one constraint per cluster
            Wc = harmonic(v[idx], new[g], bb[keep][firstb], bc[keep][firstb], k, weld=False)
            W = np.zeros((n, bc.shape[1]))
            has = new[rep] >= 0
            W[has] = Wc[new[rep[has]]]
            W[b] = bc
            return W
    L = cotmatrix(v, f)
    Q = -L
    if k > 1:
        m = massmatrix_voronoi(v, f)
        Mi = sp.diags(1.0 / np.maximum(m, 1e-300))
        for _ in range(1, k):
            Q = -(Q @ Mi @ L)
    Q = Q.tocsr()
    known = np.zeros(n, bool)
    known[b] = True
    u = np.flatnonzero(~known)
    W = np.zeros((n, bc.shape[1]))
    W[b] = bc
    if len(u):
        rhs = -(Q[u][:, b] @ bc)
        W[u] = splu(Q[u][:, u].tocsc()).solve(rhs)
    return W


# ------------------------------------------------------------------------------------------------
def thinning_processing(v, f, mask_u8, thinning_type="double", theta_1=11, theta_2=6, r=11,
                        device=None, return_parts=False):
    """thinning_utils.py:199-247 on a mesh in save_mesh's front-facing convention: v (N,3) float64
    host array, f (M,3) int, mask_u8 the (res,res) character mask (`<uid>/char/mask.png`).
    Returns the deformed vertices (and, with return_parts, the intermediate images / data)."""
    from .mesh_post import get_offset_mask
    v = np.asarray(v, np.float64)
    f = np.asarray(f, np.int64)
    mask = np.ascontiguousarray(np.asarray(mask_u8, np.uint8))
    res = mask.shape[0]
    min_thickness = 1 / res
    distance = distance_transform(mask)
    skeleton = skeletonize(mask)
    fix_mask = get_coord_dist(v[:, 0:2], distance, res) >= theta_1
    mov_mask = skeleton * (distance <= theta_2)
    mov_mask_new = remove_intersection(mov_mask.copy(), skeleton, r)
    thin_coords = get_thin_coords(mov_mask_new, res)
    if thin_coords.shape[0] == 0 or len(f) == 0:
        # no thin limb: the reference's loop over the thin pixels marks nothing, the field's
        # Dirichlet data are all zero and so is the field
        if return_parts:
            return v.copy(), {"distance": distance, "skeleton": skeleton, "mov_mask": mov_mask,
                              "mov_mask_rm_inter": mov_mask_new, "thin_coords": thin_coords,
                              "fix_mask": fix_mask, "offset_mask": np.zeros(len(v), bool),
                              "offset_values": np.zeros_like(v), "d": np.zeros_like(v)}
        return v.copy()
    coord_dists = get_coord_dist(thin_coords[:, 0:2], distance, res) / res
    dev = torch.device(device if device is not None else "cuda")
    off_v, off_m = get_offset_mask(torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev),
                                   torch.from_numpy(thin_coords).to(dev),
                                   torch.from_numpy(coord_dists).to(dev), min_thickness, thinning_type)
    offset_values, offset_mask = off_v.cpu().numpy(), off_m.cpu().numpy()
    s = fix_mask | offset_mask
    b = np.flatnonzero(s)
    d = harmonic(v, f, b, offset_values[s], 2) if offset_mask.any() else np.zeros_like(v)
    out = v + d
    if return_parts:
        return out, {"distance": distance, "skeleton": skeleton, "mov_mask": mov_mask,
                     "mov_mask_rm_inter": mov_mask_new, "thin_coords": thin_coords,
                     "coord_dists": coord_dists, "fix_mask": fix_mask, "offset_mask": offset_mask,
                     "offset_values": offset_values, "d": d}
    return out
