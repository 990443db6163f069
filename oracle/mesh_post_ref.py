"""CPU stand-ins for the third-party pieces behind the export's colour back-projection and thinning
offsets.  TEST INFRASTRUCTURE ONLY.

The reference's color_projection (instant_nsr/utils/coloring_utils.py:91-138) and get_offset_mask
(instant_nsr/utils/thinning_utils.py:96-193) are its OWN code and are run as they are by
tests/golden/make_mesh_color_golden.py; what they import is absent from this image and is served
from here (PARITY UNPINNED for these, as for every third-party op):

  * mesh_raycast.raycast(source, direction, mesh)   (un-vendored C extension): brute force over
    all triangles, float64, same hit rule as csrc/mesh_post.hip states — written independently
    with the Moeller-Trumbore form for a GENERAL direction, not the 2-D edge functions;
  * pytorch3d MeshRasterizer as used by MaskRenderer.render (zbuf > -1): numpy point-in-triangle
    rasteriser over the pixel centres of pytorch3d's NDC convention;
  * cv2.getStructuringElement(MORPH_ELLIPSE) / erode / flip / dilate: numpy restatements of
    OpenCV's published definitions.
"""
import numpy as np

MORPH_ELLIPSE = 2


def raycast(source, direction, mesh, eps_same=1e-6):
    """list of {'face', 'point', 'distance'} for every triangle the ray source + t * direction,
    t >= 0, meets (boundary included).  `mesh`: (F,3,3) float32 triangles.  Triangles that have
    `source` as a vertex (bit-equal after the float cast mesh_raycast applies to its arguments) are
    met at distance 0."""
    tri = np.asarray(mesh, np.float64)
    o = np.asarray(source, np.float32).astype(np.float64)
    d = np.asarray(direction, np.float64)
    out = []
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    incident = (np.all(a == o, 1) | np.all(b == o, 1) | np.all(c == o, 1))
    e1, e2 = b - a, c - a
    p = np.cross(d[None], e2)
    det = np.einsum("ij,ij->i", e1, p)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        tv = o[None] - a
        u = np.einsum("ij,ij->i", tv, p) * inv
        q = np.cross(tv, e1)
        v = np.einsum("ij,j->i", q, d) * inv
        t = np.einsum("ij,ij->i", q, e2) * inv
    tol = 1e-12
    hit = (det != 0) & (u >= -tol) & (v >= -tol) & (u + v <= 1 + tol) & (t >= 0)
    for f in np.nonzero(hit | incident)[0]:
        tt = 0.0 if incident[f] else float(t[f])
        pt = np.float32(o + tt * d).astype(np.float64)
        out.append({"face": int(f), "point": tuple(pt), "distance": float(np.float32(tt))})
    return out


class MaskRenderer:
    """coloring_utils.MaskRenderer stand-in: .render(v_np, f_np) -> (res,res) uint8, 255 where the
    orthographic view from +z covers the pixel centre (pytorch3d: NDC x to the LEFT, y UP, camera
    look_at_view_transform(1, 0, 0) flips x back: column c sees x = (2c+1)/res - 1, row r sees
    y = 1 - (2r+1)/res)."""

    def __init__(self, res):
        self.res = res

    def render(self, v_np, f_np):
        res = self.res
        v = np.asarray(v_np, np.float32).astype(np.float64)
        mask = np.zeros((res, res), np.uint8)
        for f in np.asarray(f_np).astype(np.int64):
            (ax, ay), (bx, by), (cx, cy) = v[f[0], :2], v[f[1], :2], v[f[2], :2]
            if (bx - ax) * (cy - ay) - (by - ay) * (cx - ax) == 0:
                continue
            c0 = int(np.ceil(((min(ax, bx, cx) + 1) * res - 1) / 2)); c1 = int(np.floor(((max(ax, bx, cx) + 1) * res - 1) / 2))
            r0 = int(np.ceil(((1 - max(ay, by, cy)) * res - 1) / 2)); r1 = int(np.floor(((1 - min(ay, by, cy)) * res - 1) / 2))
            c0, r0, c1, r1 = max(c0, 0), max(r0, 0), min(c1, res - 1), min(r1, res - 1)
            if c1 < c0 or r1 < r0:
                continue
            px = (2.0 * np.arange(c0, c1 + 1) + 1.0) / res - 1.0
            py = 1.0 - (2.0 * np.arange(r0, r1 + 1) + 1.0) / res
            X, Y = np.meshgrid(px, py)
            w0 = (X - bx) * (cy - by) - (Y - by) * (cx - bx)
            w1 = (X - cx) * (ay - cy) - (Y - cy) * (ax - cx)
            w2 = (X - ax) * (by - ay) - (Y - ay) * (bx - ax)
            ins = ((w0 >= 0) & (w1 >= 0) & (w2 >= 0)) | ((w0 <= 0) & (w1 <= 0) & (w2 <= 0))
            sub = mask[r0:r1 + 1, c0:c1 + 1]
            sub[ins] = 255
        return mask


def getStructuringElement(shape, ksize):
    """OpenCV's elliptic element: row i spans [c - dx, c + dx], dx = round(c sqrt(1 - dy^2 / r^2))."""
    assert shape == MORPH_ELLIPSE
    w, h = ksize
    r, c = h // 2, w // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    el = np.zeros((h, w), np.uint8)
    for i in range(h):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            el[i, max(c - dx, 0):min(c + dx + 1, w)] = 1
    return el


def _morph(img, kernel, iterations, op, pad_value):
    out = np.asarray(img)
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    for _ in range(iterations):
        p = np.pad(out, ((ay, kh - 1 - ay), (ax, kw - 1 - ax)), constant_values=pad_value)
        acc = None
        for i in range(kh):
            for j in range(kw):
                if kernel[i, j]:
                    s = p[i:i + out.shape[0], j:j + out.shape[1]]
                    acc = s.copy() if acc is None else op(acc, s)
        out = acc
    return out


def erode(img, kernel, iterations=1):
    return _morph(img, kernel, iterations, np.minimum, 255)     # default border: +inf


def dilate(img, kernel, iterations=1):
    return _morph(img, kernel, iterations, np.maximum, 0)       # default border: -inf


def flip(img, code):
    assert code == 1
    return np.ascontiguousarray(np.asarray(img)[:, ::-1])
