"""Export tail on the device: colour back-projection and thinning offsets
(drawingspinup_amd/nsr/mesh_post.py, csrc/mesh_post.hip) against the REFERENCE's own
color_projection / get_offset_mask run on a seeded mesh (tests/golden/mesh_color_reference.npz, made
by tests/golden/make_mesh_color_golden.py with stand-ins for the absent third-party pieces), and
the individual kernels against those stand-ins / scipy."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from drawingspinup_amd import ops
from drawingspinup_amd.nsr import mesh_post as MP
from oracle import mesh_post_ref as R

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mesh_color_reference.npz"))


def _big(key, res):
    img = Image.fromarray(GOLD[key]) if GOLD[key].ndim == 3 else Image.fromarray(GOLD[key], "L")
    return np.array(img.resize((res, res), Image.LANCZOS))


def test_color_projection_matches_reference_function(dev):
    res = int(GOLD["res"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    colors = MP.color_projection(t(GOLD["verts"]), t(GOLD["faces"]), t(_big("color_front", res)),
                                 t(_big("mask_front", res)), t(_big("color_back", res)), res=res)
    got, want = colors.cpu().numpy(), GOLD["vert_colors"]
    # three populations: coloured from the front image, from the back image, interpolated
    direct = np.abs(got - want).max(1) < 1e-6
    assert direct.mean() > 0.995, direct.mean()       # a vertex may flip population at a pixel / edge tie
    np.testing.assert_allclose(got[direct], want[direct], rtol=0, atol=1e-6)
    assert np.abs(got - want).max() < 0.25 and np.abs(got - want).mean() < 2e-4


@pytest.mark.parametrize("ty", ["double", "front", "back"])
def test_thinning_offsets_match_reference_function(dev, ty):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ov, om = MP.get_offset_mask(t(GOLD["verts"]), t(GOLD["faces"]), t(GOLD["thin_coords"]),
                                t(GOLD["coord_dists"]), float(GOLD["min_thickness"]), ty)
    want_v, want_m = GOLD["offset_values_" + ty], GOLD["offset_mask_" + ty]
    assert want_m.sum() > 100
    assert np.array_equal(om.cpu().numpy(), want_m)
    np.testing.assert_allclose(ov.cpu().numpy(), want_v, rtol=0, atol=2e-7)


def test_zray_cast_matches_brute_force_raycast(dev):
    v, f = GOLD["verts"], GOLD["faces"]
    tris = torch.from_numpy(v.astype(np.float32)[f]).to(dev).contiguous()
    faces_i = torch.from_numpy(f.astype(np.int32)).to(dev)
    lo, hi = tris[..., :2].reshape(-1, 2).amin(0).tolist(), tris[..., :2].reshape(-1, 2).amax(0).tolist()
    grid = ops.ZGrid(tris, lo, hi)
    rng = np.random.default_rng(0)
    src = np.concatenate([rng.uniform(-0.45, 0.45, (300, 2)), np.ones((300, 1))], 1).astype(np.float32)
    cnt, tn, fn, tf, ff = [x.cpu().numpy() for x in ops.zray_cast(grid, faces_i, torch.from_numpy(src).to(dev), -1)]
    tri_np = v.astype(np.float32)[f]
    for i in range(300):
        rs = R.raycast(src[i], (0, 0, -1), tri_np)
        assert len(rs) == cnt[i]
        if rs:
            near, far = min(rs, key=lambda r: r["distance"]), max(rs, key=lambda r: r["distance"])
            assert abs(near["distance"] - tn[i]) < 1e-6 and abs(far["distance"] - tf[i]) < 1e-6
    # rays from mesh vertices: incident faces at distance exactly 0, visibility = far distance 0
    ids = rng.choice(len(v), 300, replace=False)
    o = torch.from_numpy(v.astype(np.float32)[ids]).to(dev)
    cnt, tn, fn, tf, ff = [x.cpu().numpy() for x in
                           ops.zray_cast(grid, faces_i, o, +1, torch.from_numpy(ids.astype(np.int32)).to(dev))]
    vis = 0
    for k, vid in enumerate(ids):
        rs = R.raycast(v[vid], (0, 0, 1), tri_np)
        assert len(rs) == cnt[k] and cnt[k] >= 3
        far = max(r["distance"] for r in rs)
        assert (far == 0) == (tf[k] == 0)
        vis += far == 0
    assert 50 < vis < 280


def test_silhouette_and_erosion_match_stand_ins(dev):
    v, f = GOLD["verts"], GOLD["faces"]
    res = 256
    got = MP.render_mask(torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev), res, 2.0).cpu().numpy()
    want = R.MaskRenderer(res).render(v * 2, f)
    assert want.sum() > 255 * 1000 and np.array_equal(got, want)
    el = R.getStructuringElement(R.MORPH_ELLIPSE, (19, 19))
    g = torch.Generator().manual_seed(0)
    noisy = ((torch.rand(200, 160, generator=g) > 0.02) * 255).to(torch.uint8)
    for img in (torch.from_numpy(want), noisy):
        e = ops.erode_ellipse_u8(img.to(dev).contiguous(), 19).cpu().numpy()
        assert np.array_equal(e, R.erode(img.numpy(), el))


def test_knn8_blend_matches_ckdtree(dev):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(3)
    known = np.concatenate([rng.uniform(-0.5, 0.5, (4000, 2)), rng.normal(0.3, 0.01, (500, 2))]).astype(np.float32)
    rgb = rng.random((4500, 3)).astype(np.float32)
    query = np.concatenate([rng.uniform(-0.5, 0.5, (1500, 2)), rng.uniform(0.6, 0.9, (20, 2))]).astype(np.float32)
    got = ops.knn8_blend(torch.from_numpy(query).to(dev), torch.from_numpy(known).to(dev),
                         torch.from_numpy(rgb).to(dev)).cpu().numpy()
    d, idx = cKDTree(known.astype(np.float64)).query(query.astype(np.float64), 8)
    w = 1.0 / (d + 1e-6)
    w /= w.sum(1, keepdims=True)
    want = np.einsum("ijk,ij->ik", rgb[idx].astype(np.float64), w)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


def test_device_laplacian_smoothing_equals_the_sparse_lu_filter(dev):
    """save_mesh's implicit Laplacian filter (mesh_utils.py:42-45) on device tensors — Jacobi sweeps
    of dsu_umbrella_implicit_solve + device volume rescaling — against the host form (sparse LU of
    the same matrix, pinned to the reference's save_mesh by tests/test_mesh_host.py): 1e-11, and the
    whole post_process_mesh device path (smoothing + shear + scale) against the host path."""
    import numpy as np
    from drawingspinup_amd.nsr import mesh as M
    from tests.test_export_host import _uv_sphere
    v, f = _uv_sphere(96, 64)
    g = np.random.default_rng(0)
    v = v * (1 + 0.05 * g.standard_normal((v.shape[0], 1))) + 0.01 * g.standard_normal(v.shape)
    want = M.laplacian_smooth_implicit(v, f, lamb=2.0, iterations=5)
    got = M.laplacian_smooth_implicit_device(torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev),
                                             lamb=2.0, iterations=5).cpu().numpy()
    assert np.abs(got - want).max() < 1e-11
    hv, hf, _ = M.post_process_mesh(torch.from_numpy(v), torch.from_numpy(f), None, smoothing=True, shearing=True)
    dv, df, _ = M.post_process_mesh(torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev), None,
                                    smoothing=True, shearing=True)
    assert np.array_equal(hf, df) and np.abs(hv - dv).max() < 1e-10
