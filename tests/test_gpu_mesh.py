"""NSR export tail on the device: the HIP weighted-Jacobi smoothing (csrc/mesh_smooth.hip) and the
tensor-program marching cubes against the serial restatement of PyMCubes (oracle/mcubes_ref.py)."""
import numpy as np
import pytest
import torch

from drawingspinup_amd.nsr import mesh as M
from oracle import mcubes_ref as R

pytestmark = pytest.mark.gpu


def _shape(n):
    c = torch.linspace(-1, 1, n)
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    return ((x / 0.7) ** 2 + (y / 0.5) ** 2 + (z / 0.6) ** 2 <= 1.0) | \
        ((x - 0.3).abs() + y.abs() + z.abs() < 0.35)


@pytest.mark.parametrize("iters", [60, 25])
def test_device_smoothing_matches_scipy_restatement(dev, iters):
    b = _shape(20)
    got = M.smooth_constrained(b.to(dev), max_iters=iters).cpu().numpy()
    ref = R.smooth_constrained(b.numpy(), max_iters=iters)
    near = np.abs(R.signed_distance_function(b.numpy())) <= 4.5
    np.testing.assert_allclose(got[near], ref[near], rtol=0, atol=1e-9)
    assert np.all(got[b.numpy()] >= 0) and np.all(got[~b.numpy()] <= 0)
    # INT: faces of the smoothed field bit-exact vs the serial sweep on the oracle's field
    v, f = M.marching_cubes(torch.from_numpy(got).to(dev), 0.0)
    rv, rf = R.marching_cubes(ref, 0.0)                            # the oracle's own classic table
    assert np.array_equal(f.cpu().numpy(), rf)
    np.testing.assert_allclose(v.cpu().numpy(), rv, rtol=0, atol=1e-7)


def _field(n, seed):
    """A smooth field with structure at several scales and a few thousand values that are EXACTLY
    the iso value (what the projection of mcubes.smooth produces): the `<=` corner rule decides."""
    c = torch.linspace(-1, 1, n, dtype=torch.float64)
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    g = torch.Generator().manual_seed(seed)
    ph = torch.rand(6, generator=g, dtype=torch.float64) * 6.28
    vol = 0.55 - torch.sqrt((x / 0.8) ** 2 + (y / 0.6) ** 2 + (z / 0.7) ** 2) \
        + 0.08 * torch.sin(9 * x + ph[0]) * torch.sin(7 * y + ph[1]) * torch.sin(8 * z + ph[2]) \
        + 0.03 * torch.sin(23 * x + ph[3]) * torch.sin(19 * y + ph[4]) * torch.sin(21 * z + ph[5])
    vol = torch.where(vol.abs() < 2e-3, torch.zeros_like(vol), vol)
    return vol


@pytest.mark.parametrize("n,seed", [(128, 0), (256, 1)])
def test_marching_cubes_at_export_scale_matches_the_independent_oracle(dev, n, seed):
    """INT parity at >= 128^3: vertex numbering and face index arrays of the device program equal
    the serial sweep with the ORACLE's copy of the classic table (bit for bit), vertices too."""
    vol = _field(n, seed)
    assert int((vol == 0).sum()) > 500
    v, f = M.marching_cubes(vol.to(dev), 0.0)
    rv, rf = R.marching_cubes(vol.numpy(), 0.0)
    assert rf.shape[0] > 20000
    assert np.array_equal(f.cpu().numpy(), rf)
    assert np.array_equal(v.cpu().numpy(), rv)


def test_device_smoothing_equals_the_tensor_program_at_export_scale(dev):
    """96^3 (a band of ~10^5 voxels, many workgroups): HIP iteration == the torch restatement that
    the CPU tests pin to the oracle; early stopping takes the same decision."""
    b = _shape(96)
    got = M.smooth_constrained(b.to(dev)).cpu()
    ref = M.smooth_constrained(b)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=1e-10)


def test_smoothing_of_an_empty_volume(dev):
    b = torch.zeros(8, 8, 8, dtype=torch.bool, device=dev)
    out = M.smooth_constrained(b)
    assert out.shape == (8, 8, 8) and bool((out < 0).all())


def test_isosurface_glue_on_device_matches_reference_geometry_py(dev):
    """The device run of nsr.mesh.isosurface (HIP smoothing, tensor-program marching cubes, device
    bicubic resize / crop) against tests/golden/isosurface_reference.npz = the REFERENCE's own
    geometry.py:33-117 (see tests/test_mesh_host.py for the host run): faces bit-exact, vertices
    1e-7 (the device Jacobi sums in a different order than scipy)."""
    import os
    from tests.test_mesh_host import _Bits
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "isosurface_reference.npz"))

    class DevBits(_Bits):
        def isosurface_levels(self, vmin, vmax, res):
            return super().isosurface_levels(vmin, vmax, res).to(dev)

    model = DevBits(z)
    fine, coarse = M.isosurface(model, torch.from_numpy(z["front_mask"]).to(dev))
    assert fine["verts"].is_cuda
    assert np.array_equal(coarse["faces"].cpu().numpy(), z["coarse_faces"])
    assert np.array_equal(fine["faces"].cpu().numpy(), z["faces"])
    np.testing.assert_allclose(fine["verts"].cpu().numpy(), z["verts"], rtol=0, atol=1e-7)


def test_isosurface_lattice_points_match_reference_grid(dev):
    """NeuSModel.isosurface_levels evaluates the SDF on the lattice the reference builds with
    grid_vertices() + scale_anything (geometry.py:40-48,85-89): same x-major order, coordinates
    within one float32 ulp (linspace on the device vs the host)."""
    import os
    from drawingspinup_amd.nsr.model import NeuSModel
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "isosurface_reference.npz"))
    res = int(z["res"])
    model = NeuSModel().to(dev)
    seen = []
    model.geometry.forward_level = lambda pts: (seen.append(pts.clone()), pts[:, 0] * 0)[1]
    model.isosurface_levels([-1.0] * 3, [1.0] * 3, res, chunk=70000)
    coarse = torch.cat(seen, 0).cpu().numpy()
    seen.clear()
    # the fine box as the reference held it: float64 corners (here from the fixture's fine mesh
    # box: the lattice corners it produced)
    lo, hi = z["fine_pts"].min(0), z["fine_pts"].max(0)
    model.isosurface_levels(z["fine_vmin"].astype(np.float64), z["fine_vmax"].astype(np.float64), res,
                            chunk=70000)
    fine = torch.cat(seen, 0).cpu().numpy()
    idx = z["pts_index"]
    assert coarse.shape == (res ** 3, 3)
    np.testing.assert_allclose(coarse[idx], z["coarse_pts"], rtol=0, atol=1.2e-7)
    np.testing.assert_allclose(fine[idx], z["fine_pts"], rtol=0, atol=2.4e-7)
    assert lo.shape == hi.shape == (3,)


# ------------------------------------------------------------------ csrc/mesh_volume.hip
def _volumes():
    g = torch.Generator().manual_seed(3)
    blob = _shape(44)[:, :37, :]                                   # non-cubic, touches no face
    noise = torch.rand(23, 40, 70, generator=g) > 0.5              # every voxel next to the other class
    sparse = torch.rand(30, 30, 66, generator=g) > 0.985           # isolated voxels, long empty runs
    full = torch.ones(9, 10, 11, dtype=torch.bool)
    empty = torch.zeros(9, 10, 11, dtype=torch.bool)
    slab = torch.zeros(20, 18, 16, dtype=torch.bool)
    slab[:, :, :7] = True                                          # a class boundary through the faces
    one = torch.zeros(15, 15, 15, dtype=torch.bool)
    one[7, 7, 7] = True
    return {"blob": blob, "noise": noise, "sparse": sparse, "full": full, "empty": empty, "slab": slab,
            "one": one}


@pytest.mark.parametrize("radius,band_radius", [(5.0, 4.0), (3.0, 2.0), (2.5, 2.0), (8.0, 7.5)])
def test_band_distance_kernel_equals_the_tensor_program(dev, radius, band_radius):
    """dsu_volume_band_distance (integer squared distances through three byte passes, values through
    the host's table) == signed_distance_band_tensor_program ON THE DEVICE (the export's form until
    round 5) bit for bit, and its band byte == |dist| <= band_radius: blobs, noise, isolated voxels,
    one-class volumes, boundaries on faces.  (The tensor program on HOST tensors is compared to
    2 ulp only: torch's vectorised CPU sqrt is not correctly rounded for every integer — 2, 8, 19,
    32 differ from numpy's here — while the table holds numpy's, as scipy's C sqrt in the
    reference does.)"""
    for name, b in _volumes().items():
        ref = M.signed_distance_band_tensor_program(b.to(dev), radius).cpu()
        got, band = M.signed_distance_band_device(b.to(dev), radius, band_radius)
        assert got.dtype == torch.float64 and band.dtype == torch.bool, name
        assert torch.equal(got.cpu(), ref), name
        assert torch.equal(band.cpu(), ref.abs() <= band_radius), name
        assert torch.equal(M.signed_distance_band(b.to(dev), radius).cpu(), ref), name
        host = M.signed_distance_band_tensor_program(b, radius)
        np.testing.assert_allclose(got.cpu().numpy(), host.numpy(), rtol=0, atol=2e-15)


def test_cube_index_kernel_equals_the_tensor_program(dev):
    """dsu_mc_cube_index == the eight shifted comparisons of the host form, exact zeros included
    (the `<=` corner rule), on a non-cubic volume."""
    vol = _field(64, 5)[:, :50, :41].contiguous()
    assert int((vol == 0).sum()) > 50
    from drawingspinup_amd import ops
    got = ops.mc_cube_index(vol.to(dev), 0.0).cpu()
    X, Y, Z = vol.shape
    below = vol <= 0.0
    ref = torch.zeros((X - 1, Y - 1, Z - 1), dtype=torch.int64)
    for m, (dx, dy, dz) in enumerate(M.CORNERS):
        ref += below[dx:X - 1 + dx, dy:Y - 1 + dy, dz:Z - 1 + dz].to(torch.int64) << m
    assert torch.equal(got.to(torch.int64), ref)
    hv, hf = M.marching_cubes(vol, 0.0)                            # host form end to end
    dv, df = M.marching_cubes(vol.to(dev), 0.0)
    assert torch.equal(df.cpu(), hf) and torch.equal(dv.cpu(), hv)


def test_lattice_forward_equals_the_chunked_tensor_expression(dev):
    """dsu_sdf_fwd_lattice forms the export's lattice points in the kernel: the level volume is
    bit-identical to the chunked tensor expression + forward_level (same rounded float32
    coordinates, same network kernel), for the whole box and for an off-centre fine box."""
    from drawingspinup_amd.nsr.model import NeuSModel
    torch.manual_seed(11)
    model = NeuSModel().to(dev)
    res = 40
    for vmin, vmax in (([-1.0] * 3, [1.0] * 3), ([-0.4123456789, -0.61, -0.2000000001], [0.35, 0.58, 0.77])):
        a = model.isosurface_levels(vmin, vmax, res)
        b = model.isosurface_levels(vmin, vmax, res, chunk=5 * res * res)
        assert a.shape == (res, res, res) and torch.equal(a, b)
