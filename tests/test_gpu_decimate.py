"""Device rounds of the quadric decimation (csrc/mesh_decimate_gpu.hip, nsr.mesh.remesh on device
tensors) held to the contracts the serial host queue is held to in tests/test_export_host.py: exact
face count, closed 2-manifold of the same genus, orientation, vertices on the input surface, faces
spent where the surface bends, boundary outline kept, deterministic — plus an export-scale mesh
(marching cubes of a 384^3 volume, ~0.8 M triangles) with the time of the call."""
import time

import numpy as np
import pytest
import torch

from drawingspinup_amd.nsr import mesh as M
from tests.test_export_host import _edge_counts, _grid, _signed_volume, _uv_sphere

pytestmark = pytest.mark.gpu


def _dev(v, f, dev):
    return torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)


def test_parallel_remesh_sphere_contract(dev):
    v, f = _uv_sphere(320, 240)                                       # 153 k triangles
    v2, f2 = M.remesh(*_dev(v, f, dev), 3000)
    st = dict(M.last_remesh_stats)
    assert st["input_faces"] == len(f) and 3000 <= st["device_faces"] <= 12000 and st["rounds"] >= 5
    assert f2.shape == (3000, 3) and f2.dtype == np.int64 and v2.dtype == np.float64
    assert f2.min() == 0 and f2.max() == v2.shape[0] - 1 and len(np.unique(f2)) == v2.shape[0]
    u, c = _edge_counts(f2)
    assert np.all(c == 2)                                             # closed 2-manifold ...
    assert v2.shape[0] - u.shape[0] + f2.shape[0] == 2                # ... of genus 0
    d = np.concatenate([f2[:, [0, 1]], f2[:, [1, 2]], f2[:, [2, 0]]])
    assert len(np.unique(d, axis=0)) == len(d)                        # consistent orientation
    n = np.cross(v2[f2[:, 1]] - v2[f2[:, 0]], v2[f2[:, 2]] - v2[f2[:, 0]])
    assert np.all(np.einsum("ij,ij->i", n, v2[f2].mean(1)) > 0)       # outward, none flipped
    r = np.linalg.norm(v2, axis=1)
    assert abs(r - 0.5).max() < 1e-3                                  # within 0.2 % of the radius
    assert abs(_signed_volume(v2, f2) / _signed_volume(v, f) - 1) < 5e-3
    v3, f3 = M.remesh(*_dev(v, f, dev), 3000)                         # deterministic
    assert np.array_equal(v2, v3) and np.array_equal(f2, f3)
    # the serial queue alone from the same input: same quality class (volume, radius), not the
    # same mesh (the collapse order differs)
    hv, hf = M.remesh(v, f, 3000)
    assert abs(np.linalg.norm(hv, axis=1) - 0.5).max() < 1e-3
    print("sphere: device stats", st, "radius error device %.2e host %.2e"
          % (abs(r - 0.5).max(), abs(np.linalg.norm(hv, axis=1) - 0.5).max()))


def test_parallel_remesh_spends_faces_where_the_surface_bends(dev):
    bump = lambda x, y: 0.2 * np.exp(-(x * x + y * y) / (2 * 0.05 ** 2))
    v, f = _grid(301, bump)                                           # 180 k triangles, open surface
    v2, f2 = M.remesh(*_dev(v, f, dev), 1500)
    assert f2.shape[0] in (1499, 1500)
    assert np.abs(v2[:, 2] - bump(v2[:, 0], v2[:, 1])).max() < 4e-3
    rad = np.linalg.norm(v2[f2].mean(1)[:, :2], axis=1)
    assert (rad < 0.15).sum() > (rad >= 0.15).sum()
    assert np.allclose(v2[:, :2].min(0), -0.5, atol=1e-6) and np.allclose(v2[:, :2].max(0), 0.5, atol=1e-6)
    a = v2[f2]
    p, q = a[:, 1, :2] - a[:, 0, :2], a[:, 2, :2] - a[:, 0, :2]
    assert abs(0.5 * np.abs(p[:, 0] * q[:, 1] - p[:, 1] * q[:, 0]).sum() - 1.0) < 1e-3
    u, c = _edge_counts(f2)
    assert set(np.unique(c)) <= {1, 2}


def test_parallel_remesh_at_export_scale(dev):
    """A marching-cubes mesh of the size class the export produces (a blobby shape on a 384^3
    lattice), down to the reference's face_count = 50 000."""
    n = 384
    c = torch.linspace(-1, 1, n, device=dev)
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    vol = 0.9 - torch.sqrt((x / 0.95) ** 2 + (y / 0.8) ** 2 + (z / 0.9) ** 2) \
        + 0.08 * torch.sin(9 * x) * torch.sin(7 * y + 1) * torch.sin(8 * z + 2)
    v, f = M.marching_cubes(vol.double(), 0.0)
    assert f.shape[0] > 600000
    torch.cuda.synchronize()
    t0 = time.time()
    v2, f2 = M.remesh(v, f, 50000)
    dt = time.time() - t0
    st = dict(M.last_remesh_stats)
    print("export scale: %d -> %d faces in %.3f s; device stats %s" % (f.shape[0], f2.shape[0], dt, st))
    assert f2.shape[0] in (49999, 50000)
    u, cnt = _edge_counts(f2)
    assert np.all(cnt == 2)                                           # closed manifold kept
    vin, fin = v.cpu().numpy(), f.cpu().numpy()
    chi_in = vin.shape[0] - _edge_counts(fin)[0].shape[0] + fin.shape[0]
    assert v2.shape[0] - u.shape[0] + f2.shape[0] == chi_in           # same Euler characteristic
    assert abs(_signed_volume(v2, f2) / _signed_volume(vin, fin) - 1) < 2e-3
    # every output vertex lies on the input iso-surface to a fraction of a lattice cell
    pts = torch.from_numpy(v2).to(dev).float()
    g = (pts / (n - 1.0) * 2 - 1)[None, None, None][..., [2, 1, 0]]
    val = torch.nn.functional.grid_sample(vol[None, None], g, mode="bilinear", align_corners=True).flatten()
    grad = 0.5 * (n - 1) / 2                                          # |d vol / d lattice unit| ~ 1/128 .. : bound below
    assert float(val.abs().max()) < 0.02
    assert dt < 2.0
