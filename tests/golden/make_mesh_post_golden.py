"""Fixture for the host-side mesh post-processing: runs the REFERENCE's own
`save_mesh` / `shear_transformation` / `PCA` (2_charactor_reconstructor/instant_nsr/utils/
mesh_utils.py:25-93) on a small seeded mesh, with the switches this repository implements
(smoothing, nearest-vertex colours, shearing, ortho scale; thinning / colour back-projection / uv
off).  `trimesh` is not installed: a stand-in provides `Trimesh` (a record) and
`smoothing.filter_laplacian` as a DENSE numpy restatement of trimesh's published implicit umbrella
filter with volume constraint (unpinned for that one call; the product uses a sparse LU) — every
other number in the fixture comes from the reference's code.

    python tests/golden/make_mesh_post_golden.py     # needs /root/reference; writes mesh_post_reference.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/2_charactor_reconstructor/instant_nsr/utils/mesh_utils.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def seeded_mesh(level=2, seed=3):
    """Subdivided octahedron pushed onto a bumpy ellipsoid (closed, outward-oriented)."""
    v = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    v = [np.array(p, np.float64) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        f = nf
    v = np.array(v)
    rng = np.random.default_rng(seed)
    v = v * np.array([0.5, 0.3, 0.8]) * (1.0 + 0.05 * rng.standard_normal((len(v), 1)))
    v[:, 1] += 0.25 * v[:, 2]                                   # a lean for the shear step to remove
    return v, np.array(f, np.int64), rng.random((len(v), 3))


def dense_filter_laplacian(mesh, lamb=0.5, iterations=10, implicit_time_integration=False,
                           volume_constraint=True, laplacian_operator=None):
    assert implicit_time_integration
    v, f = np.array(mesh.vertices, np.float64), np.asarray(mesh.faces)
    n = len(v)
    adj = np.zeros((n, n))
    for a, b, c in f:
        adj[a, b] = adj[b, a] = adj[b, c] = adj[c, b] = adj[a, c] = adj[c, a] = 1.0
    L = adj / adj.sum(1, keepdims=True)
    AA = np.eye(n) + lamb * (np.eye(n) - L)

    def vol(p):
        return np.einsum("ij,ij->i", p[f[:, 0]], np.cross(p[f[:, 1]], p[f[:, 2]])).sum() / 6.0
    v0 = vol(v)
    for _ in range(iterations):
        v = np.linalg.solve(AA, v)
        if volume_constraint:
            v = v * (v0 / vol(v)) ** (1.0 / 3.0)
    mesh.vertices = v
    return mesh


def main():
    exported = {}

    class Trimesh:
        def __init__(self, vertices=None, faces=None, vertex_colors=None, **kw):
            self.vertices, self.faces, self.vertex_colors = vertices, faces, vertex_colors

        def export(self, path):
            exported.update(v=np.array(self.vertices), f=np.array(self.faces), c=np.array(self.vertex_colors))

    tm = types.ModuleType("trimesh")
    tm.Trimesh = Trimesh
    tm.smoothing = types.SimpleNamespace(filter_laplacian=dense_filter_laplacian)
    sys.modules["trimesh"] = tm
    for name, attrs in (("instant_nsr", {}), ("instant_nsr.utils", {}),
                        ("instant_nsr.utils.coloring_utils", dict(color_projection=None, uv_mapping=None)),
                        ("instant_nsr.utils.thinning_utils", dict(thinning_processing=None))):
        m = types.ModuleType(name)
        for k, val in attrs.items():
            setattr(m, k, val)
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ref_mesh_utils", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    verts, faces, colors = seeded_mesh()
    cfg = types.SimpleNamespace(output_dir="/tmp/_mesh_post_golden", thinning=False, smoothing=True,
                                color_back_projection=False, shearing=True, ortho_scale=1.35,
                                export_uv=False, save_name="m", input_dir=None)
    ref.save_mesh(cfg, verts.copy(), faces.copy(), colors.copy())
    sheared = ref.shear_transformation(verts.copy())
    pca = ref.PCA(verts[:, 1:3].copy())
    np.savez_compressed(os.path.join(HERE, "mesh_post_reference.npz"), verts=verts, faces=faces,
                        colors=colors, out_v=exported["v"], out_f=exported["f"], out_c=exported["c"],
                        sheared=sheared, pca=pca)
    print("wrote mesh_post_reference.npz:", exported["v"].shape, exported["f"].shape)


if __name__ == "__main__":
    main()
