"""CPU: the C-ABI library loads and exports every symbol include/dsu_hip.h declares (no
compute calls: there is no GPU here), and the product path fails loudly without a device."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dsu_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from drawingspinup_amd import _lib
    h = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/dsu_hip.h but not exported"
    # and the ctypes prototype table covers the header (a new function cannot be forgotten)
    missing = [n for n in names if n not in _lib._PROTOS and n != "dsu_strerror"]
    assert not missing, missing
    lib = _lib.lib()
    assert lib.dsu_abi_version() >= 1
    assert lib.dsu_strerror(-1).decode() == "invalid argument"


def test_host_only_entry_point_matches_oracle():
    from drawingspinup_amd import ops
    from oracle import hashgrid as oh
    for kw in (dict(), dict(n_levels=12), dict(n_levels=16, base_resolution=16, per_level_scale=1.5)):
        cfg = ops.HashGridConfig(**kw)
        lv = cfg.levels()
        ref = oh.make_levels(cfg.n_levels, cfg.log2_hashmap_size, cfg.base_resolution,
                             cfg.per_level_scale)
        assert lv["offsets"] == ref["offsets"] and lv["resolution"] == ref["resolution"]
        assert lv["hashed"] == ref["hashed"] and lv["scale"] == [float(s) for s in ref["scale"]]


def test_argument_validation_without_gpu():
    from drawingspinup_amd import _lib
    lib = _lib.lib()
    cfg = _lib.HashGridCfg(10, 3, 19, 32, 1.3195079107728942)          # 3 features: unsupported
    lv = _lib.HashGridLevels()
    assert lib.dsu_hashgrid_make_levels(ctypes.byref(cfg), ctypes.byref(lv)) == -3
    cfg = _lib.HashGridCfg(0, 2, 19, 32, 1.3)
    assert lib.dsu_hashgrid_make_levels(ctypes.byref(cfg), ctypes.byref(lv)) == -1
    # NULL pointers are rejected before any launch
    cfg = _lib.HashGridCfg(10, 2, 19, 32, 1.3195079107728942)
    assert lib.dsu_hashgrid_encode_fwd(ctypes.byref(cfg), None, None, 5, 4, None, None) == -1
    assert lib.dsu_ric_offsets(0, 4, None, None) == -1


def test_process_wide_knobs_validate_their_ranges():
    """include/dsu_hip.h: the grid caps of the NSR step's kernels and the priority of the step driver's side stream
    (1 high = the default, 2 normal, 0 low) — all restored to their defaults here."""
    from drawingspinup_amd import _lib
    lib = _lib.lib()
    assert lib.dsu_set_onewave_grid_cap(257) == -1 and lib.dsu_set_onewave_grid_cap(-1) == -1
    assert lib.dsu_set_onewave_grid_cap(192) == 0 and lib.dsu_set_onewave_grid_cap(0) == 0
    assert lib.dsu_set_scatter_grid_cap(4097) == -1 and lib.dsu_set_scatter_grid_cap(0) == 0
    for level, rc in ((-1, -1), (3, -1), (0, 0), (2, 0), (1, 0)):
        assert lib.dsu_set_nsr_side_stream_priority(level) == rc, level
    assert lib.dsu_set_nsr_side_stream_pooling(2) == -1 and lib.dsu_set_nsr_side_stream_pooling(1) == 0
    assert lib.dsu_set_nsr_side_stream_pooling(0) == 0


def test_style_training_host_side_sizes():
    """Host-only planning of the training kernels: workspace of the sliced weight gradient and
    the sampling-table size; argument checks return before any launch."""
    from drawingspinup_amd import _lib
    lib = _lib.lib()
    assert lib.dsu_deform_tap_table_bytes(32, 32) == 32 * 32 * 9 * 32
    assert lib.dsu_deform_tap_table_bytes(0, 32) == 0
    # resnet level of the shipped config: 40 x 128 x 8 x 8, 128 -> 128 deformable: 19 column
    # tiles of 7 channels, 20 chunks of 128 pixels -> 10 slices of 2 chunks
    per_slice = 128 * 128 * 9 * 4
    assert lib.dsu_conv2d_wgrad_workspace_bytes(1, 40, 128, 128, 8, 8, 3) == 10 * per_slice
    # upconv1 (192 -> 128 at 32x32): 28 tiles -> 18 slices = 504 workgroups, two full rounds
    assert lib.dsu_conv2d_wgrad_workspace_bytes(1, 40, 192, 128, 32, 32, 3) == \
        18 * 128 * 192 * 9 * 4
    # plain 7x7: flattened (channel, tap) columns in tiles of 64
    n = lib.dsu_conv2d_wgrad_workspace_bytes(0, 40, 166, 64, 32, 32, 7)
    assert n > 0 and n % (64 * 166 * 49 * 4) == 0
    assert lib.dsu_conv2d_wgrad_workspace_bytes(0, 0, 3, 3, 8, 8, 3) == 0
    assert lib.dsu_conv2d_wgrad(None, None, None, 1, 1, 8, 8, 1, 3, 1, 1, None, None, 0, None) == -1
    cfg = _lib.NormCfg(4, 8, 64, 0, 7, 1, 1e-5, 0.1)                     # act 7: invalid
    assert lib.dsu_norm_train_fwd(ctypes.byref(cfg), None, None, None, None, None, None, None,
                                  None, None) == -1
    assert lib.dsu_maxpool2_fwd(None, None, 1, 8, 8, None) == -1
    assert lib.dsu_pair_loss(None, None, 0.0, 4, 0, 1.0, None, None, None) == -1


def test_variant_library_override(tmp_path):
    """DSU_HIP_LIB points a process at a variant library (A/B measurements of kernel builds)."""
    import shutil
    import subprocess
    import sys
    from drawingspinup_amd import _lib
    alt = tmp_path / "libdsu_hip_alt.so"
    shutil.copy(_lib.LIB_PATH, alt)
    code = ("from drawingspinup_amd import _lib; assert _lib.lib().dsu_abi_version() >= 1; "
            "print(_lib.LIB_PATH)")
    env = dict(os.environ, DSU_HIP_LIB=str(alt), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().endswith("libdsu_hip_alt.so")
    env["DSU_HIP_LIB"] = str(tmp_path / "missing.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode != 0 and "missing" in out.stderr       # fails loudly, no fallback


def test_no_cpu_fallback():
    from drawingspinup_amd import ops, _lib
    cfg = ops.HashGridConfig()
    with pytest.raises(_lib.DsuError):
        ops.hashgrid_encode_fwd(cfg, torch.zeros(cfg.n_entries, 2, dtype=torch.float16),
                                torch.zeros(4, 3), 4)
    with pytest.raises(_lib.DsuError):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under drawingspinup_amd/ may import it."""
    pkg = os.path.join(ROOT, "drawingspinup_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_round2_entry_points_validate_before_launching():
    """Host-side planning and argument checks of the round-2 entry points (no GPU here: every call
    must return before a launch)."""
    from drawingspinup_amd import _lib
    lib = _lib.lib()
    # Morton sort: counters (8^bits) + block sums + keys / ranks
    assert lib.dsu_spatial_sort_workspace_bytes(1000, 6) == (262144 + 256 + 2 * 1000) * 4
    assert lib.dsu_spatial_sort_workspace_bytes(0, 4) == (4096 + 4) * 4
    assert lib.dsu_spatial_sort_workspace_bytes(10, 3) == -1 and lib.dsu_spatial_sort_workspace_bytes(10, 8) == -1
    assert lib.dsu_spatial_sort(None, 5, 1.0, 6, None, None, None, 0, None) == -1
    assert lib.dsu_spatial_sort(None, 0, 1.0, 6, None, None, None, 0, None) == 0          # empty: nothing to do
    assert lib.dsu_spatial_sort_dev(None, 5, None, 0, 1.0, 6, None, None, None, 0, None) == -1
    assert lib.dsu_points_tail(None, 0, None, None, None, 0, 0.01, None) == -1
    # optimizers
    assert lib.dsu_table_adamw(None, None, None, None, None, 6, 1e-3, 0.9, 0.99, 1e-15, 0.01, 0.1, 0.1, None) == -1  # n % 4
    assert lib.dsu_table_adamw(None, None, None, None, None, 0, 1e-3, 0.9, 0.99, 1e-15, 0.01, 0.1, 0.1, None) == 0
    assert lib.dsu_table_decay(None, None, 8, 0.5, None) == -1
    assert lib.dsu_adamw_multi(None, 3, 0.9, 0.99, 1e-15, 0.01, None) == -1
    assert lib.dsu_adamw_multi(None, 0, 0.9, 0.99, 1e-15, 0.01, None) == 0
    assert lib.dsu_adamw_multi(None, _lib.ADAMW_MAX_TENSORS + 1, 0.9, 0.99, 1e-15, 0.01, None) == -1
    # export smoothing
    assert lib.dsu_smooth_energy_partials() == 1024
    assert lib.dsu_smooth_iterate(None, 5, None, None, 0.5, 10, None, None, None) == -1
    assert lib.dsu_smooth_iterate(None, 0, None, None, 0.5, 10, None, None, None) == 0
    # TELEA: host function, runs here
    import numpy as np
    img = np.full((5, 6, 3), 9, np.uint8)
    out = np.zeros_like(img)
    m = np.zeros((5, 6), np.uint8)
    m[2, 3] = 1
    P = ctypes.c_void_p
    assert lib.dsu_inpaint_telea_u8c3(P(img.ctypes.data), P(m.ctypes.data), 5, 6, 3, P(out.ctypes.data)) == 0
    # a one-pixel hole in a constant image: Ia / s + 0.5 = 9.5, rounded to even = 10 (OpenCV's
    # formula adds 0.5 AND rounds); everything else untouched
    assert out[2, 3].tolist() == [10, 10, 10]
    out[2, 3] = 9
    assert (out == 9).all()
    assert lib.dsu_inpaint_telea_u8c3(P(img.ctypes.data), P(m.ctypes.data), 2, 6, 3, P(out.ctypes.data)) == -3


def test_texture_partial_map_is_a_bijection_onto_the_gradient_block():
    """Host-only: every parameter of the texture MLP receives exactly one element of the partial
    vector, the padding elements none."""
    import numpy as np
    from drawingspinup_amd import ops
    m = ops.texture_partial_map()
    n_tex = 64 * 16 + 64 + 64 * 64 + 64 + 3 * 64 + 3
    dst = m[m >= 0]
    assert dst.size == n_tex and np.array_equal(np.sort(dst), np.arange(n_tex))
    assert (m < 0).sum() == m.size - n_tex


def test_occgrid_refresh_host_side():
    """Workspace planning and argument checks of the native occupancy refresh (no launch)."""
    import ctypes as C
    from drawingspinup_amd import _lib
    lib = _lib.lib()
    n = 128 ** 3
    need = lib.dsu_occgrid_refresh_workspace_bytes(128)
    # iota + occupied + cells (int32), points (3 f32) + sdf (f32): 7 words per cell at least
    assert 7 * 4 * n <= need < 9 * 4 * n
    assert lib.dsu_occgrid_refresh_workspace_bytes(0) == -1
    assert lib.dsu_occgrid_refresh_workspace_bytes(2048) == -1          # 2^33 cells
    a = _lib.OccRefreshArgs()
    assert lib.dsu_occgrid_refresh(C.byref(a), None) == -1              # NULL pointers
    assert lib.dsu_occgrid_refresh(None, None) == -1
    assert lib.dsu_nsr_driver_occ_refresh(None, C.byref(a), None) == -1
