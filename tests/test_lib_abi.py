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
