"""Bit-reproducible synthetic parameters for the mvdiffusion parity fixtures.
TEST INFRASTRUCTURE ONLY (used by tests/golden/make_mv_reference_golden.py and the tests that
read its fixture; never imported by the product).

A reduced-width UNet still has tens of millions of parameters, too many to commit, so the fixture
stores only the (name, shape) list of the REFERENCE's state_dict and every value is regenerated
from the name with integer arithmetic (splitmix64 over the element index, seeded by the CRC32 of
the name): no dependence on any library's random stream.  Values are rounded to float16 so that
the float64 reference run and the f16 device model share them exactly."""
import zlib

import numpy as np
import torch

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def det_uniform(key, n):
    """n float64 values in [-1, 1), a pure function of (key, index)."""
    seed = np.uint64(zlib.crc32(key.encode()))
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * _G + seed * _M1
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (2.0 ** -52) - 1.0


def det_tensor(key, shape, scale=1.0, offset=0.0, dtype=torch.float64):
    n = int(np.prod(shape)) if len(shape) else 1
    v = det_uniform(key, n) * scale + offset
    return torch.from_numpy(v.reshape(tuple(shape))).to(dtype)


def synth_state_dict(names_shapes, f16_round=True):
    """names_shapes: iterable of (name, shape).  Matrices / conv kernels: uniform with variance
    1/fan_in; norm scales 1 + 0.1 u; every other vector 0.05 u.  (The reference zero-initialises
    the joint attention's to_out, transformer_mv2d.py:499,516 — overwritten here on purpose, a
    zero projection would hide that branch.)"""
    sd = {}
    for name, shape in names_shapes:
        shape = tuple(int(s) for s in shape)
        if len(shape) > 1:
            fan_in = int(np.prod(shape[1:]))
            t = det_tensor(name, shape, scale=(3.0 / fan_in) ** 0.5)
        elif "norm" in name and name.endswith("weight"):
            t = det_tensor(name, shape, scale=0.1, offset=1.0)
        else:
            t = det_tensor(name, shape, scale=0.05)
        sd[name] = t.half().double() if f16_round else t
    return sd
