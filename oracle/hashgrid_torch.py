"""Multi-threaded CPU form of the geometry network's training work (hash grid + SDF MLP, 7-point
finite differences, forward AND backward through torch autograd) for bench.py's `cpu_baseline`.
TEST INFRASTRUCTURE ONLY — the product never imports it.

Same algorithm as oracle/hashgrid.py (tcnn HashGrid as called at instant_nsr/models/
network_utils.py:46,55; VolumeSDF.forward, models/geometry.py:135-187), written with the torch CPU
operators a CPU port of the reference would use — `index_select` gathers from an f32 table,
`F.linear`, `F.softplus(beta=100)` — so that `torch.set_num_threads(cores)` applies
("restatement on torch CPU with set_num_threads").  Arithmetic is f32 (the numpy
oracle reproduces tcnn's f16 FMA chain bit for bit; this one is a throughput baseline and agrees
with it to f16 rounding: tests/test_oracle_nsr.py)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import hashgrid as oh

_PRIMES = (1, 2654435761, 805459861)


def encode(table, x, lv, active_levels):
    """table (entries, 2) f32 (requires_grad for the backward), x (N,3) in [0,1] -> (N, 2*active)."""
    outs = []
    for l in range(active_levels):
        scale = float(lv["scale"][l])
        res = int(lv["resolution"][l])
        off, size = int(lv["offsets"][l]), int(lv["offsets"][l + 1] - lv["offsets"][l])
        pos = x * scale + 0.5
        cell = torch.floor(pos)
        w1 = pos - cell
        cell = cell.to(torch.int64)
        acc = 0
        for c in range(8):
            d = [(c >> k) & 1 for k in range(3)]
            cc = [cell[:, k] + d[k] for k in range(3)]
            if lv["hashed"][l]:
                idx = (cc[0] * _PRIMES[0]) ^ (cc[1] * _PRIMES[1]) ^ (cc[2] * _PRIMES[2])
                idx = (idx & 0xFFFFFFFF) % size
            else:
                # tcnn's stride rule for dense levels: x + y * res + z * res^2, wrapped to the level
                idx = (cc[0] + cc[1] * res + cc[2] * res * res) % size
            w = 1.0
            for k in range(3):
                w = w * (w1[:, k] if d[k] else (1.0 - w1[:, k]))
            acc = acc + torch.index_select(table, 0, idx + off) * w[:, None]
        outs.append(acc)
    return torch.cat(outs, -1)


def sdf_network(table, mlp, pts, radius, lv, active_levels, n_levels=10):
    xc = (pts + radius) / (2 * radius)
    enc = encode(table, xc, lv, active_levels)
    pad = torch.zeros(pts.shape[0], 2 * (n_levels - active_levels))
    h = torch.cat([xc * 2 - 1, enc, pad], -1)
    h = F.softplus(F.linear(h, mlp[0], mlp[1]), beta=100)
    return F.linear(h, mlp[2], mlp[3])


def sdf_fd(table, mlp, pts, radius, eps, lv, active_levels):
    """sdf, FD gradient, feature, laplacian from 7 evaluations (geometry.py:158-176)."""
    offs = torch.tensor([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1],
                         [0, 0, -1]], dtype=pts.dtype) * eps
    allp = (pts[None] + offs[:, None]).clamp(-radius, radius).reshape(-1, 3)
    out = sdf_network(table, mlp, allp, radius, lv, active_levels).view(7, pts.shape[0], -1)
    sdf = out[:, :, 0]
    grad = torch.stack([0.5 * (sdf[1] - sdf[2]), 0.5 * (sdf[3] - sdf[4]), 0.5 * (sdf[5] - sdf[6])], -1) / eps
    lap = (sdf[1:].sum(0) - 6 * sdf[0]) / (eps * eps)
    return sdf[0], grad, out[0], lap


def training_work_seconds(n_points, active_levels=5, threads=None, seed=0):
    """Wall-clock of ONE forward + backward of the 7-evaluation geometry pass for n_points points
    (gradients to the table and the MLP), as a step of the optimisation performs it."""
    import time
    if threads:
        torch.set_num_threads(threads)
    lv = oh.make_levels()
    g = torch.Generator().manual_seed(seed)
    table = ((torch.rand(int(lv["offsets"][10]), 2, generator=g) * 2 - 1) * 0.1).requires_grad_(True)
    mlp = [(torch.randn(*s, generator=g) * 0.2).requires_grad_(True)
           for s in [(64, 23), (64,), (13, 64), (13,)]]
    pts = torch.rand(n_points, 3, generator=g) * 2 - 1
    t = time.time()
    sdf, grad, feat, lap = sdf_fd(table, mlp, pts, 1.0, 0.02, lv, active_levels)
    fwd = time.time() - t
    loss = sdf.sum() + grad.sum() + feat.sum()
    loss.backward()
    return fwd, time.time() - t
