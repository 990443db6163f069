#!/usr/bin/env python
"""Benchmark of the DrawingSpinUp hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = ONE drawing through the hot path on each rank (weak scaling: one drawing per GPU
per step, BASELINE.json configs[4] per-GPU = the configuration the metric is quoted on):
    6-view x 2-domain diffusion (75 DDIM steps, UNet B=12, VAE encode/decode, CLIP embed)
 -> Instant-NSR reconstruction (3000 optimisation steps, 128^3 occupancy grid, 2 x 512^3 SDF export)
 -> 24-frame 512x512 stylisation (stage-1 GeneratorJ_RIC + stage-2 GeneratorJ per frame)
on synthetic 512x512 drawings and random-init weights.  Not inside the timed region (stated in
config.workload): FFC-ResNet contour removal (BASELINE configs[0]: CPU plumbing), CPU marching
cubes / mesh post-processing, Blender rendering, PNG I/O.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mv-steps", type=int, default=75)
    ap.add_argument("--nsr-steps", type=int, default=3000)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of ONE kernel family on the stream it is launched on (torch's current
    stream — libdsu_hip launches there).  Wraps ops.sdf_fd_bwd."""

    def __init__(self):
        self.events, self.points, self.levels = [], [], []
        self.enabled = False

    def install(self):
        from drawingspinup_amd import ops
        orig = ops.sdf_fd_bwd
        timer = self

        def timed(cfg, table, mlp, pts, radius, eps, active, *a, **k):
            if not timer.enabled:
                return orig(cfg, table, mlp, pts, radius, eps, active, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig(cfg, table, mlp, pts, radius, eps, active, *a, **k)
            e.record()
            timer.events.append((s, e))
            timer.points.append(pts.shape[0])
            timer.levels.append(int(active))
            return out
        ops.sdf_fd_bwd = timed
        import drawingspinup_amd.nsr.model as m
        m.ops = ops

    def summary(self):
        if not self.events:
            return None
        ms = [s.elapsed_time(e) for s, e in self.events]
        # algorithmic bytes per launch: per point 7 evaluations x
        # active_levels x 8 corners x (4 B f16x2 table read + 8 B f32x2 gradient scatter)
        # + 12 B position + 72 B upstream gradients
        bytes_ = [n * (7 * l * 8 * (4 + 8) + 12 + 72) for n, l in zip(self.points, self.levels)]
        avg_ms = sum(ms) / len(ms)
        avg_bytes = sum(bytes_) / len(bytes_)
        return {"launches": len(ms), "avg_ms": avg_ms, "avg_bytes": avg_bytes,
                "gbps": avg_bytes / (avg_ms * 1e-3) / 1e9}


def cpu_baseline(nsr_steps, frames, mv_steps):
    """Bounded CPU run of the oracle restatements, extrapolated to one drawing."""
    import numpy as np
    from oracle import hashgrid as oh, mv_ref as mr, style_ref as sr
    torch.set_num_threads(os.cpu_count() or 1)
    cores = torch.get_num_threads()
    lv = oh.make_levels()
    g = torch.Generator().manual_seed(0)
    tab = ((torch.rand(lv["offsets"][10], 2, generator=g) * 2 - 1) * 0.1).half().numpy()
    mlp = [np.random.default_rng(0).normal(size=s) * 0.2 for s in [(64, 23), (64,), (13, 64), (13,)]]
    pts = (np.random.default_rng(1).random((20000, 3)) * 2 - 1).astype(np.float32)
    t = time.time()
    oh.sdf_fd(tab, mlp, pts, 1.0, 0.02, lv, 4)
    t_eval = (time.time() - t) / (20000 * 7)                      # s per network evaluation (fwd)
    # one NSR step = ~1.86 M evaluations forward; backward costs ~2x forward
    t_nsr = nsr_steps * 1.86e6 * 3 * t_eval + 2 * 512 ** 3 * t_eval
    # stylisation: one stage-2-shaped 7x7 166->64 conv on a 128x128 crop, scaled by FLOPs
    x, w = torch.randn(1, 166, 128, 128), torch.randn(64, 166, 7, 7)
    t = time.time()
    sr.conv_bn_act(x, w, None, 1, 3)
    t_conv = time.time() - t
    gmac_crop = 166 * 49 * 64 * 128 * 128 / 1e9
    t_style = frames * (148.8 + 271.9) / gmac_crop * t_conv
    # diffusion: one level-0 multi-view attention call of the oracle (B=12,N=256 reduced), scaled
    q = torch.randn(12, 256, 320, dtype=torch.float64)
    t = time.time()
    mr.mv_attention_core(q, q, q, 8, 6)
    t_att = time.time() - t
    gmac_att = 12 * 8 * 256 * (6 * 256) * 40 * 2 / 1e9
    t_mv = mv_steps * 1456.3 / gmac_att * t_att
    total = t_nsr + t_style + t_mv
    return {"value": 1.0 / total, "unit": "drawings/s", "cores": cores, "kind": "port",
            "sample": ("numpy/torch oracle on host cores, extrapolated by work: NSR from 140k "
                       "hash-grid+MLP evaluations (%.2e s/eval, x3 for fwd+bwd, %d steps + 2x512^3 "
                       "export = %.0f s); stylisation from one 7x7 166->64 conv on a 128^2 crop "
                       "(%.0f s / %d frames); diffusion from one level-0 multi-view attention "
                       "(%.0f s / %d steps)" % (t_eval, nsr_steps, t_nsr, t_style, frames, t_mv,
                                                mv_steps))}


def main():
    args = parse()
    from drawingspinup_amd import dist as ddist
    rank, world, local = ddist.init()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using {world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from drawingspinup_amd.drawing import DrawingPipeline
    timer = KernelTimer()
    timer.install()
    pipe = DrawingPipeline(dev, seed=0, mv_steps=args.mv_steps, nsr_steps=args.nsr_steps,
                           n_frames=args.frames)
    # shared read-only weights: RCCL broadcast from rank 0 over xGMI, once
    bcast_bytes = sum(ddist.broadcast_module(m, 0) for m in pipe.shared_modules())

    stage_t = {"mv": 0.0, "nsr": 0.0, "style": 0.0}

    def one_drawing(seed, timed):
        from drawingspinup_amd.drawing import synthetic_drawing, synthetic_frames
        drawing = synthetic_drawing(seed, device=dev)
        torch.cuda.synchronize(); t0 = time.time()
        normals, colors = pipe.multiview(drawing, 123456 + seed)
        torch.cuda.synchronize(); t1 = time.time()
        system, inside = pipe.reconstruct(normals, colors, drawing, 123456 + seed)
        torch.cuda.synchronize(); t2 = time.time()
        frames = pipe.stylize(synthetic_frames(seed, args.frames, device=dev))
        torch.cuda.synchronize(); t3 = time.time()
        if timed:
            stage_t["mv"] += t1 - t0; stage_t["nsr"] += t2 - t1; stage_t["style"] += t3 - t2
        return colors, inside.sum(), frames

    for w in range(args.warmup):
        one_drawing(1000 + rank * 100 + w, False)
    ddist.barrier(); torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.time()
    last = None
    for s in range(args.steps):
        last = one_drawing(rank * 100 + s, True)
    torch.cuda.synchronize(); ddist.barrier()
    elapsed = ddist.max_over_ranks(time.time() - t0, dev)
    timer.enabled = False
    # per-rank gather of the (small) outputs only
    ddist.gather_tensor(last[1].reshape(1).float())

    if rank == 0:
        drawings = world * args.steps
        ks = timer.summary()
        roof = None
        if ks:
            roof = {"bound": "hbm", "kernel": "sdf_fd_bwd_mfma_kernel", "achieved": ks["gbps"],
                    "peak": 8000.0, "unit": "GB/s", "frac": ks["gbps"] / 8000.0,
                    # PMC passes are separate runs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): the
                    # per-launch HBM-side bytes of this kernel at the first schedule stage
                    "traffic": 5.40e8,
                    "traffic_pmc_reference": "profiles/round1_pmc_sdf_kernels.txt: 540 MB HBM-side "
                                             "(FETCH 36 MB + WRITE 504 MB) per launch at N=262144, "
                                             "4 levels vs 726 MB algorithmic for that launch",
                    "launches": ks["launches"], "avg_launch_ms": ks["avg_ms"],
                    "alg_bytes_per_launch": ks["avg_bytes"],
                    # per-stage view (SURVEY.md §8d algorithmic FLOPs / wall time of the stage):
                    # diffusion 2.913 TFLOP per UNet forward at B=12 (f16 MFMA peak 2.5 PFLOP/s),
                    # stylisation 0.30 + 0.54 TFLOP per frame (f32 MFMA peak 157 TFLOP/s)
                    "stages": {
                        "mv": {"bound": "mfma", "unit": "TFLOP/s", "peak": 2500.0,
                               "achieved": 2.913 * args.mv_steps / max(stage_t["mv"] / args.steps, 1e-9)},
                        "style": {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.0,
                                  "achieved": 0.84 * args.frames / max(stage_t["style"] / args.steps, 1e-9)},
                    }}
            for st in roof["stages"].values():
                st["frac"] = st["achieved"] / st["peak"]
        out = {
            "metric": "drawings/sec end-to-end (512x512, 6 views, 24 frames)",
            "value": drawings / elapsed, "unit": "drawings/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (diffusion) / f16 table + f32 MLP (NSR) / f32 (stylisation)",
            "data": "synthetic",
            "config": {"workload": "one drawing per GPU: 6-view diffusion (%d DDIM steps, B=12) -> "
                                   "NSR recon (%d steps, 2x512^3 export) -> %d-frame stage1+stage2 "
                                   "stylisation; NOT timed: FFC-ResNet contour removal, CPU mesh "
                                   "post-processing, Blender, PNG I/O"
                                   % (args.mv_steps, args.nsr_steps, args.frames),
                       "drawings_per_step": world, "parallelism": f"replica-per-drawing x{world}",
                       "stage_seconds_rank0": {k: v / args.steps for k, v in stage_t.items()},
                       "weights_broadcast_bytes": bcast_bytes},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.nsr_steps, args.frames, args.mv_steps)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
