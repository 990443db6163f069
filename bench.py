#!/usr/bin/env python
"""Benchmark of the DrawingSpinUp hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config drawing|unet|nsr50k|frames]

`--config drawing` (default; BASELINE.json's metric, configs[4] per GPU): one "step" = ONE drawing
through the hot path on each rank (weak scaling, one drawing per GPU per step):
    FFC-ResNet contour-removal generator + masks (512x512)
 -> 6-view x 2-domain diffusion (75 DDIM steps, UNet B=12, VAE encode/decode, CLIP embed)
 -> Instant-NSR reconstruction (3000 optimisation steps, 128^3 occupancy grid, 2 x 512^3 SDF export)
 -> 24-frame 512x512 stylisation (stage-1 GeneratorJ_RIC + stage-2 GeneratorJ per frame)
on synthetic 512x512 drawings and random-init weights.  Inside the timed region as well: the
contour stage's TELEA inpainting tail (host code of the library) and the export's smoothing /
marching cubes (device), the fine stage's quadric remeshing to 50 000 faces (host code of the
library) and save_mesh's smoothing / colour back-projection / shear — the reference YAML's export
switches for a uid outside the thinning list.  Not inside (stated in config.workload): Blender
rendering, PNG / OBJ file I/O.
`--config unet` (BASELINE configs[1]): one step = one forward of the multi-view UNet on the 12-sample
batch of one drawing (6 views x 2 domains, 256x256 images = 32x32 latents, f16, random weights).
`--config nsr50k` (BASELINE configs[2] micro-benchmark): one step = one NSR optimisation
iteration with 50 000 rays marched through a 128^3 occupancy grid (synthetic sphere).
`--config frames` (BASELINE configs[3]): one step = 24 frames through stage 1 + stage 2, the
frames sharded over the ranks (strong scaling).

`--gpus N` with N > 1 launched WITHOUT torch.distributed.run spawns the N ranks itself (one
process per GPU, RCCL), so `python bench.py --gpus 8` and the torchrun form measure the same job.
Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E ~8 TB/s
F16_MFMA_PEAK_TF = 2500.0    # dense f16/bf16
F32_MFMA_PEAK_TF = 157.3     # v_mfma_f32_32x32x2_f32


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="drawing", choices=["drawing", "unet", "nsr50k", "frames"])
    ap.add_argument("--mv-steps", type=int, default=75)
    ap.add_argument("--nsr-steps", type=int, default=3000)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=3,
                    help="--config drawing: drawings in flight per GPU (one Python thread + one stream + "
                         "one DrawingPipeline each); 1 = one drawing at a time, as in rounds 1-5")
    ap.add_argument("--side-priority", type=int, default=None, choices=(0, 1, 2),
                    help="priority of the NSR step driver's side stream: 1 high, 2 normal, 0 low (default: high with "
                         "drawings in flight, normal with one drawing at a time; profiles/round6_side_stream_priority.txt)")
    ap.add_argument("--side-pool", type=int, default=0, choices=(0, 1),
                    help="1: step drivers hand their side stream on to the next driver (one stream per drawing in flight "
                         "instead of one per drawing; not measured yet, default off)")
    ap.add_argument("--inflight-skew", type=float, default=None,
                    help="seconds between the starts of the workers' first drawings of a region (stage skew; "
                         "default: 4.2 s / inflight)")
    ap.add_argument("--onewave-grid", type=int, default=None,
                    help="workgroups of the NSR step's two one-wave-per-SIMD kernels (dsu_set_onewave_grid_cap); "
                         "default: 256 with one drawing at a time, 192 with several in flight")
    ap.add_argument("--scatter-grid", type=int, default=None,
                    help="workgroups of the geometry backward's scatter kernel (dsu_set_scatter_grid_cap); default 256")
    ap.add_argument("--fit-priority", type=int, default=0,
                    help="1: with drawings in flight, the NSR optimisation runs on a low-priority stream of its own "
                         "and the other stages on a high-priority stream")
    ap.add_argument("--nsr-slots", type=int, default=0,
                    help="at most this many drawings inside the NSR optimisation at a time (0 = no limit)")
    a = ap.parse_args(argv)
    if a.steps is None:
        a.steps = {"drawing": 2, "unet": 50, "nsr50k": 50, "frames": 3}[a.config]
    if a.warmup is None:
        a.warmup = {"drawing": 1, "unet": 5, "nsr50k": 10, "frames": 1}[a.config]
    return a


# ------------------------------------------------------------------------------------------------
# HIP-event timing of kernel families on the stream they are launched on (torch's current stream —
# libdsu_hip launches there).  Each family carries its algorithmic work per launch.
# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """The NSR step is sequenced inside the library (dsu_nsr_driver_step), so its two geometry
    families are timed there, with HIP events on the stream the kernels run on
    (dsu_nsr_driver_timing); the families launched from Python are wrapped here."""

    # Every event pair costs main-queue time (measured: the four records of an NSR step 28 us of its
    # 1.05 ms; ~40 000 pairs around the convolutions / attentions of one drawing's diffusion), so the
    # families are SAMPLED: every STRIDE-th call (step) is timed, the averages are over those.
    STRIDE = 7

    def __init__(self, stride=None):
        self._enabled = False
        self.fam = {}
        self.stride = int(stride or self.STRIDE)
        self._calls = {}
        import random
        self._rng = random.Random(0x5eed)

    @property
    def enabled(self):
        return self._enabled

    @enabled.setter
    def enabled(self, on):
        from drawingspinup_amd.nsr import system as nsr_system
        self._enabled = bool(on)
        nsr_system.native_timing["enabled"] = bool(on)
        nsr_system.native_timing["stride"] = self.stride

    def _wrap(self, module, name, family, work):
        orig = getattr(module, name)
        timer = self

        def timed(*a, **k):
            if not timer.enabled:
                return orig(*a, **k)
            # sampled by a seeded draw per call, not by call count: the diffusion UNet issues a
            # periodic sequence of convolution / attention shapes, and a count-based stride that
            # shares a factor with the calls per forward would time the same layers every step
            timer._calls[family] = timer._calls.get(family, 0) + 1
            if timer._rng.random() * timer.stride >= 1.0:
                return orig(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig(*a, **k)
            e.record()
            f = timer.fam.setdefault(family, {"events": [], "work": 0.0})
            f["events"].append((s, e))
            f["work"] += work(*a, **k)
            return out
        setattr(module, name, timed)

    def install(self):
        from drawingspinup_amd import ops
        # hash grid + SDF MLP (N1-N5): algorithmic bytes per point = 7 evaluations x active levels
        # x 8 corners x 4 B (f16 x 2) [+ 12 B position, + outputs / upstream gradients]
        self._wrap(ops, "sdf_fd_bwd", "sdf_fd_bwd",
                   lambda cfg, tab, mlp, pts, radius, eps, active, *a, **k:
                   pts.shape[0] * (7 * int(active) * 8 * 4 + 12 + 72))
        self._wrap(ops, "sdf_fd_fwd", "sdf_fd_fwd",
                   lambda cfg, tab, mlp, pts, radius, eps, active, *a, **k:
                   pts.shape[0] * (7 * int(active) * 8 * 4 + 12 + 72))

        def conv_flops(x, w, bias=None, k=3, stride=1, pad=1, upsample2x=False, *a, **kw):
            B, H, W, Cin = x.shape
            IH, IW = (2 * H, 2 * W) if upsample2x else (H, W)
            OH, OW = (IH + 2 * pad - k) // stride + 1, (IW + 2 * pad - k) // stride + 1
            return 2.0 * B * OH * OW * w.shape[0] * Cin * k * k
        self._wrap(ops, "conv2d_nhwc_f16", "conv_f16", conv_flops)

        def attn_flops(q, k, vt, seg, heads, seg_len, scale=None):
            return 4.0 * q.shape[0] * q.shape[1] * q.shape[2] * seg.shape[1] * seg_len
        self._wrap(ops, "mv_attention", "mv_attention", attn_flops)
        # the python modules captured `ops.<name>` at call time through the module attribute, so
        # re-binding the attribute on `ops` is enough

    def summary(self):
        from drawingspinup_amd.nsr import system as nsr_system
        rows = []
        merged = {name: [len(f["events"]), sum(s.elapsed_time(e) for s, e in f["events"]), f["work"]]
                  for name, f in self.fam.items()}
        flops = {}
        for name, tot in nsr_system.native_timing["totals"].items():
            n, ms, work = tot[:3]
            t = merged.setdefault(name, [0, 0.0, 0.0])
            t[0] += n; t[1] += ms; t[2] += work
            if len(tot) > 3:
                flops[name] = flops.get(name, 0.0) + tot[3]
        for name, (n, ms, work) in merged.items():
            f = {"work": work}
            if not n or ms <= 0:
                continue
            hbm = name.startswith("sdf_")
            ach = f["work"] / (ms * 1e-3) / (1e9 if hbm else 1e12)
            peak = HBM_PEAK_GBS if hbm else F16_MFMA_PEAK_TF
            calls = self._calls.get(name)
            rows.append({"kernel": name, "bound": "hbm" if hbm else "mfma", "launches": n,
                         "sampled_call_fraction": (n / calls) if calls else 1.0 / self.stride,
                         "total_ms": ms, "avg_launch_ms": ms / n,
                         "alg_work_per_launch": f["work"] / n, "achieved": ach, "peak": peak,
                         "unit": "GB/s" if hbm else "TFLOP/s", "frac": ach / peak})
            if name in flops and flops[name] > 0:
                # what the geometry kernels are actually bound by: the f32 arithmetic of the 64-wide
                # MLP (f32 MFMA / VALU FMA peak 157.3 TFLOP/s), at one wave per SIMD in the backward
                # pair (+ 64-bit LDS atomics in its scatter half), not the table traffic
                tf = flops[name] / (ms * 1e-3) / 1e12
                rows[-1]["bound_actual"] = {
                    "bound": "f32 MLP arithmetic (MFMA f32 / bf16 x 3 with f32 accumulation, VALU f32; priced at the f32 MFMA peak)",
                    "achieved": tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": tf / F32_MFMA_PEAK_TF,
                    "note": ("MLP part at one wave per SIMD (460-480 VGPRs) + scatter on 64-bit LDS atomics"
                             if name == "sdf_fd_bwd" else
                             "level-outer gathers (shared corners) + VALU MLP, 3-4 waves per SIMD")}
        rows.sort(key=lambda r: -r["total_ms"])
        return rows


# ------------------------------------------------------------------------------------------------
# CPU baseline: the same operators on the GPU box's host cores, bounded samples.
# ------------------------------------------------------------------------------------------------
def cpu_baseline(nsr_steps, frames, mv_steps):
    """kind "port": /root/reference does not exist on the GPU box and the reference has no CPU
    path for its CUDA-only ops, so the baseline runs (a) this repository's restatements that are
    pinned to the reference's own classes by fixtures and execute the SAME torch CPU operators
    the reference's modules would (FFC-ResNet generator, GeneratorJ: nn.Conv2d / BatchNorm /
    activations), (b) the UNet oracle run in float32, (c) the torch-CPU form of the hash-grid oracle
    (all `cores` threads).  Each leg is a bounded sample, extrapolated by the stated factor."""
    from oracle import mv_ref as mr, style_net_ref as snr
    # at most 32 threads: on the GPU box's 256 hardware threads torch's CPU convolutions ran 10-90x
    # SLOWER with one thread per hardware thread than with a few dozen (measured: one FFC-ResNet
    # forward 146 s at 256 threads vs 1.6 s at 8)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cores = torch.get_num_threads()
    legs = {}
    # (a1) contour remover: one 512x512 forward of the FFC-ResNet generator (27 M parameters)
    from drawingspinup_amd.contour.ffc import LAMA_FOURIER_GENERATOR, make_generator
    torch.manual_seed(0)
    gen = make_generator(**LAMA_FOURIER_GENERATOR).eval()
    x = torch.rand(1, 4, 512, 512)
    with torch.no_grad():
        t = time.time(); gen(x); legs["contour_s"] = time.time() - t
    # (a2) stylisation: one 512x512 GeneratorJ frame (272 GMAC); the stage-1 GeneratorJ_RIC
    # (149 GMAC, deformable) is charged at the same rate per MAC
    from drawingspinup_amd.drawing import STYLE_ARGS
    from drawingspinup_amd.style.generators import build_model
    g2 = build_model("GeneratorJ", STYLE_ARGS, "cpu").eval()
    xf = torch.rand(1, 6, 512, 512) * 2 - 1
    with torch.no_grad():
        t = time.time(); snr.generator_j_forward(g2, xf); t_g2 = time.time() - t
    legs["style_s"] = frames * t_g2 * (271.9 + 148.8) / 271.9
    # (b) diffusion: ONE UNet forward of the full-width architecture at the BASELINE shape
    # (12, 8, 32, 32) in float32 through the oracle's functional UNet
    # (torch CPU convolutions / linears / attention), x75 steps (VAE / CLIP not charged)
    from drawingspinup_amd.mv.unet import UNetMV2DConditionModel
    torch.manual_seed(0)
    un = UNetMV2DConditionModel().half()
    ref = mr.UNetRef(un.state_dict(), (320, 640, 1280, 1280),
                     ("CrossAttnDownBlockMV2D",) * 3 + ("DownBlock2D",),
                     ("UpBlock2D",) + ("CrossAttnUpBlockMV2D",) * 3, layers_per_block=2,
                     dtype=torch.float32)
    g = torch.Generator().manual_seed(1)
    sample = torch.randn(12, 8, 32, 32, generator=g)
    t = time.time()
    ref(sample, torch.tensor([500]), torch.randn(12, 1, 768, generator=g), torch.randn(12, 10, generator=g))
    t_unet = time.time() - t
    legs["mv_s"] = mv_steps * t_unet
    del un, ref
    # (c) NSR: the geometry network's share of one optimisation step (7 finite-difference
    # evaluations per point, forward AND backward through autograd) with the multi-threaded
    # torch-CPU restatement (oracle/hashgrid_torch.py: index_select gathers + F.linear, SURVEY.md
    # 8d) on 131 072 points after a warm-up, scaled to the step's 262 144 + 4 096 points; marching,
    # compositing, texture MLP and losses are not charged.  Export: 2 x 512^3 forward-only
    # evaluations.
    from oracle import hashgrid_torch as ht
    n_pts = 131072                                   # half a step's points (round 4: 40 000)
    ht.training_work_seconds(8192, active_levels=5, threads=cores)          # warm-up (thread pools, allocator)
    t_fwd, t_fb = ht.training_work_seconds(n_pts, active_levels=5, threads=cores)
    step_pts = 262144 + 4096
    legs["nsr_s"] = nsr_steps * t_fb * step_pts / n_pts + 2 * 512 ** 3 * t_fwd / (7 * n_pts)
    total = sum(legs.values())
    return {"value": 1.0 / total, "unit": "drawings/s", "cores": cores, "kind": "port",
            "seconds_per_drawing": total, "legs_seconds": legs,
            "sample": ("host cores, torch CPU / numpy: one 512^2 FFC-ResNet forward (%.2f s); one 512^2 "
                       "GeneratorJ frame (%.2f s) x %d frames x (272+149)/272 GMAC; one float32 UNet "
                       "forward at (12,8,32,32) through the oracle's functional UNet (%.1f s) x %d steps; "
                       "geometry forward+backward (7 evaluations per point, torch CPU autograd, %d threads) "
                       "of %d points (%.2f s) x 266 240 / %d x %d steps + 2 x 512^3 export evaluations "
                       "at the forward rate"
                       % (legs["contour_s"], t_g2, frames, t_unet, mv_steps, cores, n_pts, t_fb, n_pts,
                          nsr_steps))}


# ------------------------------------------------------------------------------------------------
def run(args):
    from drawingspinup_amd import dist as ddist
    rank, world, local = ddist.init()
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; measuring {world} rank(s)",
              file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    timer = KernelTimer()
    timer.install()
    if args.config == "unet":
        out = bench_unet(args, ddist, rank, world, dev, timer)
    elif args.config == "nsr50k":
        out = bench_nsr50k(args, ddist, rank, world, dev, timer)
    elif args.config == "frames":
        out = bench_frames(args, ddist, rank, world, dev, timer)
    else:
        out = bench_drawing(args, ddist, rank, world, dev, timer)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():        # world > 1, or the forced one-rank group
        torch.distributed.destroy_process_group()


def _sync(dev):
    """device-wide synchronise (a no-op for the CPU tensors of the gloo tests)."""
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize(dev)


def _timed_loop(args, ddist, dev, timer, body):
    for w in range(args.warmup):
        body(w, False)
    ddist.barrier(); _sync(dev)
    timer.enabled = True
    t0 = time.time()
    last = None
    for s in range(args.steps):
        last = body(s, True)
    _sync(dev); ddist.barrier()
    elapsed = ddist.max_over_ranks(time.time() - t0, dev)
    timer.enabled = False
    return elapsed, last


def _inflight_loop(args, ddist, rank, dev, timer, pipes, inputs, stage_t, sub_t, gathered):
    """K = len(pipes) drawings in flight on this rank's GPU: K worker threads, each with its own
    stream and pipeline, take the drawings of the region in turn (drawing j -> worker j % K) and run
    them back to back; the main thread keeps the collectives (the per-drawing gather of the outputs
    to rank 0) in one order on every rank.  Warm-up region, barrier + synchronize, timed region of
    args.steps * K drawings, synchronize + barrier, max over ranks — the contract's brackets around
    the whole region.  Returns (elapsed, {"latency": ..., "alone": ...})."""
    import queue
    import threading
    from drawingspinup_amd.drawing import DrawingPipeline  # noqa: F401  (import check before threads start)
    K = len(pipes)
    on_gpu = torch.device(dev).type == "cuda"

    class _HostStream:                      # the gloo tests drive this loop with CPU stubs
        def synchronize(self):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False
    fitprio = bool(getattr(args, "fit_priority", 0)) and on_gpu
    streams = [torch.cuda.Stream(dev, priority=-1 if fitprio else 0) if on_gpu else _HostStream() for _ in range(K)]
    if fitprio:
        for p_ in pipes:
            p_.fit_stream = torch.cuda.Stream(dev, priority=0)
    lat = []

    def one(pipe, stream, j, timed):
        seed = (rank * 100 + j) if timed else (1000 + rank * 100 + j)
        drawing, frames_in, edges_in = inputs[(timed, j)]
        sync = stream.synchronize
        with (torch.cuda.stream(stream) if on_gpu else stream):
            t0 = time.time()
            cleaned = pipe.remove_contour(drawing)
            sync(); t1 = time.time()
            normals, colors = pipe.multiview(cleaned, 123456 + seed)
            sync(); t2 = time.time()
            system, inside = pipe.reconstruct(normals, colors, cleaned, 123456 + seed)
            sync(); t3 = time.time()
            frames = pipe.stylize(frames_in, edges_in)
            views = torch.cat([normals, colors]).to(torch.float16)
            sync(); t4 = time.time()
        return (t0, t1, t2, t3, t4), views, frames, dict(pipe.substage_seconds)

    grid_cap = getattr(args, "onewave_grid", None)
    grid_cap = 192 if grid_cap is None else int(grid_cap)
    if on_gpu:
        from drawingspinup_amd import _lib as dsu_lib
        dsu_lib.check(dsu_lib.lib().dsu_set_onewave_grid_cap(grid_cap), "dsu_set_onewave_grid_cap")
        sc_cap = getattr(args, "scatter_grid", None)
        dsu_lib.check(dsu_lib.lib().dsu_set_scatter_grid_cap(0 if sc_cap is None else int(sc_cap)),
                      "dsu_set_scatter_grid_cap")
    skew = getattr(args, "inflight_skew", None)
    skew = (4.2 / K if skew is None else float(skew)) if on_gpu else 0.0
    slots = int(getattr(args, "nsr_slots", 0) or 0)
    gate = threading.BoundedSemaphore(slots) if slots > 0 else None
    for p_ in pipes:
        if hasattr(p_, "fit_gate"):
            p_.fit_gate = gate

    def region(n_drawings, timed):
        done = queue.Queue()

        def worker(k):
            try:
                if on_gpu:
                    torch.cuda.set_device(dev)
                if k and skew > 0:
                    time.sleep(skew * k)              # the workers start a fraction of a drawing apart
                for j in range(k, n_drawings, K):
                    done.put((j, one(pipes[k], streams[k], j, timed)))
            except BaseException as e:          # surfaces in the main thread
                done.put((-1, e))
        th = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(K)]
        for t in th:
            t.start()
        for _ in range(n_drawings):
            j, res = done.get()
            if j < 0:
                raise res
            (t0, t1, t2, t3, t4), views, frames, sub = res
            tg = time.time()
            gv = ddist.gather_tensor(views)           # main thread only: one order of collectives per rank
            gf = ddist.gather_tensor(frames)
            if timed:
                for k_, v in zip(stage_t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, time.time() - tg)):
                    stage_t[k_] += v
                for k_ in sub_t:
                    sub_t[k_] += sub.get(k_, 0.0)
                lat.append(t4 - t0)
                if gv is not None:
                    gathered["views"], gathered["frames"] = gv, gf
        for t in th:
            t.join()

    region(args.warmup * K, False)
    ddist.barrier(); _sync(dev)
    timer.enabled = True
    t0 = time.time()
    region(args.steps * K, True)
    _sync(dev); ddist.barrier()
    elapsed = ddist.max_over_ranks(time.time() - t0, dev)
    timer.enabled = False
    info = {"latency": {"mean": sum(lat) / max(len(lat), 1), "max": max(lat) if lat else None,
                        "drawings": len(lat)},
            "schedule": {"start_skew_s": skew, "nsr_slots": slots, "onewave_grid": grid_cap}}
    # one drawing alone, beside the clock: the kernel families without co-running drawings
    if rank == 0 and hasattr(timer, "fam"):
        concurrent = timer.summary()
        saved = (timer.fam, dict(timer._calls))
        from drawingspinup_amd.nsr import system as nsr_system
        saved_tot = nsr_system.native_timing["totals"]
        timer.fam, timer._calls, nsr_system.native_timing["totals"] = {}, {}, {}
        timer.enabled = True
        if on_gpu:
            dsu_lib.lib().dsu_set_onewave_grid_cap(0)          # alone: one workgroup per CU
            dsu_lib.lib().dsu_set_scatter_grid_cap(0)
            # ... and the side stream at normal priority, as for one drawing at a time (DSU_ALONE_SIDE_PRIO overrides):
            # at the in-flight priority this drawing lands in either mode of profiles/round6_side_stream_priority.txt
            dsu_lib.lib().dsu_set_nsr_side_stream_priority(int(os.environ.get("DSU_ALONE_SIDE_PRIO", "2")))
        one(pipes[0], streams[0], 0, True)
        _sync(dev)
        timer.enabled = False
        rows = timer.summary()
        # diagnostic (DSU_ALONE_REPEAT=n): the same drawing alone n more times, the allocator's cache dropped before the
        # last one — tells a per-process placement effect from a per-drawing one (the two modes of `alone`)
        for rep in range(int(os.environ.get("DSU_ALONE_REPEAT", "0"))):
            print("[alone %d] %s" % (rep, [(r["kernel"], round(r["avg_launch_ms"], 4)) for r in rows[:3]]), file=sys.stderr)
            if rep == int(os.environ["DSU_ALONE_REPEAT"]) - 1 and on_gpu:
                torch.cuda.empty_cache()
            timer.fam, timer._calls, nsr_system.native_timing["totals"] = {}, {}, {}
            timer.enabled = True
            one(pipes[0], streams[0], 0, True)
            _sync(dev)
            timer.enabled = False
            rows = timer.summary()
            print("[alone %d'] %s" % (rep, [(r["kernel"], round(r["avg_launch_ms"], 4)) for r in rows[:3]]), file=sys.stderr)
        info["alone"] = [{k: r[k] for k in ("kernel", "launches", "avg_launch_ms", "achieved", "peak", "unit", "frac")
                          if k in r} | ({"bound_actual_frac": r["bound_actual"]["frac"]} if "bound_actual" in r else {})
                         for r in rows[:3]]
        timer.fam, timer._calls = saved
        nsr_system.native_timing["totals"] = saved_tot
        del concurrent
    return elapsed, info


_PMC_CACHE = []


def _pmc():
    """profiles/round6_pmc.json (tools/pmc_round6.sh; the previous rounds. files when it is absent) or None."""
    if not _PMC_CACHE:
        for name in ("round6_pmc.json", "round5_pmc.json", "round4_pmc.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    _PMC_CACHE.append(dict(json.load(f), source="profiles/" + name))
                break
            except (OSError, ValueError):
                continue
        else:
            _PMC_CACHE.append(None)
    return _PMC_CACHE[0]


def _sq_share(prefixes, field):
    """dispatch-weighted mean of an SQ share (profiles/round4_pmc.json) over the kernels whose
    name starts with one of `prefixes`, weighted by their wave cycles."""
    pmc = _pmc()
    if not pmc:
        return None
    num = den = 0.0
    for k, v in pmc["kernels"].items():
        if any(k.startswith(p) for p in prefixes) and v.get("SQ_WAVE_CYCLES") and v.get(field) is not None:
            w = v["SQ_WAVE_CYCLES"] * v.get("dispatches", 1)
            num += v[field] * w
            den += w
    return num / den if den else None


def _roofline(timer, extra=None):
    rows = timer.summary()
    if not rows:
        return None
    top = dict(rows[0])
    top["kernels"] = rows[:3]
    # `launches` / `total_ms` count the TIMED launches: every stride-th call of a family
    top["timed_launch_stride"] = timer.stride
    # HBM-side bytes per launch of the dominant family: not collectable inside this process (PMC
    # counters need their own rocprofv3 --pmc passes, FETCH_SIZE and WRITE_SIZE one pass each, as
    # MI355X_MICROARCH.md prescribes).  profiles/round6_pmc.json holds this round's passes over
    # tools/pmc_sdf_kernels.py (N = 262 144 Morton-ordered samples, 5 levels): per kernel
    # (2 x FETCH_SIZE + WRITE_SIZE) — the guide's gfx950 correction for coalesced reads — and the
    # workload's algorithmic bytes; the measured ratio is applied to this run's mean algorithmic
    # bytes per launch.
    pmc = _pmc()
    fam = {"sdf_fd_bwd": ("sdf_fd_bwd_pipe_kernel", "sdf_fd_bwd_mfma_kernel", "sdf_fd_scatter_kernel"),
           "sdf_fd_fwd": ("sdf_fd_fwd_kernel", "sdf_fd_fwd_shared_kernel")}
    top["traffic"] = None
    if pmc and top["kernel"] in fam:
        side = sum(v.get("hbm_side_bytes", 0.0) for k, v in pmc["kernels"].items()
                   if any(k.startswith(f) for f in fam[top["kernel"]]))
        if side > 0:
            top["traffic"] = side / pmc["sdf_algorithmic_bytes"] * top["alg_work_per_launch"]
            top["traffic_source"] = pmc["source"]
    if extra:
        top.update(extra)
    return top


def _style_seconds_with(pipe, inp, dev, **switches):
    """The stylisation stage once more, beside the clock, with other arithmetic switches of
    style/generators.py (EVAL_X3 / EVAL_DEFORM_X3)."""
    try:
        from drawingspinup_amd.style import generators
    except ImportError:
        return None
    if not hasattr(pipe, "stylize") or not hasattr(generators, "EVAL_DEFORM_X3"):
        return None
    _, frames_in, edges_in = inp
    saved = {k: getattr(generators, k) for k in switches}
    for k, v in switches.items():
        setattr(generators, k, v)
    try:
        pipe.stylize(frames_in[:2], edges_in[:2])             # warm-up (weight layouts, tap tables)
        _sync(dev); t = time.time()
        pipe.stylize(frames_in, edges_in)
        _sync(dev)
        return time.time() - t
    finally:
        for k, v in saved.items():
            setattr(generators, k, v)


def bench_drawing(args, ddist, rank, world, dev, timer, pipe=None):
    """`pipe`: a DrawingPipeline-like object (tests/test_dist_gloo.py drives this function over gloo
    with a stub); None builds the real one."""
    from drawingspinup_amd.drawing import (DrawingPipeline, synthetic_drawing, synthetic_edges,
                                           synthetic_frames)
    stub = pipe is not None
    stub_list = list(pipe) if isinstance(pipe, (list, tuple)) else None      # tests: one stub per drawing in flight
    if stub_list:
        pipe = stub_list[0]
    if pipe is None:
        pipe = DrawingPipeline(dev, seed=0, mv_steps=args.mv_steps, nsr_steps=args.nsr_steps,
                               n_frames=args.frames)
    bcast_bytes = sum(ddist.broadcast_module(m, 0) for m in pipe.shared_modules())
    # drawings in flight: one pipeline (own module instances: the per-module caches are not shared
    # between threads; same seed = same random weights) per concurrent drawing
    pipes = [pipe]
    if stub_list:
        for p2 in stub_list[1:]:
            for m in p2.shared_modules():
                ddist.broadcast_module(m, 0)
            pipes.append(p2)
    if not stub:
        import copy
        for _ in range(max(1, int(getattr(args, "inflight", 1))) - 1):
            try:
                # a device-side copy of the (already broadcast) first instance: no second random
                # initialisation of 910 M parameters on the host, no second broadcast
                p2 = copy.deepcopy(pipe)
            except Exception:                                          # noqa: BLE001
                p2 = DrawingPipeline(dev, seed=0, mv_steps=args.mv_steps, nsr_steps=args.nsr_steps,
                                     n_frames=args.frames)
                for m in p2.shared_modules():
                    ddist.broadcast_module(m, 0)
            p2.time_substages = True
            pipes.append(p2)
    stage_t = {"contour": 0.0, "mv": 0.0, "nsr": 0.0, "style": 0.0, "gather": 0.0}
    sub_t = {"nsr_matting": 0.0, "nsr_fit": 0.0, "nsr_export": 0.0, "nsr_post": 0.0}
    pipe.time_substages = True               # one extra synchronize between fit and export

    # inputs are generated before the clock starts and wait in HBM (the reference reads them
    # from disk; PNG decoding is listed as not timed)
    def make_inputs(seed):
        fr = synthetic_frames(seed, args.frames, device=dev)
        return (synthetic_drawing(seed, device=dev), fr, synthetic_edges(fr))
    n_in = len(pipes)
    inputs = {(False, w): make_inputs(1000 + rank * 100 + w) for w in range(args.warmup * n_in)}
    inputs.update({(True, s): make_inputs(rank * 100 + s) for s in range(args.steps * n_in)})
    _sync(dev)
    gathered = {}

    def one_drawing(s, timed):
        seed = (rank * 100 + s) if timed else (1000 + rank * 100 + s)
        drawing, frames_in, edges_in = inputs[(timed, s)]
        _sync(dev); t0 = time.time()
        cleaned = pipe.remove_contour(drawing)
        _sync(dev); t1 = time.time()
        normals, colors = pipe.multiview(cleaned, 123456 + seed)
        _sync(dev); t2 = time.time()
        system, inside = pipe.reconstruct(normals, colors, cleaned, 123456 + seed)
        _sync(dev); t3 = time.time()
        frames = pipe.stylize(frames_in, edges_in)
        _sync(dev); t4 = time.time()
        # the per-rank gather of OUTPUTS to rank 0 (north_star): the twelve predicted
        # views (12x3x256^2) and the stylised frames (n x 4 x 512^2 uint8), inside the clock.  The
        # mesh stays with its rank (the reference writes it to disk per uid).
        views = ddist.gather_tensor(torch.cat([normals, colors]).to(torch.float16))
        out_frames = ddist.gather_tensor(frames)
        _sync(dev); t5 = time.time()
        if timed:
            for k, v in zip(stage_t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                stage_t[k] += v
            for k in sub_t:
                sub_t[k] += pipe.substage_seconds.get(k, 0.0)
            if views is not None:
                gathered["views"], gathered["frames"] = views, out_frames
        return colors, inside.sum(), frames

    inflight = len(pipes)
    flight = None
    if torch.device(dev).type == "cuda":
        # side stream of the NSR step driver: high priority with drawings in flight (the optimisation keeps its precedence
        # over the other drawings' launches), normal with one drawing at a time (include/dsu_hip.h: streams of non-default
        # priority put every fourth and later drawing of a process into a 0.8 s slower mode)
        from drawingspinup_amd import _lib as dsu_lib_
        side_prio = getattr(args, "side_priority", None)
        side_prio = (2 if inflight == 1 else 1) if side_prio is None else int(side_prio)
        dsu_lib_.check(dsu_lib_.lib().dsu_set_nsr_side_stream_priority(side_prio), "dsu_set_nsr_side_stream_priority")
        dsu_lib_.check(dsu_lib_.lib().dsu_set_nsr_side_stream_pooling(int(getattr(args, "side_pool", 0) or 0)),
                       "dsu_set_nsr_side_stream_pooling")
    if inflight == 1:
        elapsed, last = _timed_loop(args, ddist, dev, timer, one_drawing)
    else:
        elapsed, flight = _inflight_loop(args, ddist, rank, dev, timer, pipes, inputs, stage_t, sub_t, gathered)
    if rank != 0:
        return None
    assert len(gathered["views"]) == world and len(gathered["frames"]) == world
    # seconds per DRAWING and stage (with drawings in flight: wall time of the stage on its stream,
    # beside the other drawings' kernels — the stages of one drawing then add up to its latency)
    per = {k: v / (args.steps * inflight) for k, v in stage_t.items()}
    per.update({k: v / (args.steps * inflight) for k, v in sub_t.items()})
    # The clock runs the reference's arithmetic per operator: stage 1's deform_conv2d layers in exact
    # f32 (torchvision im2col + f32 addmm, TF32 off by default), stage 2's nn.Conv2d layers on bf16 x 3
    # products (finer than the cuDNN TF32 the reference gets by default), IS-Net in exact f32 (f32
    # ONNX session).  Beside the clock, for comparison with earlier rounds' lines: the stage with
    # every layer on bf16 x 3, and with every layer in exact f32.
    per["style_all_bf16x3"] = _style_seconds_with(pipe, inputs[(True, args.steps - 1)], dev,
                                                  EVAL_X3=True, EVAL_DEFORM_X3=True)
    per["style_all_exact_f32"] = _style_seconds_with(pipe, inputs[(True, args.steps - 1)], dev,
                                                     EVAL_X3=False, EVAL_DEFORM_X3=False)
    stages = {
        "mv": {"bound": "mfma", "unit": "TFLOP/s", "peak": F16_MFMA_PEAK_TF,
               "achieved": 2.913 * args.mv_steps / max(per["mv"], 1e-9)},
        # achieved = ALGORITHMIC flops (0.84 TFLOP per frame for the two generators:
        # 0.298 stage 1 + 0.544 stage 2) / stage time, against the bf16 MFMA peak the contract names.
        # Stage 1 runs exact f32 products (f32 MFMA peak 157.3 TFLOP/s), stage 2 three bf16 MFMA
        # products per f32 product: the stage is a mix of the two pipes, so the fraction against
        # either single peak understates it; the per-pipe peaks are listed beside it.
        "style": {"bound": "mfma", "unit": "TFLOP/s", "peak": F16_MFMA_PEAK_TF,
                  "achieved": 0.84 * args.frames / max(per["style"], 1e-9),
                  "stage1_tflop_exact_f32": 0.298 * args.frames,
                  "stage2_tflop_bf16x3_issued": 3 * 0.544 * args.frames,
                  "f32_mfma_peak": F32_MFMA_PEAK_TF},
    }
    for st in stages.values():
        st["frac"] = st["achieved"] / st["peak"]
    # matrix-pipe busy share of the wave cycles and the parked / issue-stalled shares of the stage's
    # MFMA kernels, from the SQ pass (profiles/round{5,4}_pmc.json; one UNet forward / one frame)
    for name, pre in (("mv", ("conv_f16_kernel", "mv_attention_kernel")), ("style", ("conv_x3_kernel",))):
        for field in ("mfma_busy_of_wave_cycles", "parked_frac", "issue_stall_frac"):
            v = _sq_share(pre, field)
            if v is not None:
                stages[name][field] = v
    roof = _roofline(timer, {"stages": stages,
                             "traffic_note": "HBM-side bytes per launch of the dominant family "
                                             "(MLP part + scatter): (2 x FETCH_SIZE + "
                                             "WRITE_SIZE) of separate rocprofv3 --pmc passes "
                                             "(roofline.traffic_source, tools/pmc_round6.sh) relative to "
                                             "that workload's algorithmic bytes, applied to this run's "
                                             "mean algorithmic bytes per launch"})
    if flight and roof is not None:
        # kernel timing of ONE drawing alone beside the clock (the timed region's event pairs see the
        # kernels of the other drawings in flight inside their intervals)
        roof["in_flight"] = inflight
        roof["note_in_flight"] = ("avg_launch_ms / achieved / frac are HIP-event intervals inside the timed region, "
                                  "where %d drawings share the GPU: an interval contains the co-running kernels' "
                                  "share of the machine; roofline.alone = the same families with one drawing alone "
                                  "on the GPU (one more drawing after the clock, one workgroup per CU for the one-wave kernels, "
                                  "side stream at normal priority as for one drawing at a time)" % inflight)
        roof["alone"] = flight.get("alone")
        lead = next((r for r in (roof["alone"] or []) if r.get("kernel") == roof.get("kernel")), None)
        if lead:                                  # the leading family alone, next to the shared-machine figures
            roof["frac_alone"], roof["avg_launch_ms_alone"] = lead["frac"], lead["avg_launch_ms"]
    out = {
        "metric": "drawings/sec end-to-end (512x512, 6 views, 24 frames)",
        "value": world * inflight * args.steps / elapsed, "unit": "drawings/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (diffusion) / f16 table + f32 MLP (NSR) / f32 (stage-1 deformable convolutions, "
                 "IS-Net, contour) / f32 with bf16 x 3 products (stage-2 convolutions; the reference: "
                 "cuDNN TF32)",
        "data": "synthetic",
        "config": {"workload": ("%d drawings in flight per GPU (a step = %d drawings per GPU: one thread + one "
                                "stream + one pipeline instance each, every drawing the full chain below; "
                                "latency per drawing in config.latency_s), each drawing: " % (inflight, inflight)
                                if inflight > 1 else "one drawing per GPU: ") +
                               "contour removal (FFC-ResNet generator, masks, host "
                               "TELEA inpainting, 512^2) -> 6-view diffusion (%d DDIM steps, B=12) -> "
                               "IS-Net matting forward on the four 1024^2 side views (mv.py:113-150; random "
                               "weights: the filled-silhouette stand-in supplies the masks) -> NSR "
                               "recon (%d steps, 2x512^3 export: smoothing, marching cubes, quadric remeshing "
                               "to 50 000 faces, then save_mesh's Laplacian smoothing + colour "
                               "back-projection + shear: the reference YAML's export switches for a uid "
                               "outside the thinning list) -> %d-frame stage1+stage2 stylisation (stage 2 "
                               "on the edge-overlaid stage-1 output); NOT timed: Blender, PNG / OBJ file "
                               "I/O.  Image resampling between the stages (512 -> 256 RGBA bicubic, 256 -> 1024 "
                               "and 1024 -> 2048 LANCZOS) is Pillow's arithmetic restated on the device, "
                               "bit for bit (tests/test_mv_preprocess.py), on the 8-bit images the "
                               "reference's PNG hand-offs hold" % (args.mv_steps, args.nsr_steps, args.frames),
                   "drawings_per_step": world * inflight,
                   "parallelism": f"replica-per-drawing x{world}" + (f", {inflight} in flight per GPU" if inflight > 1 else ""),
                   "latency_s": flight.get("latency") if flight else None,
                   "inflight_schedule": flight.get("schedule") if flight else None,
                   # arithmetic per stage next to the reference's own (file:line in DESIGN.md 4)
                   "stage_dtype": {
                       "contour": "f32 (exact f32 MFMA; reference: cuDNN / cuFFT f32)",
                       "mv": "f16 (reference: weight_dtype = torch.float16, mv.py:15)",
                       "matting": "f32 (exact f32 MFMA; reference: f32 ONNX session, mv.py:17-18)",
                       "nsr": "f16 table + f32 MLP (reference: tcnn f16 grid, f32 VanillaMLP)",
                       "style_stage1": "f32 (exact f32 MFMA; reference: torchvision deform_conv2d "
                                       "= im2col + f32 addmm, TF32 off by default)",
                       "style_stage2": "f32 activations, bf16 x 3 products, f32 accumulation (2^-15 "
                                       "per product; reference: cuDNN with TF32 allowed by default, 2^-11)"},
                   "stage_seconds_rank0": per, "weights_broadcast_bytes": bcast_bytes,
                   "gathered_bytes_per_step": sum(t.numel() * t.element_size() for k in gathered
                                                  for t in gathered[k])},
        "roofline": roof,
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.nsr_steps, args.frames, args.mv_steps)
    return out


def bench_unet(args, ddist, rank, world, dev, timer):
    """BASELINE configs[1]: 'single drawing, mvdiffusion 6-view 256x256 UNet forward, random weights':
    UNetMV2DConditionModel.forward (unet_mv2d_condition.py:760-1054) on (12, 8, 32, 32) latents —
    noise + image latents of 6 views x 2 domains — with the CLIP embedding and the camera / domain
    class labels; every rank runs its own replica."""
    from drawingspinup_amd.mv.unet import UNetMV2DConditionModel
    torch.manual_seed(0)
    unet = UNetMV2DConditionModel().half().to(dev).eval()
    ddist.broadcast_module(unet, 0)
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.randn(12, 8, 32, 32, generator=g).half().to(dev)
    ctx = torch.randn(12, 1, 768, generator=g).half().to(dev)
    cl = torch.randn(12, 10, generator=g).half().to(dev)
    ts = torch.tensor([500], device=dev)

    @torch.no_grad()
    def step(s, timed):
        return unet(x, ts, ctx, cl)

    elapsed, out = _timed_loop(args, ddist, dev, timer, step)
    if rank != 0:
        return None
    assert out.shape == (12, 4, 32, 32) and bool(torch.isfinite(out).all())
    fps = world * args.steps / elapsed
    tflop = 2.913                                  # algorithmic flops of one forward
    return {"metric": "multi-view UNet forwards/sec (6 views x 2 domains, 256x256, B=12)",
            "value": fps, "unit": "forwards/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "one UNetMV2DConditionModel forward per step on (12,8,32,32) f16 "
                                   "latents, one CLIP token, 10 class-label values; random-init weights "
                                   "of the shipped architecture (320/640/1280/1280, 2 layers per block)",
                       "parallelism": f"replica x{world}"},
            "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": F16_MFMA_PEAK_TF,
                         "achieved": tflop * fps / world, "frac": tflop * fps / world / F16_MFMA_PEAK_TF,
                         "traffic": None, "kernels": timer.summary()[:3]}}


def bench_nsr50k(args, ddist, rank, world, dev, timer):
    """BASELINE configs[2]: '128^3 hash-grid 50k rays/iter' = 128^3 occupancy grid + 50 000 rays per
    optimisation iteration on the synthetic sphere; every rank runs its own replica."""
    from drawingspinup_amd.nsr.model import Cfg, DEFAULT_MODEL_CONFIG
    from drawingspinup_amd.nsr.system import OrthoData, OrthoNeuSSystem
    R = 50000
    cfg = Cfg(DEFAULT_MODEL_CONFIG)
    cfg["train_num_rays"], cfg["max_train_num_rays"], cfg["dynamic_ray_sampling"] = R, R, False
    ds = OrthoData.synthetic_sphere(1024, device=dev)
    sysm = OrthoNeuSSystem(model_config=cfg, device=dev, seed=rank)
    sysm.dataset = ds
    for _ in range(300):                      # occupancy grid warm-up on a small ray count
        sysm.train_num_rays = 2048
        sysm.training_step()
    samples = [0]

    def step(s, timed):
        sysm.train_num_rays = R
        r = sysm.training_step()
        if timed:
            samples[0] += int(r["n_samples"])
        return r

    elapsed, _ = _timed_loop(args, ddist, dev, timer, step)
    if rank != 0:
        return None
    n = samples[0] / max(args.steps, 1)
    return {"metric": "NSR optimisation iterations/sec (128^3 occupancy grid, 50k rays/iter)",
            "value": world * args.steps / elapsed, "unit": "iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 table + f32 MLP", "data": "synthetic",
            "config": {"workload": "NSR micro-benchmark: 50 000 rays per iteration marched at "
                                   "3.383e-3 through a 128^3 occupancy grid of the synthetic sphere, "
                                   "full optimisation step (7-evaluation SDF forward/backward, "
                                   "texture MLP, compositing, losses, AdamW)",
                       "rays_per_s": world * R * args.steps / elapsed,
                       "samples_per_iteration": n, "samples_per_s": world * n * args.steps / elapsed,
                       "point_evaluations_per_s": world * 7 * (n + 4096) * args.steps / elapsed,
                       "parallelism": f"replica x{world}"},
            "roofline": _roofline(timer)}


def bench_frames(args, ddist, rank, world, dev, timer, pipe=None, size=512):
    """BASELINE configs[3]: 24 frames 512x512 through stage 1 + stage 2, frame f on rank f mod N,
    stage 2 on the rank of its stage-1 frame (no exchange), outputs gathered to rank 0.
    `pipe` / `size`: stub pipeline and frame size of the gloo test."""
    from drawingspinup_amd.drawing import DrawingPipeline, synthetic_edges, synthetic_frames
    if pipe is None:
        pipe = DrawingPipeline(dev, seed=0, n_frames=args.frames, with_mv=False, with_contour=False)
    for m in (pipe.gen1, pipe.gen2):
        ddist.broadcast_module(m, 0)
    frames = synthetic_frames(0, args.frames, size, device=dev)
    edges = synthetic_edges(frames)
    mine = ddist.shard(list(range(args.frames)), rank, world)
    per_rank = (args.frames + world - 1) // world

    def step(s, timed):
        out = torch.zeros(per_rank, 4, frames.shape[2], frames.shape[3], dtype=torch.uint8, device=dev)
        if len(mine):
            out[:len(mine)] = pipe.stylize(frames[mine], edges[mine])
        return ddist.gather_tensor(out)                 # equal shapes: padded to ceil(frames / N)

    elapsed, parts = _timed_loop(args, ddist, dev, timer, step)
    if rank != 0:
        return None
    # rank 0 puts the shards back in frame order (frame f = row f // N of rank f % N's part)
    ordered = torch.stack([parts[f % world][f // world] for f in range(args.frames)])
    fps = args.frames * args.steps / elapsed
    return {"metric": "stylised frames/sec (512x512, stage 1 + stage 2)", "value": fps,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 activations, bf16 x 3 products with f32 accumulation", "data": "synthetic",
            "config": {"workload": "%d frames 512x512 per step through GeneratorJ_RIC + GeneratorJ, "
                                   "frames sharded round-robin over the ranks, outputs gathered to "
                                   "rank 0" % args.frames,
                       "frames_per_rank": len(mine), "parallelism": f"frame-shard x{world}",
                       "gathered_frames_checksum": int(ordered.to(torch.int64).sum())},
            # algorithmic flops (0.84 TFLOP per frame) / time; the bf16 x 3 kernels
            # issue three MFMA products per f32 product, reported beside it
            "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": F16_MFMA_PEAK_TF,
                         "achieved": 0.84 * fps, "frac": 0.84 * fps / F16_MFMA_PEAK_TF,
                         "issued_mfma_tflops": 3 * 0.84 * fps, "f32_mfma_peak": F32_MFMA_PEAK_TF,
                         "traffic": None, "kernels": timer.summary()[:3]}}


def _spawn_worker(local_rank, args, port):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank),
                       "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    run(args)


def main():
    args = parse()
    launched = int(os.environ.get("WORLD_SIZE", "1")) > 1 or "RANK" in os.environ
    if args.gpus > 1 and not launched:
        # `python bench.py --gpus N`: spawn the N ranks here (one process per GPU)
        import torch.multiprocessing as mp
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_spawn_worker, args=(args, port), nprocs=args.gpus, join=True)
        return
    run(args)


if __name__ == "__main__":
    main()
