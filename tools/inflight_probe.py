"""Drawings in flight per GPU: K Python threads, each with its own DrawingPipeline (own module
instances: the per-module caches are not shared) and its own stream, each running whole drawings
back to back, the second started half a drawing later (stage-skewed).  Reports drawings/s and the
latency of a drawing for K = 1 and K = 2 on the same box, and checks that a drawing's outputs do not
depend on what runs beside it.
    python tools/inflight_probe.py [drawings_per_thread] [K ...]"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd.drawing import DrawingPipeline, synthetic_drawing, synthetic_edges, synthetic_frames

dev = torch.device("cuda:0")
per_thread = int(sys.argv[1]) if len(sys.argv) > 1 else 3
Ks = [int(a) for a in sys.argv[2:]] or [1, 2]
nsr_steps = int(os.environ.get("NSR_STEPS", "3000"))
mv_steps = int(os.environ.get("MV_STEPS", "75"))
pipes = [DrawingPipeline(dev, seed=0, mv_steps=mv_steps, nsr_steps=nsr_steps, n_frames=24) for _ in range(max(Ks))]


def one(pipe, seed, stream):
    with torch.cuda.stream(stream):
        drawing = synthetic_drawing(seed, device=dev)
        fr = synthetic_frames(seed, 24, device=dev)
        ed = synthetic_edges(fr)
        t0 = time.time()
        cleaned = pipe.remove_contour(drawing)
        normals, colors = pipe.multiview(cleaned, 123456 + seed)
        system, inside = pipe.reconstruct(normals, colors, cleaned, 123456 + seed)
        frames = pipe.stylize(fr, ed)
        sig = (float(colors.double().sum()), int(inside.sum()), int(frames.long().sum()),
               int(pipe.last_mesh_post["faces"].shape[0]) if getattr(pipe, "last_mesh_post", None) else -1)
        stream.synchronize()
        return time.time() - t0, sig


# warm-up: every pipeline once, alone
for p in pipes:
    one(p, 999, torch.cuda.Stream(dev))
torch.cuda.synchronize()
ref = {}
for K in Ks:
    lat, sigs, lock = [], {}, threading.Lock()

    def worker(k):
        s = torch.cuda.Stream(dev)
        torch.cuda.set_device(dev)
        if k:
            time.sleep(2.2 * k / K)                    # stage skew
        for j in range(per_thread):
            seed = 10 * k + j
            dt, sig = one(pipes[k], seed, s)
            with lock:
                lat.append(dt); sigs[seed] = sig
    torch.cuda.synchronize(); t0 = time.time()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); el = time.time() - t0
    skew = 2.2 * (K - 1) / K
    print(f"K={K}: {K * per_thread} drawings in {el:.2f} s ({el - skew:.2f} s without the start skew) -> "
          f"{K * per_thread / el:.4f} drawings/s ({K * per_thread / (el - skew):.4f} steady-ish); latency mean "
          f"{sum(lat) / len(lat):.2f} s max {max(lat):.2f} s", flush=True)
    for seed, sig in sorted(sigs.items()):
        if seed in ref and ref[seed] != sig:
            print(f"   seed {seed}: outputs differ from the run alone: {ref[seed]} vs {sig}")
        ref.setdefault(seed, sig)
# the same seeds once more alone (K = 1 order), to compare the K = 2 signatures against
if max(Ks) > 1:
    for seed in sorted(ref):
        _, sig = one(pipes[0], seed, torch.cuda.Stream(dev))
        print(f"   seed {seed}: alone {sig} {'==' if sig == ref[seed] else '!='} concurrent {ref[seed]}")
