"""CPU, world_size 2 over gloo: the N>1 path of bench.py — weight broadcast from rank 0,
round-robin sharding, gather of the per-rank outputs, max-over-ranks timing."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from drawingspinup_amd import dist as ddist
    r, w, _ = ddist.init(backend="gloo")
    torch.manual_seed(100 + rank)                     # ranks start with DIFFERENT weights
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    net[1].running_mean.add_(rank + 1.0)
    nbytes = ddist.broadcast_module(net, 0, bucket_bytes=64)      # tiny buckets: several messages
    flat = torch.cat([t.detach().reshape(-1).float() for t in list(net.parameters()) + list(net.buffers())])
    mine = ddist.shard(list(range(10)), r, w)
    out = ddist.gather_tensor(torch.tensor([float(sum(mine))]), 0)
    tmax = ddist.max_over_ranks(1.0 + rank, "cpu")
    ddist.barrier()
    q.put((rank, flat.sum().item(), nbytes, mine, None if out is None else [float(o) for o in out], tmax))
    torch.distributed.destroy_process_group()


def test_broadcast_shard_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, n0, m0, g0, t0), (r1, s1, n1, m1, g1, t1) = res
    assert s0 == s1                                   # rank 1 now holds rank 0's weights + buffers
    assert n0 == n1 and n0 > 0
    assert m0 == [0, 2, 4, 6, 8] and m1 == [1, 3, 5, 7, 9]
    assert g0 == [20.0, 25.0] and g1 is None          # gathered on rank 0 only
    assert t0 == 2.0 and t1 == 2.0                    # max over ranks


# ------------------------------------------------------------------------------------------------
# bench.py's N > 1 paths themselves (the timed loop, the frame shard / pad / gather / re-order of
# `--config frames`, the output gather of `--config drawing`) over gloo, with a stub in place of
# the GPU pipeline: the 8-GPU runs are the driver's, this is what can be executed here.
# ------------------------------------------------------------------------------------------------
class _StubTimer:
    enabled = False

    def summary(self):
        return []


class _StubPipe:
    """DrawingPipeline's interface on CPU tensors; every output is a fixed function of its input
    and of the (broadcast) weights, so rank 0 can check what it gathered."""

    def __init__(self, rank):
        torch.manual_seed(500 + rank)                                 # different until broadcast
        self.gen1 = torch.nn.Conv2d(6, 3, 1)
        self.gen2 = torch.nn.Conv2d(6, 3, 1)
        self.time_substages, self.substage_seconds = False, {}

    def shared_modules(self):
        return [self.gen1, self.gen2]

    def remove_contour(self, d):
        return d

    def multiview(self, drawing, seed):
        base = torch.nn.functional.interpolate(drawing[None, :3], size=(16, 16))[0]
        n = torch.stack([base * (0.1 * (v + 1)) for v in range(6)])
        return n, 1.0 - n

    def reconstruct(self, normals, colors, drawing, seed):
        self.substage_seconds = {"nsr_fit": 0.0, "nsr_export": 0.0, "nsr_post": 0.0}
        return None, (colors > 0.5)

    @torch.no_grad()
    def stylize(self, frames, edges=None):
        s1 = torch.tanh(self.gen1(frames))
        q = ((s1.clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)
        if edges is not None:
            q = torch.where((edges < 255)[:, None], torch.zeros_like(q), q)
        s2 = torch.tanh(self.gen2(torch.cat([q.float() / 255 * 2 - 1, frames[:, 3:]], 1)))
        rgb = ((s2.clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)
        return torch.cat([rgb, (frames[:, 3:4] * 255).to(torch.uint8)], 1)


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from drawingspinup_amd import dist as ddist
    from drawingspinup_amd.drawing import synthetic_edges, synthetic_frames
    r, w, _ = ddist.init(backend="gloo")
    threads = torch.get_num_threads()
    dev = torch.device("cpu")
    pipe = _StubPipe(rank)
    args = bench.parse(["--gpus", str(world), "--config", "frames", "--frames", "5", "--steps", "2",
                        "--warmup", "1"])
    out_f = bench.bench_frames(args, ddist, r, w, dev, _StubTimer(), pipe=pipe, size=32)
    want = None
    if r == 0:                                # what one process computes for all five frames
        fr = synthetic_frames(0, 5, 32, device=dev)
        want = int(pipe.stylize(fr, synthetic_edges(fr)).to(torch.int64).sum())
    args = bench.parse(["--gpus", str(world), "--config", "drawing", "--frames", "3", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"])
    out_d = bench.bench_drawing(args, ddist, r, w, dev, _StubTimer(), pipe=pipe)
    # the same with two drawings in flight per rank (bench._inflight_loop: worker threads, the
    # collectives kept in one order by the main thread)
    args = bench.parse(["--gpus", str(world), "--config", "drawing", "--frames", "3", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--inflight", "2"])
    out_i = bench.bench_drawing(args, ddist, r, w, dev, _StubTimer(), pipe=[pipe, _StubPipe(rank + 7)])
    ddist.barrier()
    q.put((rank, threads, out_f, want, out_d, out_i))
    torch.distributed.destroy_process_group()


def test_bench_frames_and_drawing_paths_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, th0, f0, want, d0, i0), (_, th1, f1, _, d1, i1) = res
    # two in flight: a step is 2 drawings per rank, the value counts all of them, every drawing's
    # outputs were gathered (the last gather holds both ranks' tensors), latency is reported
    assert i1 is None and i0["config"]["drawings_per_step"] == 4 and i0["steps"] == 2
    assert abs(i0["value"] - 2 * 2 * 2 / (i0["ms_per_step"] * 2e-3)) < 1e-6 * i0["value"]
    assert i0["config"]["latency_s"]["drawings"] == 4 and "2 drawings in flight" in i0["config"]["workload"]
    assert i0["config"]["gathered_bytes_per_step"] == 2 * (12 * 3 * 16 * 16 * 2 + 3 * 4 * 512 * 512)
    cores = len(os.sched_getaffinity(0))
    assert th0 == th1 == max(1, cores // 2)                # the ranks split the host cores
    assert f1 is None and d1 is None                       # only rank 0 reports
    # frames: 5 frames over 2 ranks = 3 + 2 (padded to 3), gathered and put back in frame order
    assert f0["n_gpus"] == 2 and f0["scaling"] == "strong" and f0["steps"] == 2
    assert f0["config"]["frames_per_rank"] == 3
    assert f0["config"]["gathered_frames_checksum"] == want
    assert abs(f0["value"] - 5 * 2 / (f0["ms_per_step"] * 2e-3)) < 1e-6 * f0["value"]
    # drawing: one drawing per rank per step, whole-job value, the views and frames of BOTH ranks
    assert d0["n_gpus"] == 2 and d0["scaling"] == "weak" and d0["config"]["drawings_per_step"] == 2
    assert abs(d0["value"] - 2 * 2 / (d0["ms_per_step"] * 2e-3)) < 1e-6 * d0["value"]
    per_rank = 12 * 3 * 16 * 16 * 2 + 3 * 4 * 512 * 512
    assert d0["config"]["gathered_bytes_per_step"] == 2 * per_rank
    assert d0["config"]["weights_broadcast_bytes"] > 0 and "gather" in d0["config"]["stage_seconds_rank0"]


def _forced_world1(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), DSU_DIST_FORCE="1", LOCAL_WORLD_SIZE="1")
    from drawingspinup_amd import dist as ddist
    r, w, _ = ddist.init(backend="gloo")
    net = torch.nn.Linear(4, 3)
    nbytes = ddist.broadcast_module(net, 0)
    out = ddist.gather_tensor(torch.arange(3.0), 0)
    ddist.barrier()
    q.put((r, w, torch.distributed.is_initialized(), nbytes, [float(v) for v in out[0]],
           ddist.max_over_ranks(2.5, "cpu")))
    torch.distributed.destroy_process_group()


def test_forced_single_rank_group_runs_the_collectives():
    """DSU_DIST_FORCE=1: a process group of one rank is created and broadcast / gather /
    all-reduce go through the backend instead of being short-circuited (the form that shows RCCL
    alive on a 1-GPU box)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_world1, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=120)
    p.join(60)
    assert got[:3] == (0, 1, True)
    assert got[3] == (4 * 3 + 3) * 4            # the bytes really broadcast
    assert got[4] == [0.0, 1.0, 2.0] and got[5] == 2.5


def test_threads_are_shared_by_the_ranks_of_one_node(monkeypatch):
    from drawingspinup_amd import dist as ddist
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert ddist._ranks_on_this_node(16) == 8        # 2 nodes x 8 ranks: cores / 8, not / 16
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert ddist._ranks_on_this_node(4) == 4
