"""One process per GPU; drawings / frames shard embarrassingly.

The only collectives on the path: an RCCL broadcast of the shared read-only weights once
(root 0) and a gather of the small per-rank outputs.  There is no per-step collective: every
drawing owns its diffusion sample, its NSR optimisation and its per-character generators.
"""
import os

import torch
import torch.distributed as dist


_FORCED = False      # a process group of ONE rank whose collectives really run (see init)


def _ranks_on_this_node(world):
    """LOCAL_WORLD_SIZE (torchrun sets it); else `world` (single node assumed)."""
    try:
        n = int(os.environ.get("LOCAL_WORLD_SIZE", "0"))
    except ValueError:
        n = 0
    return n if 0 < n <= world else world


def _active():
    return dist.is_initialized() and (dist.get_world_size() > 1 or _FORCED)


def init(backend=None, force=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local).
    force (default: DSU_DIST_FORCE=1 in the environment): create the process group even for a
    world of one rank and run the collectives of this module on it — one GPU is enough to see RCCL
    initialise and the broadcast / gather / all-reduce return (profiles/round5_rccl_world1.txt)."""
    global _FORCED
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force is None:
        force = os.environ.get("DSU_DIST_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        # the ranks of one node share its host cores: the host steps of a drawing (TELEA tail, the
        # serial finish of the decimation, sparse solves, torch's intra-op pools) get an equal
        # share instead of N pools of `cores` threads each
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, cores // _ranks_on_this_node(world)))
    # also when the caller (or a torchrun wrapper, or an earlier init()) created the group: a forced
    # one-rank run must exercise the collectives, not skip them silently
    _FORCED = bool(force) and world == 1 and dist.is_initialized()
    if _FORCED and rank == 0:
        print("[dsu dist] forced one-rank process group: collectives run (DSU_DIST_FORCE)",
              flush=True)
    return rank, world, local


def shard(items, rank, world):
    """Static round-robin partition: item i -> rank i % world (uids / frames)."""
    return [it for i, it in enumerate(items) if i % world == rank]


@torch.no_grad()
def broadcast_module(module, src=0, bucket_bytes=256 << 20):
    """Broadcast every parameter and buffer of `module` from rank `src`, coalesced into large
    flat buckets (xGMI is point-to-point: few big messages, not one per tensor)."""
    if not _active():
        return 0
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.numel() > 0]
    total = 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in by_dtype.items():
        bucket, size = [], 0
        for t in ts + [None]:
            if t is not None and (size + t.numel() * t.element_size() <= bucket_bytes or not bucket):
                bucket.append(t)
                size += t.numel() * t.element_size()
                continue
            flat = torch.cat([b.detach().reshape(-1) for b in bucket])
            dist.broadcast(flat, src)
            off = 0
            for b in bucket:
                b.detach().copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            total += size
            bucket, size = ([t], t.numel() * t.element_size()) if t is not None else ([], 0)
    return total


def gather_tensor(t, dst=0):
    """Gather equally-shaped per-rank tensors on `dst` (list on dst, None elsewhere): a gather,
    not an all-gather — only rank `dst` receives the (small) outputs."""
    if not _active():
        return [t]
    world = dist.get_world_size()
    out = [torch.empty_like(t) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(t.contiguous(), out, dst)
    return out


def barrier():
    if _active():
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if _active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
