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
