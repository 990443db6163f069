"""Are some device allocations slower than others?  (NSR drawings run in a fast or a slow mode depending on
where their state was allocated.)  Fresh caching-allocator segments of several sizes, each timed with a streaming update
(read + write) and a random 4-byte gather; printed with the pointer's offset inside its 2 MiB page.
usage: mem_speed_probe.py [rounds]"""
import sys, torch
dev = "cuda"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps


def probe(t):
    n = t.numel()
    idx = torch.randint(0, n, (1 << 22,), device=dev, generator=g)
    ms_stream = timeit(lambda: t.mul_(1.0))
    ms_gather = timeit(lambda: torch.index_select(t, 0, idx))
    return 2 * n * 4 / ms_stream / 1e6, (1 << 22) * 4 / ms_gather / 1e6


sizes_mb = [15.2, 30.4, 64, 200.5, 512]
for r in range(rounds):
    keep = []
    for mb in sizes_mb:
        for k in range(4):
            t = torch.empty(int(mb * (1 << 20)) // 4, dtype=torch.float32, device=dev).fill_(1.0)
            keep.append(t)
            s, gth = probe(t)
            print("round %d  %7.1f MiB  ptr %#x  off2M %7d  stream %7.1f GB/s  gather %6.1f GB/s" %
                  (r, mb, t.data_ptr(), t.data_ptr() % (2 << 20), s, gth), flush=True)
    # free every other tensor, drop the cache: the next round's segments come from a fragmented address space
    keep = keep[::2]
    torch.cuda.empty_cache()
