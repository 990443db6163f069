import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd import ops
dev = torch.device("cuda:0")
cfg = ops.HashGridConfig()
g = torch.Generator().manual_seed(0)
tab = ((torch.rand(cfg.n_entries, 2, generator=g) * 2 - 1) * 0.1).half().to(dev)
mlp = [(torch.randn(64, 23, generator=g) * 0.3).to(dev), (torch.randn(64, generator=g) * 0.05).to(dev),
       (torch.randn(13, 64, generator=g) * 0.2).to(dev), (torch.randn(13, generator=g) * 0.1).to(dev)]
saved = torch.load("/tmp/fwd_inputs.pt")
def timeit(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / reps * 1e3, 1)
out = {}
for step, d in sorted(saved.items()):
    act = d["active"]
    for n in (131072, 196608, 229376, 245760, 258048, 262144, 266240, 278528, 294912):
        if n > d["pts"].shape[0]:
            pts = torch.cat([d["pts"], d["pts"][: n - d["pts"].shape[0]]])
        else:
            pts = d["pts"][:n].contiguous()
        fn = lambda: ops.sdf_fd_fwd(cfg, d["tab"], d["mlp"], pts, d["radius"], d["eps"], act, True, True, False, enc_cache=True)
        out[f"act{act}_n{n}"] = timeit(fn)
print(json.dumps(out))
