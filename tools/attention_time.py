"""us per call of the multi-view attention kernel at the UNet's shapes (B = 12, 6 views x 2 domains:
each query batch attends to the 6 views of its domain).  usage: attention_time.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drawingspinup_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator().manual_seed(0)
seg = torch.tensor([[6 * (b // 6) + v for v in range(6)] for b in range(12)], dtype=torch.int32, device=dev)
out = []
for tokens, width in ((1024, 320), (256, 640), (64, 1280), (16, 1280)):
    q = torch.randn(12, tokens, width, generator=g).half().to(dev)
    k = torch.randn(12, tokens, width, generator=g).half().to(dev)
    vt = torch.randn(12, width, tokens, generator=g).half().to(dev)
    for _ in range(5):
        ops.mv_attention(q, k, vt, seg, 8, tokens)
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        ops.mv_attention(q, k, vt, seg, 8, tokens)
    e.record()
    torch.cuda.synchronize()
    out.append("%dx%d(d=%d) %.1f us" % (tokens, width, width // 8, s.elapsed_time(e) / reps * 1e3))
print(os.path.basename(os.environ.get("DSU_HIP_LIB", "default")), " | ".join(out))
