"""Time the two shipped generators in eval mode at the bench's shape (4 frames of 512x512),
exact-f32 kernels vs the bf16 x 3 kernels, and report the largest output difference.

    python tools/style_eval_time.py [batch] [reps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from drawingspinup_amd.style import generators as G
from oracle import style_ref


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    for name in ("GeneratorJ_RIC", "GeneratorJ"):
        net = G.build_model(name, style_ref.FULLSIZE_ARGS)
        net.load_state_dict(style_ref.seeded_state_dict(net.state_dict(), 7))
        net = net.to(dev).eval()
        x = torch.cat([style_ref.fullsize_frame(11 + i) for i in range(batch)]).to(dev)
        out = {}
        for x3 in (False, True):
            G.EVAL_X3 = x3
            with torch.no_grad():
                y = net(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    y = net(x)
                torch.cuda.synchronize()
            out[x3] = (y, (time.perf_counter() - t0) / reps)
        d = (out[True][0] - out[False][0]).abs()
        print("%-15s f32 %.2f ms   bf16x3 %.2f ms   (x%.2f)   max|dy| %.2e  mean %.2e" % (
            name, out[False][1] * 1e3, out[True][1] * 1e3, out[False][1] / out[True][1],
            float(d.max()), float(d.mean())))


if __name__ == "__main__":
    main()
