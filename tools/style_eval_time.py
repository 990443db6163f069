"""Time the two shipped generators in eval mode at the bench's shape (4 frames of 512x512):
the older exact-f32 kernels (style_conv.hip, im2col through LDS), the packed-weight exact-f32
kernels (style_conv_x3.hip, F32) and the bf16 x 3 kernels, and report the largest output difference
to the older exact-f32 result.

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
        packed_ok = G._packed_ok
        for mode in ("f32_lds", "f32_packed", "x3"):
            G.EVAL_X3 = G.EVAL_DEFORM_X3 = mode == "x3"
            # f32_lds: route every layer to style_conv.hip (the kernels before round 6)
            G._packed_ok = (lambda conv, exact: False) if mode == "f32_lds" else packed_ok
            if mode == "f32_lds":
                dc = G.ops.deform_conv3x3_x3
                G.ops.deform_conv3x3_x3 = lambda x, off, pk, *a, _w={}: G.ops.deform_conv3x3(
                    x[:, :pk.C].contiguous(), off, pk._src, *a)
                PW = G.ops.PackedConvWeight

                class _Raw:                      # carries the OIHW weight instead of a packed one
                    def __init__(self, w, exact=False):
                        self._src, self.C = w, w.shape[1]
                G.ops.PackedConvWeight = _Raw
            for m in net.modules():
                m.__dict__.pop("_dsu_pack", None)
            with torch.no_grad():
                y = net(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    y = net(x)
                torch.cuda.synchronize()
            out[mode] = (y, (time.perf_counter() - t0) / reps)
            if mode == "f32_lds":
                G.ops.deform_conv3x3_x3, G.ops.PackedConvWeight = dc, PW
        G._packed_ok = packed_ok
        ref = out["f32_lds"][0]
        print("%-15s f32 (LDS im2col) %.2f ms   f32 (packed) %.2f ms   bf16x3 %.2f ms   "
              "max|dy| packed %.2e  x3 %.2e" % (
                  name, out["f32_lds"][1] * 1e3, out["f32_packed"][1] * 1e3, out["x3"][1] * 1e3,
                  float((out["f32_packed"][0] - ref).abs().max()),
                  float((out["x3"][0] - ref).abs().max())))


if __name__ == "__main__":
    main()
