"""Where nsr/mesh.py's remesh() (fine mesh of the 512^3 export -> 50 000 faces) spends its time: the
parts of the Python wrapper with a synchronise + wall clock around each, on a marching-cubes mesh of
an analytic field of the export's size (~3 M faces).  Under rocprofv3 --kernel-trace --stats the
kernel table of the same run says what the device call itself consists of.
    python tools/remesh_profile.py [res] [reps]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from drawingspinup_amd import _lib, ops  # noqa: E402
from drawingspinup_amd.nsr import mesh as M  # noqa: E402

dev = torch.device("cuda:0")
res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = torch.linspace(-1, 1, res, dtype=torch.float64, device=dev)
x, y, z = torch.meshgrid(c, c, c, indexing="ij")
vol = 0.62 - torch.sqrt((x / 0.8) ** 2 + (y / 0.6) ** 2 + (z / 0.7) ** 2) \
    + 0.05 * torch.sin(9 * x + 0.3) * torch.sin(7 * y + 1.1) * torch.sin(8 * z + 2.0)
del x, y, z
verts, faces = M.marching_cubes(vol, 0.0)
del vol
verts = verts / (res - 1.0)
print("mesh", tuple(verts.shape), tuple(faces.shape))


def sync():
    torch.cuda.synchronize()
    return time.time()


for rep in range(reps):
    t0 = sync()
    v, f = M.remesh(verts, faces, 50000)
    t1 = sync()
    print(f"rep {rep}: remesh {1e3 * (t1 - t0):.1f} ms -> {f.shape[0]} faces; stats {M.last_remesh_stats}")
    # the same call part by part
    T = {}
    t = sync()
    vv = verts.detach().to(torch.float64).contiguous().clone()
    ff = faces.detach().to(dev, torch.int32).contiguous().clone()
    nv, nf = vv.shape[0], ff.shape[0]
    lib = _lib.lib()
    ws_bytes = int(lib.dsu_mesh_decimate_parallel_workspace_bytes(nv, nf))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    quad = torch.empty(nv, 10, dtype=torch.float64, device=dev)
    T["copies + workspace"] = sync() - t
    t = sync()
    out_nf = C.c_int64(0)
    stats = (C.c_int32 * 3)()
    ops.check(lib.dsu_mesh_decimate_parallel(
        vv.data_ptr(), nv, ff.data_ptr(), nf, int(M.PARALLEL_STOP * 50000), int(M.PARALLEL_FLOOR * 50000), 1.0, 0,
        200, quad.data_ptr(), C.byref(out_nf), stats, ws.data_ptr(), ws_bytes,
        torch.cuda.current_stream(dev).cuda_stream), "dsu_mesh_decimate_parallel")
    T["dsu_mesh_decimate_parallel"] = sync() - t
    t = sync()
    ff = ff[:out_nf.value]
    used, inv = torch.unique(ff.reshape(-1).long(), return_inverse=True)
    T["unique"] = sync() - t
    t = sync()
    hv = np.ascontiguousarray(vv[used].cpu().numpy())
    hq = np.ascontiguousarray(quad[used].cpu().numpy())
    hf = np.ascontiguousarray(inv.reshape(-1, 3).to(torch.int32).cpu().numpy())
    T["gather + copies to the host"] = sync() - t
    t = sync()
    M._remesh_host(hv, hf, 50000, 1.0, True, quadrics=hq)
    T["serial queue on the host"] = sync() - t
    for k, val in T.items():
        print(f"    {k:32s} {1e3 * val:8.1f} ms")
    print(f"    workspace {ws_bytes / 1e6:.0f} MB, rounds {stats[0]}")
