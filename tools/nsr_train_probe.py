import sys, time, torch
sys.path.insert(0,'.')
from drawingspinup_amd.nsr.system import OrthoNeuSSystem, OrthoData
dev='cuda'
ds = OrthoData.synthetic_sphere(1024, device=dev)
sysm = OrthoNeuSSystem(device=dev)
sysm.dataset = ds
torch.cuda.synchronize(); t=time.time()
for s in range(int(sys.argv[1]) if len(sys.argv)>1 else 300):
    r = sysm.training_step()
    if (s+1) % 50 == 0:
        torch.cuda.synchronize()
        print(s+1, f"{(time.time()-t)/50*1000:.2f} ms/step", "loss %.4f"%float(r['loss']), "rays", r['n_rays'], "samples", r['n_samples'], {k: round(float(v),4) for k,v in r.items() if k not in ('loss','n_rays','n_samples')}, flush=True)
        t=time.time()
torch.cuda.synchronize(); t=time.time()
m = sysm.export_mesh()
torch.cuda.synchronize(); print("export (2x512^3 SDF, smoothing, marching cubes, colours): %.3f s"%(time.time()-t), "verts", tuple(m["verts"].shape), "faces", tuple(m["faces"].shape), m.get("vmin"), m.get("vmax"))
