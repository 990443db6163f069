import sys, time, torch
sys.path.insert(0,'.')
from drawingspinup_amd import ops
dev='cuda'
cfg=ops.HashGridConfig()
g=torch.Generator().manual_seed(0)
tab=((torch.rand(cfg.n_entries,2,generator=g)*2-1)*0.1).half().to(dev)
mlp=[(torch.randn(64,23,generator=g)*0.3).to(dev),(torch.randn(64,generator=g)*0.05).to(dev),(torch.randn(13,64,generator=g)*0.2).to(dev),(torch.randn(13,generator=g)*0.1).to(dev)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
N=262144
# ray-like coherent points: 2048 rays x 128 samples along z
r=torch.rand(2048,2,generator=g)*1.0-0.5
t=torch.linspace(-0.6,0.6,128)
pts=torch.cat([r[:,None,:].expand(-1,128,-1), t[None,:,None].expand(2048,-1,1)],-1).reshape(-1,3).contiguous().to(dev)
rnd=(torch.rand(N,3,generator=g)*2-1).to(dev)
d=[torch.randn(N,device=dev),torch.randn(N,3,device=dev),torch.randn(N,13,device=dev),torch.randn(N,device=dev)*1e-3]
gt=torch.zeros(cfg.n_params,device=dev)
for name,p in [("coherent",pts),("random",rnd)]:
  for act in [4,7]:
    f=timeit(lambda: ops.sdf_fd_fwd(cfg,tab,mlp,p,1.0,0.02,act))
    b=timeit(lambda: ops.sdf_fd_bwd(cfg,tab,mlp,p,1.0,0.02,act,*d,grad_table=gt))
    s1=timeit(lambda: ops.sdf_fwd(cfg,tab,mlp,p,1.0,act,1))
    print(f"{name} active={act}: fd_fwd {f:.3f} ms  fd_bwd {b:.3f} ms  sdf_fwd(1) {s1:.3f} ms  -> {7*N/f/1e6:.1f} Meval/ms fwd")
big=(torch.rand(2097152,3,generator=g)*2-1).to(dev)
for act in [4,7,10]:
    s1=timeit(lambda: ops.sdf_fwd(cfg,tab,mlp,big,1.0,act,1),5)
    print(f"export chunk 2M pts active={act}: {s1:.3f} ms = {2097152/s1/1e6:.2f} Gpts/s; alg bytes {2097152*(act*32+16)/s1/1e6:.1f} GB/s")
