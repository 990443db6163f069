import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd import ops
dev='cuda'
cfg=ops.HashGridConfig()
g=torch.Generator().manual_seed(0)
tab=((torch.rand(cfg.n_entries,2,generator=g)*2-1)*0.1).half().to(dev)
mlp=[(torch.randn(64,23,generator=g)*0.3).to(dev),(torch.randn(64,generator=g)*0.05).to(dev),(torch.randn(13,64,generator=g)*0.2).to(dev),(torch.randn(13,generator=g)*0.1).to(dev)]
N=262144
r=torch.rand(2048,2,generator=g)*1.0-0.5
t=torch.linspace(-0.6,0.6,128)
pts=torch.cat([r[:,None,:].expand(-1,128,-1), t[None,:,None].expand(2048,-1,1)],-1).reshape(-1,3).contiguous().to(dev)
d=[torch.randn(N,device=dev),torch.randn(N,3,device=dev),torch.randn(N,13,device=dev),torch.randn(N,device=dev)*1e-3]
gt=torch.zeros(cfg.n_params,device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for act in (4,6):
    print("active",act,"bwd ms", timeit(lambda: ops.sdf_fd_bwd(cfg,tab,mlp,pts,1.0,0.02,act,*d,grad_table=gt)))
