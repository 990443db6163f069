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
# round 2: the step evaluates the points in Morton order (ops.spatial_sort), eps = the finest cell
ACT = int(os.environ.get("PMC_ACTIVE", "5"))
EPS = 1.0 / 128
for _ in range(5):
    ps, perm = ops.spatial_sort(pts, 1.0, 6)
    out = ops.sdf_fd_fwd(cfg,tab,mlp,ps,1.0,EPS,ACT,True,True,False,enc_cache=True,perm=perm)
    ops.sdf_fd_bwd(cfg,tab,mlp,ps,1.0,EPS,ACT,d[0],d[1],d[2],None,grad_table=gt,enc_cache=out[4],perm=perm)
torch.cuda.synchronize()
