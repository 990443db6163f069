import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tools')
from drawingspinup_amd import ops
dev='cuda'; cfg=ops.HashGridConfig(); g=torch.Generator().manual_seed(0)
tab=((torch.rand(cfg.n_entries,2,generator=g)*2-1)*0.1).half().to(dev)
mlp=[(torch.randn(64,23,generator=g)*0.3).to(dev),(torch.randn(64,generator=g)*0.05).to(dev),(torch.randn(13,64,generator=g)*0.2).to(dev),(torch.randn(13,generator=g)*0.1).to(dev)]
N=266240
r=torch.rand(2080,2,generator=g)-0.5; t=torch.linspace(-0.6,0.6,128)
pts=torch.cat([r[:,None,:].expand(-1,128,-1), t[None,:,None].expand(2080,-1,1)],-1).reshape(-1,3).contiguous().to(dev)
d=[torch.randn(N,device=dev),torch.randn(N,3,device=dev),torch.randn(N,13,device=dev),None]
gt=torch.zeros(cfg.n_params,device=dev)
ps, perm = ops.spatial_sort(pts, 1.0, 6)
out = ops.sdf_fd_fwd(cfg,tab,mlp,ps,1.0,1/128,5,True,True,False,enc_cache=True,perm=perm)
def f(): ops.sdf_fd_bwd(cfg,tab,mlp,ps,1.0,1/128,5,d[0],d[1],d[2],None,grad_table=gt,enc_cache=out[4],perm=perm)
for _ in range(5): f()
torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
for _ in range(30): f()
e.record(); torch.cuda.synchronize(); print(os.path.basename(os.environ.get('DSU_HIP_LIB','default')), "sdf_fd_bwd pair (MLP part + scatter) %.1f us" % (s.elapsed_time(e)/30*1e3))
