import os, sys, time, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd import ops
from drawingspinup_amd.mv.pipeline import build_random_pipeline
pipe = build_random_pipeline()
unet = pipe.unet
x = torch.randn(12,8,32,32,device='cuda').half(); ctx=torch.randn(12,1,768,device='cuda').half(); cl=torch.randn(12,10,device='cuda').half()
ts = torch.tensor([500],device='cuda')
for _ in range(2): unet(x, ts, ctx, cl)
rec = []
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
        r = f(*a, **k)
        e.record()
        if name == 'conv2d_nhwc_f16':
            xx, w = a[0], a[1]
            B,H,W,C = xx.shape; O = w.shape[0]; kk = k.get('k', a[3] if len(a)>3 else 3)
            st = k.get('stride',1); up = k.get('upsample2x', False)
            OH, OW = r.shape[1], r.shape[2]
            key = (name, B,H,W,C,O,kk,st,int(up)); fl = 2.0*B*OH*OW*O*kk*kk*C
        elif name in ('linear_f16', 'linear_geglu_f16'):
            xx, w = a[0], a[1]
            K = xx.shape[-1]; M = xx.numel() // K; N = w.shape[0]
            key = (name, M, K, N, int(k.get('transposed_tokens', 0) > 0)); fl = 2.0 * M * K * N
        elif name == 'mv_attention':
            q = a[0]; key = (name,)+tuple(q.shape)+tuple(a[1].shape); fl = 0
        else:
            key = (name,)+tuple(a[0].shape); fl = 0
        rec.append((key, s, e, fl))
        return r
    setattr(ops, name, g)
for n in ('conv2d_nhwc_f16','mv_attention','groupnorm_nhwc_f16','layernorm_f16','geglu_f16','linear_f16','linear_geglu_f16'):
    wrap(n)
# the mv modules may have imported the functions by name: patch there too
import drawingspinup_amd.mv.unet as U
torch.cuda.synchronize(); t=time.time()
unet(x, ts, ctx, cl)
torch.cuda.synchronize(); print("unet fwd (instrumented) %.2f ms"%((time.time()-t)*1e3))
agg = collections.OrderedDict()
for key, s, e, fl in rec:
    ms = s.elapsed_time(e)
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0]+=1; a[1]+=ms; a[2]+=fl
tot = collections.Counter()
for key,(c,ms,fl) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    tot[key[0]] += ms
    print(f"{key!s:70s} x{c:3d} {ms:8.3f} ms  {ms/c*1e3:8.1f} us/call  {fl/ms/1e9 if ms>0 else 0:7.1f} TFLOP/s")
print(dict(tot))
