import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd.nsr.system import OrthoNeuSSystem, OrthoData
dev='cuda'
ds = OrthoData.synthetic_sphere(1024, device=dev)
sysm = OrthoNeuSSystem(device=dev)
sysm.dataset = ds
for s in range(60): sysm.training_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
t=time.time()
pr.enable()
for s in range(300): sysm.training_step()
pr.disable()
torch.cuda.synchronize(); print("ms/step", (time.time()-t)/300*1e3)
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(22)
