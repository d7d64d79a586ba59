import sys, traceback
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import numpy as np, torch
from detinit import det_init_, synth_cloud
from oracle import torch_ref
from tomosar2height_amd import TomoSAR2Height
from tomosar2height_amd.config import berlin_config, munich_config
dev=torch.device('cuda:0')
torch.backends.cuda.matmul.allow_tf32=False
cfg = berlin_config()
ref = det_init_(torch_ref.TomoSAR2Height(cfg), seed=21)
model = TomoSAR2Height(cfg); model.load_state_dict(ref.state_dict(), strict=True); model.to(dev)
cloud = synth_cloud(20000, seed=77); cloud[0,:3000,:2]=cloud[0,0,:2]
w = torch.randn(512,512,generator=torch.Generator().manual_seed(1))
pa_ref,_ = ref(input_cloud=cloud); (pa_ref.squeeze()*w).mean().backward()
pa,_ = model(input_cloud=cloud.to(dev)); (pa.squeeze()*w.to(dev)).mean().backward()
print('height rel', ((pa.detach().cpu()-pa_ref.detach()).abs().max()/pa_ref.abs().max()).item())
# also: torch_ref on GPU (pure torch ops on device) to separate "my kernels" from "GPU conv numerics"
ref_gpu = det_init_(torch_ref.TomoSAR2Height(cfg), seed=21).to(dev)
pg,_ = ref_gpu(input_cloud=cloud.to(dev)); (pg.squeeze()*w.to(dev)).mean().backward()
print('torch_ref-on-GPU height rel', ((pg.detach().cpu()-pa_ref.detach()).abs().max()/pa_ref.abs().max()).item())
rows=[]
for (k,p),(_,q),(_,r) in zip(model.named_parameters(), ref.named_parameters(), ref_gpu.named_parameters()):
    if p.grad is None: continue
    s=q.grad.abs().max().item()+1e-30
    rows.append((k, (p.grad.cpu()-q.grad).abs().max().item()/s, (r.grad.cpu()-q.grad).abs().max().item()/s))
rows.sort(key=lambda t:-t[1])
for r in rows[:12]: print('%-55s hip-vs-cpu %.2e   torchgpu-vs-cpu %.2e'%r)
try:
    from conftest import load_golden
except Exception: pass
try:
    m = det_init_(TomoSAR2Height(munich_config(use_image=True)), seed=8).to(dev); m.set_channels_last(True)
    with torch.no_grad(): m(input_cloud=cloud.to(dev), input_image=torch.randn(1,3,512,512,device=dev))
    print('munich cl ok')
except Exception: traceback.print_exc()
