import contextlib, io, os, sys, collections
import torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
import bench
from hipdwc import host, synth
dev = torch.device("cuda:0")
from solver import Solver
cfg = synth.make_config(image_size=128)
torch.manual_seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Solver(cfg, dev, None).to(dev)
tr.copy_nets(); host.set_noise(host.DeviceNoise())
b = synth.make_batch(16, 128, seed=1, device=dev); b["txt_lens"] = b["txt_lens"].cpu()
def count(root):
    seen=set(); c=collections.Counter(); st=[root]
    while st:
        n=st.pop()
        if n is None or n in seen: continue
        seen.add(n); c[type(n).__name__]+=1
        for nx,_ in n.next_functions: st.append(nx)
    return c
# monkeypatch backward to capture the loss tensors
caps=[]
orig=torch.Tensor.backward
def bw(self,*a,**k):
    caps.append(count(self.grad_fn)); return orig(self,*a,**k)
torch.Tensor.backward=bw
bench.run_iteration(tr,b,cfg,0)
for i,c in enumerate(caps):
    print("backward call",i, "nodes", sum(c.values()))
    for k,v in c.most_common(30): print("   %-40s %d"%(k,v))
