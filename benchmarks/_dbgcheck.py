import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "dwc-gan_amd"))
from hipdwc import ops
ops.set_precision("bf16")
x = torch.randn(128, 256, 32, 32, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 256, 3, 3, device="cuda") * 0.05
with torch.no_grad():
    y = ops.conv2d(x, w, None, 1, 1, "none")
print("DBG", os.environ.get("DWC_HALO_DBG"), "mean|y|", float(y.float().abs().mean()))
