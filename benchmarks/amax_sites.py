"""Which tensors still need a stand-alone dwc_absmax pass in one c1 iteration (nobody raised their slot while writing them): call site in
hipdwc/ops.py and size.  Development aid.  usage: python benchmarks/amax_sites.py"""
import collections, contextlib, io, os, sys, traceback
import torch
REPO = os.getcwd()
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
import bench
from hipdwc import ops, host, synth, _lib
dev = torch.device("cuda:0")
from solver import Solver
cfg = synth.make_config(image_size=128)
torch.manual_seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    trainer = Solver(cfg, dev, None).to(dev)
trainer.copy_nets()
host.set_noise(host.DeviceNoise())
batch = synth.make_batch(16, 128, seed=1, device=dev)
batch["txt_lens"] = batch["txt_lens"].cpu()
for it in range(3):
    bench.run_iteration(trainer, batch, cfg, it)
torch.cuda.synchronize()
cnt = collections.Counter()
lib = _lib.load()
real = lib.dwc_absmax
class Wrap:
    def __call__(self, ptr, n, slot, ep, st):
        frame = "?"
        for fs in reversed(traceback.extract_stack(limit=12)[:-1]):
            if "/dwc-gan_amd/" in fs.filename and fs.name not in ("amax_of",):
                frame = "%s:%d %s" % (fs.filename.split("/dwc-gan_amd/")[-1], fs.lineno, fs.name); break
        cnt[(frame, n)] += 1
        return real(ptr, n, slot, ep, st)
lib.dwc_absmax = Wrap()
bench.run_iteration(trainer, batch, cfg, 3)
torch.cuda.synchronize()
for (f, n), c in cnt.most_common(30):
    print("%3d  %10d elems (%.1f MB)  %s" % (c, n, n * 4 / 1e6, f))
