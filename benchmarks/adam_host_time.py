import sys, os, time, io, contextlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
import torch, bench
from hipdwc import ops, host, synth
from solver import Solver
ops.set_precision("fp32")
dev = torch.device("cuda:0")
cfg = synth.make_config(image_size=128)
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Solver(cfg, dev, None).to(dev)
tr.copy_nets()
batch = synth.make_batch(16, 128, seed=1, device=dev); batch["txt_lens"] = batch["txt_lens"].cpu()
for it in range(3): bench.run_iteration(tr, batch, cfg, it)
torch.cuda.synchronize()
for name, opt in (("gen", tr.gen_opt), ("dis", tr.dis_opt)):
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); opt.step(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts.sort()
    print("%s_opt.step(): host %.0f us median, %.0f us min (%d tensors)" % (name, ts[10] * 1e6, ts[0] * 1e6, len(opt.param_groups[0]["params"])))
