#!/usr/bin/env python3
"""Which Python lines issue the stock-torch (aten) ops of one training iteration -- a TorchDispatchMode logger, for when
torch.profiler's with_stack returns no frames (development aid for the launch diet).

    python benchmarks/torch_op_sites.py [c1|c2]
"""
import collections
import contextlib
import io
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
import bench  # noqa: E402
from hipdwc import ops, host, synth  # noqa: E402

SKIP = ("aten.view", "aten.detach", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.slice",
        "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.as_strided", "aten.reshape", "aten.empty", "aten.split",
        "aten.unbind", "aten.narrow", "aten.is_", "aten.size", "aten.stride", "aten.chunk", "aten._local_scalar", "aten.lift",
        "aten.set_", "aten.resize_", "aten.new_empty", "aten.empty_like", "aten.record_stream", "aten.is_pinned", "aten._has")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
            if on_gpu or "zeros" in name or "full" in name or "arange" in name:
                frame = "?"
                for fs in reversed(traceback.extract_stack(limit=14)[:-1]):
                    if "/dwc-gan_amd/" in fs.filename and "torch_op_sites" not in fs.filename:
                        frame = "%s:%d %s" % (fs.filename.split("/dwc-gan_amd/")[-1], fs.lineno, fs.name)
                        break
                shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
                self.n[(name, frame, shp)] += 1
        return func(*args, **(kwargs or {}))


def main():
    conf = sys.argv[1] if len(sys.argv) > 1 else "c1"
    B = bench.CONFIGS[conf]["per_gpu_batch"]
    ops.set_precision(bench.CONFIGS[conf]["precision"])
    dev = torch.device("cuda:0")
    from solver import Solver
    cfg = synth.make_config(image_size=128)
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = Solver(cfg, dev, None).to(dev)
    trainer.copy_nets()
    host.set_noise(host.DeviceNoise())
    batch = synth.make_batch(B, 128, seed=1, device=dev)
    batch["txt_lens"] = batch["txt_lens"].cpu()
    for it in range(3):
        bench.run_iteration(trainer, batch, cfg, it)
    torch.cuda.synchronize()
    with Sites() as s:
        bench.run_iteration(trainer, batch, cfg, 3)
        torch.cuda.synchronize()
    print("aten ops on GPU tensors in one iteration (views excluded): %d" % sum(s.n.values()))
    for (name, frame, shp), n in s.n.most_common(90):
        print("%4d  %-34s %-52s %s" % (n, name, frame, shp))


if __name__ == "__main__":
    main()
