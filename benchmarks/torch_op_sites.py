#!/usr/bin/env python3
"""Which Python call sites (forward) / autograd nodes (backward) issue the stock-torch device ops of one training iteration.

    python benchmarks/torch_op_sites.py [c1|c2]

torch.profiler (benchmarks/torch_kernel_sites.py) counts the launched kernels but records no Python stack for them on this build; this
tool runs one iteration under a TorchDispatchMode and logs every aten op that touches a device tensor and is not a pure view, with the
innermost frame inside this repository (forward) or the autograd node that is executing (backward).  Development aid for the launch diet.
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

VIEWS = {"view", "_unsafe_view", "reshape", "expand", "slice", "select", "t", "transpose", "permute", "detach", "alias", "as_strided",
         "unsqueeze", "squeeze", "split", "split_with_sizes", "unbind", "chunk", "narrow", "_reshape_alias", "view_as", "unfold",
         "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "set_", "resize_", "is_pinned", "_local_scalar_dense",
         "lift_fresh", "detach_", "record_stream", "_pin_memory", "is_same_size", "sym_size", "sym_stride", "sym_numel", "stride", "size",
         "dim", "numel", "is_contiguous", "storage_offset", "sym_storage_offset", "_has_compatible_shallow_copy_type", "result_type",
         "unsafe_split", "flatten", "contiguous", "unflatten", "movedim", "view_as_real", "_to_copy_noop"}


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in VIEWS:
            return out
        flat = [a for a in torch.utils._pytree.tree_leaves((args, kwargs, out)) if isinstance(a, torch.Tensor)]
        if not any(a.is_cuda for a in flat):
            return out
        shapes = [tuple(a.shape) for a in torch.utils._pytree.tree_leaves(args) if isinstance(a, torch.Tensor)][:3]
        node = torch._C._current_autograd_node()
        if node is not None:
            where = "bwd:" + type(node).__name__
        else:
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/dwc-gan_amd/" in fr.filename or fr.filename.endswith("/bench.py"):
                    where = "%s:%d %s" % (fr.filename.split("/dwc-gan_amd/")[-1], fr.lineno, fr.name)
                    break
        self.sites[(name, where, str(shapes))] += 1
        return out


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
    log = Log()
    with log:
        bench.run_iteration(trainer, batch, cfg, 3)
    torch.cuda.synchronize()
    print("non-view aten ops on device tensors in one iteration: %d" % sum(log.sites.values()))
    by_where = collections.Counter()
    for (name, where, shp), n in log.sites.items():
        by_where[where] += n
    print("---- by site")
    for where, n in by_where.most_common(70):
        print("%5d  %s" % (n, where))
    print("---- by (op, site, shapes)")
    for (name, where, shp), n in log.sites.most_common(120):
        print("%5d  %-26s %-52s %s" % (n, name, where[:52], shp[:70]))


if __name__ == "__main__":
    main()
