#!/usr/bin/env python3
"""Loss-trajectory parity of the HIP trainer against the trajectories recorded from the reference.

    python benchmarks/traj_parity.py s64_b4_default [steps]
    python benchmarks/traj_parity.py s128_b16_nolstmdrop [steps]

Same seed, same synthetic batch, same random stream (HostNoise) as tests/golden/make_golden.py
used for the reference run; prints |delta| of loss_dis_all / loss_gen_total per step.
"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd"))
from hipdwc import host, synth  # noqa: E402


def run(tag, steps=None, device="cuda:0", verbose=True):
    with open(os.path.join(REPO, "tests", "golden", "traj_%s.json" % tag)) as f:
        gold = json.load(f)
    from solver import Solver
    S, B = gold["S"], gold["B"]
    cfg = synth.make_config(image_size=S, lstm_dropout=gold["lstm_dropout"])
    rows = gold["rows"][:steps] if steps else gold["rows"]
    host.set_noise(host.HostNoise())
    out = []
    try:
        torch.manual_seed(gold["seed"])
        trainer = Solver(cfg, torch.device(device), None).to(device)
        trainer.copy_nets()
        batch = synth.make_batch(B, S, seed=gold["batch_seed"], device=device)
        batch["txt_lens"] = batch["txt_lens"].cpu()
        a = (batch["x_real"], batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"], batch["label_src"],
             batch["label_trg"], cfg)
        for it, want in enumerate(rows):
            trainer.dis_update(*a, it)
            trainer.gen_update(*a, it)
            trainer.smooth_moving()
            trainer.update_learning_rate()
            trainer.update_attention_status(it)
            d, g = float(trainer.loss_dis_all.detach()), float(trainer.loss_gen_total.detach())
            out.append((it, d, want["loss_dis_all"], g, want["loss_gen_total"]))
            if verbose and (it < 5 or it % 10 == 9):
                print("it %3d  dis %.6f (ref %.6f, d=%.2e)   gen %.6f (ref %.6f, d=%.2e)" % (
                    it, d, want["loss_dis_all"], abs(d - want["loss_dis_all"]), g, want["loss_gen_total"],
                    abs(g - want["loss_gen_total"])), flush=True)
    finally:
        host.set_noise(host.DeviceNoise())
    return out


if __name__ == "__main__":
    run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
