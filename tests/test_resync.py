"""Per-step parity over many optimiser steps, with the chaotic divergence taken out (north star: "matched G/D losses
(+-1e-3 after 100 steps)").

Two fp32 evaluations of this GAN separate exponentially (tests/test_trajectory.py measures that on the reference
itself), so a free-running 100-step comparison cannot distinguish a systematic per-step error of the HIP path from
chaos.  Here the two are separated: before EVERY step the CPU oracle is re-synchronised to the HIP trainer's state
(all weights of G and D, Adam's first/second moments and step counts, the schedule position, the loss-weight decay), both
then run the SAME step from the SAME random stream, and all 16 loss scalars of that step are held to
1e-3 * max(1, |value|).  What is bounded is therefore the per-step error of the HIP kernels at the operating points the
HIP trajectory actually visits during training (weights after 1...100 Adam steps), which is the content of the north
star's tolerance; the accumulated drift of a free run is bounded separately, against the reference's own, in
test_trajectory.py.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hipdwc import host, ops, synth          # noqa: E402
from oracle import dwcgan_oracle as orc      # noqa: E402

DEV = "cuda:0"
SCALARS = ("loss_dis", "loss_dis_all", "loss_ds", "loss_gen_adv", "loss_gen_cycrecon_x", "loss_gen_recon_c_fake",
           "loss_gen_recon_c_rand", "loss_gen_recon_c_real", "loss_gen_recon_s_fake", "loss_gen_recon_s_rand",
           "loss_gen_recon_s_real", "loss_gen_recon_x", "loss_gen_total", "loss_gen_vgg", "loss_kl_trg", "loss_kl_x")


def _sync_oracle(oracle, trainer):
    """HIP trainer state -> oracle (weights, Adam moments and step counts, schedule, attention flag, ds weight)."""
    with torch.no_grad():
        for net, params, opt_o, opt_h in ((trainer.gen, oracle.gen, oracle.gen_opt, trainer.gen_opt),
                                          (trainer.dis, oracle.dis, oracle.dis_opt, trainer.dis_opt)):
            named = dict(net.named_parameters())
            assert set(named) == set(params)
            for k, p in named.items():
                params[k].data.copy_(p.detach().cpu())
                st = opt_h.state.get(p, {})
                if len(st):
                    opt_o.m[k].copy_(st["exp_avg"].detach().cpu())
                    opt_o.v[k].copy_(st["exp_avg_sq"].detach().cpu())
                    opt_o.t[k] = int(st["step"])
                else:
                    opt_o.m[k].zero_()
                    opt_o.v[k].zero_()
                    opt_o.t[k] = 0
    oracle.init_ds_w = trainer.init_ds_w
    oracle.use_attention = trainer.use_attention
    oracle.sched_steps = trainer.gen_scheduler.last_epoch


def _resync_run(S, B, steps, seed=1234, threads=16, precision="fp32", tol=1e-3, check=None):
    """`check`: the steps at which the oracle is synchronised and compared (default: every step).  The HIP trainer runs ALL
    `steps` steps either way, so a checked step k is the trainer's state after k of its own optimiser steps."""
    from solver import Solver
    ops.set_precision(precision)
    if threads:                                      # the oracle leg: oversubscribed hosts (128 threads) run this graph 5x slower
        torch.set_num_threads(min(threads, os.cpu_count() or 1))
    dev = torch.device(DEV)
    cfg = synth.make_config(image_size=S)            # the shipped configuration, every dropout on
    host.set_noise(host.HostNoise())
    try:
        torch.manual_seed(seed)
        trainer = Solver(cfg, dev, None).to(dev)
        trainer.copy_nets()
        batch = synth.make_batch(B, S, seed=seed)
        db = {k: v.to(dev) for k, v in batch.items()}
        oracle = orc.OracleSolver(cfg, {k: v.cpu() for k, v in trainer.gen.state_dict().items()},
                                  {k: v.cpu() for k, v in trainer.dis.state_dict().items()})
        oracle.copy_nets()
        worst = np.zeros(len(SCALARS))
        checked = sorted(set(range(steps)) if check is None else set(check))
        signed = np.zeros((len(checked), len(SCALARS)))
        for it in range(steps):
            if it in checked:
                _sync_oracle(oracle, trainer)
                rng = torch.get_rng_state()
                oracle.iteration(batch, it)
                torch.set_rng_state(rng)
            a = (db["x_real"], db["c_src"], db["c_trg"], db["txt"], db["txt_lens"], db["label_src"], db["label_trg"], cfg, it)
            trainer.dis_update(*a)
            trainer.gen_update(*a)
            trainer.smooth_moving()
            trainer.update_learning_rate()
            trainer.update_attention_status(it)
            torch.cuda.synchronize()
            if it not in checked:
                continue
            row = checked.index(it)
            for j, k in enumerate(SCALARS):
                got, want = float(getattr(trainer, k)), oracle.losses[k]
                signed[row, j] = (got - want) / max(1.0, abs(want))
                worst[j] = max(worst[j], abs(signed[row, j]))
                assert abs(got - want) <= tol * max(1.0, abs(want)), (it, k, got, want)
        return worst, signed
    finally:
        host.set_noise(host.DeviceNoise())
        ops.set_precision("fp32")


def test_hip_resync_100_steps_s64_b4():
    """100 optimiser steps at 64x64, batch 4 (BASELINE configs[0] shape): the 16 scalars of a step within 1e-3, checked at
    steps 0-7, every 8th step after that and step 99 (20 oracle iterations instead of 100: the GPU suite has a wall-clock
    limit and the CPU oracle is what it spends; r02 measured every one of the 100 steps at <= 9e-7, r04 checked 38)."""
    worst, signed = _resync_run(64, 4, 100, check=list(range(8)) + list(range(8, 96, 8)) + [99])
    print("worst |rel err| per scalar over 100 steps:", dict(zip(SCALARS, np.round(worst, 7))))
    # a systematic bias would show as a mean signed error comparable to the worst one; report and bound it
    bias = np.abs(signed.mean(axis=0))
    print("mean signed rel err per scalar:", dict(zip(SCALARS, np.round(signed.mean(axis=0), 8))))
    assert bias.max() <= 2e-4, bias


def test_hip_resync_24_steps_s128_b16():
    """24 optimiser steps at BASELINE configs[1] (128x128, batch 16), the headline configuration itself: the 16 scalars within
    1e-3 at 16 of the 24 steps -- 0-6, every second one after that, and 23 (each oracle iteration costs ~7 s of CPU at this size; r03
    checked six, r04 ten; the seconds come from the bf16 full-size cases, which read fixtures since r05)."""
    worst, signed = _resync_run(128, 16, 24, check=[0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16, 18, 20, 22, 23])
    print("worst |rel err| per scalar over 24 steps:", dict(zip(SCALARS, np.round(worst, 7))))
    # the signed-bias bound of the 64x64 run, at the headline shape (VERDICT r05 "weak" 1): a systematic error of the HIP step would
    # show as a mean signed error comparable to the worst one, rounding noise averages out
    print("mean signed rel err per scalar:", dict(zip(SCALARS, np.round(signed.mean(axis=0), 8))))
    assert np.abs(signed.mean(axis=0)).max() <= 2e-4, signed.mean(axis=0)


def test_hip_resync_bf16_24_steps_s64_b4():
    """The bf16 path (BASELINE configs[2]'s arithmetic) over 24 optimiser steps, re-synchronised like the fp32 runs above:
    the fp32 CPU oracle takes the bf16 trainer's state (fp32 master weights, Adam moments) before every step, both run
    the step, and every one of the 16 scalars must agree within 5e-3 * max(1, |value|) (bf16 activations carry 8
    significant bits, losses are fp32 means over >= 10^3 of them; measured r03: worst 1.4e-3 (loss_dis), all others <= 7e-4).  A whole-iteration
    comparison over one or two steps cannot see a systematic bias of the bf16 weight gradients working through Adam --
    this can: the state the oracle is synchronised TO is the one bf16 gradients produced, so a bias would move the
    operating point step after step, and the mean SIGNED error over the run (which averages rounding noise out and keeps
    a bias) is bounded at 1e-3 (measured: <= 2.6e-4)."""
    worst, signed = _resync_run(64, 4, 24, precision="bf16", tol=5e-3, check=list(range(6)) + list(range(6, 24, 3)))
    print("bf16 worst |rel err| per scalar over 24 steps:", dict(zip(SCALARS, np.round(worst, 5))))
    print("bf16 mean signed rel err per scalar:", dict(zip(SCALARS, np.round(signed.mean(axis=0), 6))))
    # drift of the error itself: the second half of the run must not be systematically worse than the first
    first, second = np.abs(signed[:6]).mean(axis=0), np.abs(signed[6:]).mean(axis=0)
    print("bf16 mean |rel err| first / second half:", dict(zip(SCALARS, zip(np.round(first, 5), np.round(second, 5)))))
    assert np.abs(signed.mean(axis=0)).max() <= 1e-3, signed.mean(axis=0)
    assert (second <= 3.0 * first + 1e-3).all(), (first, second)
