#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (it needs /root/reference); the GPU box and the
test-suite only ever read the small .npz/.json files this script writes.  Nothing of
the reference's source is copied: the reference modules are imported, driven with
synthetic inputs, and their numeric outputs recorded.

    python tests/golden/make_golden.py ops tiny init      # seconds
    python tests/golden/make_golden.py traj64             # ~5 min
    python tests/golden/make_golden.py traj128            # ~50 min (S=128, B=16, 100 steps)
    python tests/golden/make_golden.py vgg                # seconds (perceptual loss, seeded random VGG16)

The reference's ``utils.py`` imports torchvision/torchfile at module top
(reference utils.py:23-29) although the training step never uses them; both are
absent here, so empty stand-in modules are registered before the import.  That is
harness plumbing and touches nothing on the path being recorded.
"""
import json
import os
import sys
import time
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REPO, "dwc-gan_amd", "hipdwc"))
import synth  # noqa: E402  (the product's synthetic-batch generator; plain python, no HIP)

warnings.filterwarnings("ignore")


def import_reference():
    for name in ("torchvision", "torchvision.transforms", "torchvision.utils", "torchfile"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "torchfile":
                m.load = lambda *a, **k: None
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.path.insert(0, REF)
    import solver as ref_solver          # noqa
    import networks.networks as ref_nets  # noqa
    import networks.networks_v2 as ref_v2  # noqa
    import gmm as ref_gmm                # noqa
    import tools as ref_tools            # noqa
    return ref_solver, ref_nets, ref_v2, ref_gmm, ref_tools


def t2n(t):
    return t.detach().cpu().numpy().copy()


def checksum(t):
    t = t.detach().double()
    return [float(t.sum()), float((t * t).sum())]


LOSS_NAMES = ["loss_dis", "loss_dis_all", "loss_ds", "loss_gen_adv", "loss_gen_cycrecon_x",
              "loss_gen_recon_c_fake", "loss_gen_recon_c_rand", "loss_gen_recon_c_real",
              "loss_gen_recon_s_fake", "loss_gen_recon_s_rand", "loss_gen_recon_s_real",
              "loss_gen_recon_x", "loss_gen_total", "loss_gen_vgg", "loss_kl_trg", "loss_kl_x"]


def read_losses(trainer):
    return {k: float(getattr(trainer, k)) for k in LOSS_NAMES}


# ----------------------------------------------------------------------------------------
# per-op goldens
# ----------------------------------------------------------------------------------------
def gen_ops(ref_nets, ref_gmm, ref_tools):
    out = {}
    g = torch.Generator().manual_seed(7)

    def rnd(*shape):
        return torch.randn(*shape, generator=g)

    # (name, B, Cin, Cout, H, k, s, p, norm, act)
    conv_cases = [
        ("stem7_relu", 2, 3, 8, 12, 7, 1, 3, "none", "relu"),
        ("down4_in_relu", 2, 8, 16, 12, 4, 2, 1, "in", "relu"),
        ("res3_in_none", 2, 16, 16, 6, 3, 1, 1, "in", "none"),
        ("res3_adain_relu", 3, 16, 16, 6, 3, 1, 1, "adain", "relu"),
        ("up5_ln_relu", 3, 16, 8, 10, 5, 1, 2, "ln", "relu"),
        ("up5_ln_relu_b1", 1, 16, 8, 10, 5, 1, 2, "ln", "relu"),
        ("head7_tanh", 2, 8, 3, 12, 7, 1, 3, "none", "tanh"),
        ("head7_sigmoid", 2, 8, 1, 12, 7, 1, 3, "none", "sigmoid"),
        ("dis4_lrelu", 2, 3, 8, 12, 4, 2, 1, "none", "lrelu"),
        ("dis4_lrelu_2x2", 2, 8, 8, 2, 4, 2, 1, "none", "lrelu"),
    ]
    for (name, B, ci, co, H, k, s, p, norm, act) in conv_cases:
        blk = ref_nets.Conv2dBlock(ci, co, k, s, p, norm=norm, activation=act, pad_type="reflect")
        with torch.no_grad():
            blk.conv.weight.copy_(rnd(co, ci, k, k) * 0.2)
            blk.conv.bias.copy_(rnd(co) * 0.1)
            if norm == "ln":
                blk.norm.gamma.copy_(torch.rand(co, generator=g))
                blk.norm.beta.copy_(rnd(co) * 0.1)
        x = rnd(B, ci, H, H).requires_grad_(True)
        extra = {}
        if norm == "adain":
            aw = (rnd(B * co) * 0.5 + 1.0).requires_grad_(True)
            ab = (rnd(B * co) * 0.5).requires_grad_(True)
            blk.norm.weight, blk.norm.bias = aw, ab
        y = blk(x)
        gy = rnd(*y.shape)
        (y * gy).sum().backward()
        rec = {"x": x, "w": blk.conv.weight, "b": blk.conv.bias, "y": y, "gy": gy,
               "dx": x.grad, "dw": blk.conv.weight.grad, "db": blk.conv.bias.grad}
        if norm == "ln":
            rec.update({"gamma": blk.norm.gamma, "beta": blk.norm.beta,
                        "dgamma": blk.norm.gamma.grad, "dbeta": blk.norm.beta.grad})
        if norm == "adain":
            rec.update({"aw": aw, "ab": ab, "daw": aw.grad, "dab": ab.grad})
        for kk, v in rec.items():
            out["conv/%s/%s" % (name, kk)] = t2n(v)
        out["conv/%s/meta" % name] = np.array([B, ci, co, H, k, s, p], dtype=np.int64)

    # bilinear x2 upsample (reference networks_v2.py:154) and x0.5 pyramid (networks.py:113)
    x = rnd(2, 4, 5, 6).requires_grad_(True)
    y = torch.nn.Upsample(scale_factor=2, mode="bilinear")(x)
    gy = rnd(*y.shape)
    (y * gy).sum().backward()
    for kk, v in {"x": x, "y": y, "gy": gy, "dx": x.grad}.items():
        out["up2/%s" % kk] = t2n(v)
    x = rnd(2, 3, 8, 12).requires_grad_(True)
    y = torch.nn.functional.interpolate(x, scale_factor=0.5, mode="bilinear")
    gy = rnd(*y.shape)
    (y * gy).sum().backward()
    for kk, v in {"x": x, "y": y, "gy": gy, "dx": x.grad}.items():
        out["down2/%s" % kk] = t2n(v)

    # GMM KL (reference gmm.py:13-22) and the L1 variant (gmm.py:33-41)
    mus = [rnd(5, 8) for _ in range(8)]
    lvs = [rnd(5, 8) * 0.3 for _ in range(8)]
    c = (torch.rand(5, 8, generator=g) < 0.5).float() * 2 - 1
    out["gmm/mus"] = t2n(torch.stack(mus))
    out["gmm/logvars"] = t2n(torch.stack(lvs))
    out["gmm/c"] = t2n(c)
    out["gmm/kl"] = np.array(float(ref_gmm.gmm_kl_distance_sp(mus, lvs, c, torch.tensor(0.25))))
    out["gmm/em"] = np.array(float(ref_gmm.gmm_earth_mover_distance_sp(mus, c)))

    # style sampling layout (reference tools.py:65-70) under a fixed global seed
    torch.manual_seed(99)
    z = ref_tools.dist_sampling_split(c, 8, 0.5, torch.device("cpu"))
    out["sample/c"] = t2n(c)
    out["sample/z_seed99"] = t2n(z)

    # discriminator losses on a tiny D (reference networks.py:116-170)
    torch.manual_seed(5)
    dparams = {"n_layer": 3, "gan_type": "lsgan", "dim": 4, "norm": "none", "activ": "lrelu",
               "num_scales": 2, "pad_type": "reflect", "num_cls": 8, "image_size": 32,
               "dataset": "CelebA"}
    D = ref_nets.MsImageDis(3, dparams)
    xf, xr = rnd(3, 3, 32, 32), rnd(3, 3, 32, 32)
    lab = (torch.rand(3, 8, generator=g) < 0.5).float()
    ld = D.calc_dis_loss(xf, xr, lab, lab, 1.0, 1.0)
    lg = D.calc_gen_loss(xf, lab, 1.0, 1.0)
    for kname, v in D.state_dict().items():
        out["dis/sd/%s" % kname] = t2n(v)
    out["dis/x_fake"], out["dis/x_real"], out["dis/label"] = t2n(xf), t2n(xr), t2n(lab)
    out["dis/loss_dis"], out["dis/loss_gen"] = np.array(float(ld)), np.array(float(lg))
    outs = D(xf)
    for i, (src, cls) in enumerate(outs):
        out["dis/out%d_src" % i], out["dis/out%d_cls" % i] = t2n(src), t2n(cls)

    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **out)
    print("ops_golden.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------------------
# whole-solver goldens
# ----------------------------------------------------------------------------------------
def build_ref_solver(ref_solver, cfg, seed=1234):
    torch.manual_seed(seed)            # reference train.py:23
    trainer = ref_solver.Solver(cfg, torch.device("cpu"), None)
    trainer.copy_nets()                # reference train.py:87
    return trainer


def one_iteration(trainer, batch, cfg, it):
    """The body of the reference training loop (reference train.py:102-111), n_critic=1."""
    a = (batch["x_real"], batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"],
         batch["label_src"], batch["label_trg"], cfg, it)
    trainer.dis_update(*a)
    trainer.gen_update(*a)
    trainer.smooth_moving()
    trainer.update_learning_rate()
    trainer.update_attention_status(it)


def gen_tiny(ref_solver):
    cfg = synth.make_config(image_size=32, tiny=True)
    B = 3
    trainer = build_ref_solver(ref_solver, cfg)
    out = {}
    for k, v in trainer.gen.state_dict().items():
        out["init/gen/%s" % k] = t2n(v)
    for k, v in trainer.dis.state_dict().items():
        out["init/dis/%s" % k] = t2n(v)
    batch = synth.make_batch(B, 32, seed=4321)
    for k, v in batch.items():
        out["batch/%s" % k] = t2n(v)
    # state of the global CPU generator right after construction: lets a checker that is
    # handed the recorded initial weights continue the random stream where the reference did
    out["rng_state_after_init"] = torch.get_rng_state().numpy().copy()

    # module outputs at the initial weights, eval-free (training mode, but with the RNG
    # state saved/restored so the recorded step below starts from the post-init stream)
    rng = torch.get_rng_state()
    trainer.eval()  # dropout off for the deterministic module-level vectors
    with torch.no_grad():
        content, mus, lvs = trainer.gen.encode(batch["x_real"])
        style = torch.cat(mus, 1)
        tmu, tlv = trainer.gen.encode_txt(style, batch["txt"], batch["txt_lens"])
        img, att = trainer.gen.decode(content, style)
        d_out = trainer.dis(batch["x_real"])
    trainer.train()
    torch.set_rng_state(rng)
    out["mod/content"] = t2n(content)
    out["mod/style_mu"] = t2n(torch.stack(mus))
    out["mod/style_logvar"] = t2n(torch.stack(lvs))
    out["mod/txt_mu"] = t2n(torch.stack(tmu))
    out["mod/txt_logvar"] = t2n(torch.stack(tlv))
    out["mod/dec_img"], out["mod/dec_att"] = t2n(img), t2n(att)
    for i, (src, cls) in enumerate(d_out):
        out["mod/dis%d_src" % i], out["mod/dis%d_cls" % i] = t2n(src), t2n(cls)

    losses = []
    for it in range(3):
        one_iteration(trainer, batch, cfg, it)
        losses.append(read_losses(trainer))
        if it == 0:
            # gradients left on the parameters after the first iteration
            # (D grads: from gen_update's backward, the last to touch them — reference
            #  solver.py:239 leaves D weight grads behind; G grads: gen_update)
            for k, p in trainer.gen.named_parameters():
                if p.grad is not None:
                    out["grad_it0/gen/%s" % k] = t2n(p.grad)
            for k, v in trainer.gen.state_dict().items():
                out["after_it0/gen/%s" % k] = t2n(v)
            for k, v in trainer.dis.state_dict().items():
                out["after_it0/dis/%s" % k] = t2n(v)
            for k, v in trainer.gen_copy.state_dict().items():
                out["ema_it0/gen/%s" % k] = t2n(v)
    out["losses_json"] = np.frombuffer(json.dumps(losses).encode(), dtype=np.uint8)
    out["init_ds_w"] = np.array(trainer.init_ds_w)
    out["lr"] = np.array(trainer.gen_opt.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(HERE, "tiny_step.npz"), **out)
    print("tiny_step.npz written; losses it0:", losses[0]["loss_dis_all"], losses[0]["loss_gen_total"])

    # D-only gradient fixture: a dis_update from the same init, grads on D params
    trainer = build_ref_solver(ref_solver, cfg)
    a = (batch["x_real"], batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"],
         batch["label_src"], batch["label_trg"], cfg, 0)
    # run dis_update but capture grads before the optimiser step by wrapping step()
    grabbed = {}
    real_step = trainer.dis_opt.step

    def grab_then_step(*args, **kw):
        for k, p in trainer.dis.named_parameters():
            grabbed[k] = t2n(p.grad)
        return real_step(*args, **kw)
    trainer.dis_opt.step = grab_then_step
    trainer.dis_update(*a)
    np.savez_compressed(os.path.join(HERE, "tiny_dis_grads.npz"),
                        loss_dis=np.array(float(trainer.loss_dis_all)), **grabbed)
    print("tiny_dis_grads.npz written")


def gen_penalties(ref_solver):
    """dis_update of the imported reference with BOTH penalties on (reference solver.py:337-350; gp_w 0 / use_r1 False in the shipped
    configuration): tiny networks from the seed-1234 initialisation, the tiny fixture's batch, iteration 15 (so that (iters + 1) %
    d_reg_every == 0 and the R1 term is live).  Scalars and every D gradient, grabbed in front of dis_opt.step."""
    cfg = synth.make_config(image_size=32, tiny=True)
    cfg["gp_w"], cfg["use_r1"] = 10.0, True
    trainer = build_ref_solver(ref_solver, cfg)
    batch = synth.make_batch(3, 32, seed=4321)
    out = {"rng_state_after_init": torch.get_rng_state().numpy().copy()}
    grabbed = {}
    real_step = trainer.dis_opt.step

    def grab_then_step(*args, **kw):
        for k, p in trainer.dis.named_parameters():
            grabbed["grad/" + k] = t2n(p.grad)
        return real_step(*args, **kw)
    trainer.dis_opt.step = grab_then_step
    x_real = batch["x_real"].clone()                  # (the reference sets requires_grad on the tensor it is handed)
    trainer.dis_update(x_real, batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"], batch["label_src"],
                       batch["label_trg"], cfg, 15)
    out.update(grabbed)
    for k in ("loss_dis", "loss_dis_all", "loss_gp", "loss_r1"):
        out[k] = np.float64(float(getattr(trainer, k)))
    out["meta"] = np.frombuffer(json.dumps({"gp_w": 10.0, "use_r1": True, "iters": 15, "B": 3, "S": 32, "batch_seed": 4321,
                                            "seed": 1234}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "tiny_penalties.npz"), **out)
    print("tiny_penalties.npz written:", {k: float(out[k]) for k in ("loss_dis", "loss_dis_all", "loss_gp", "loss_r1")})


def gen_init_checksums(ref_solver):
    res = {}
    for S in (64, 128):
        cfg = synth.make_config(image_size=S)
        trainer = build_ref_solver(ref_solver, cfg)
        res["S%d" % S] = {
            "gen": {k: checksum(v) for k, v in trainer.gen.state_dict().items()},
            "dis": {k: checksum(v) for k, v in trainer.dis.state_dict().items()},
            "n_gen": sum(p.numel() for p in trainer.gen.parameters()),
            "n_dis": sum(p.numel() for p in trainer.dis.parameters()),
        }
    with open(os.path.join(HERE, "init_checksums.json"), "w") as f:
        json.dump(res, f)
    print("init_checksums.json written")


def gen_traj(ref_solver, S, B, steps, lstm_dropout, tag, threads=None):
    if threads:
        torch.set_num_threads(threads)
    cfg = synth.make_config(image_size=S, lstm_dropout=lstm_dropout)
    trainer = build_ref_solver(ref_solver, cfg)
    batch = synth.make_batch(B, S, seed=1234)
    rows, t0 = [], time.time()
    path = os.path.join(HERE, "traj_%s.json" % tag)
    for it in range(steps):
        one_iteration(trainer, batch, cfg, it)
        row = read_losses(trainer)
        row["init_ds_w"] = trainer.init_ds_w
        rows.append(row)
        if it % 5 == 0 or it == steps - 1:
            print("[%s] it %d  dis %.6f gen %.6f  (%.1fs)" % (
                tag, it, row["loss_dis_all"], row["loss_gen_total"], time.time() - t0), flush=True)
            with open(path, "w") as f:
                json.dump({"S": S, "B": B, "seed": 1234, "batch_seed": 1234,
                           "lstm_dropout": lstm_dropout, "torch": torch.__version__,
                           "threads": torch.get_num_threads(), "rows": rows}, f)



# ----------------------------------------------------------------------------------------
# full-size single iterations (the shapes tests/test_hip_parity.py::test_full_size_iteration_vs_reference checks)
# ----------------------------------------------------------------------------------------
FULL_DIS_GRADS = ("cnns_feat.0.0.conv.weight", "cnns_feat.0.4.conv.weight", "cnns_feat.1.2.conv.weight", "cnns_cls.0.weight")
FULL_GEN_GRADS = ("enc_content.model.0.conv.weight", "dec.model.0.model.1.model.0.conv.weight", "dec.model.2.conv.weight",
                  "dec.image_content.conv.weight", "mlp.model.2.fc.weight", "enc_style.model.3.conv.weight",
                  "enc_txt.lstm.weight_hh_l0", "enc_content.model.3.model.2.model.1.conv.weight")
FULL_SAMPLE = 8192


def sample_idx(n):
    """The element numbers of a flattened n-element tensor that a full-size fixture keeps (at most FULL_SAMPLE, evenly spread)."""
    if n <= FULL_SAMPLE:
        return np.arange(n)
    return (np.arange(FULL_SAMPLE, dtype=np.int64) * n) // FULL_SAMPLE


def grad_record(out, prefix, named):
    for k, g in named:
        flat = g.detach().reshape(-1)
        out["%s/%s/sample" % (prefix, k)] = t2n(flat[torch.from_numpy(sample_idx(flat.numel()))])
        d = flat.double()
        out["%s/%s/stats" % (prefix, k)] = np.array([float(d.abs().max()), float(d.sum()), float((d * d).sum())])


class DiskOffload:
    """saved_tensors_hooks that park every large tensor autograd saves for backward in a file (the unmodified reference at batch 64
    holds ~60 GB of them; this container has 62 GB and no swap).  Values are unchanged: the raw fp32 goes to disk, tensors that
    are at least a quarter exact zeros (ReLU outputs and their padded copies) as a bit mask + the non-zero values (lossless; the
    batch-128 G step would not fit the disk otherwise), and a tensor that autograd saves twice (the ReLU's output is also the next
    block's input) is written once."""

    def __init__(self, root, min_bytes=4 << 20, min_free=6 << 30):
        self.root, self.min_bytes, self.min_free, self.n, self.bytes, self.on_disk = root, min_bytes, min_free, 0, 0, 0
        self.seen = {}
        os.makedirs(root, exist_ok=True)

    def pack(self, t):
        if t.device.type != "cpu" or t.numel() * t.element_size() < self.min_bytes or t.dtype not in (torch.float32, torch.int64, torch.bool):
            return t
        key = (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype)
        hit = self.seen.get(key)
        if hit is not None and hit[1]() is not None:          # the same live tensor, already parked
            return hit[0]
        import shutil
        import weakref
        if shutil.disk_usage(self.root).free < self.min_free:
            raise RuntimeError("DiskOffload: less than %d GB free under %s" % (self.min_free >> 30, self.root))
        path = os.path.join(self.root, "t%06d.npy" % self.n)
        self.n += 1
        self.bytes += t.numel() * t.element_size()
        a = t.detach().contiguous().numpy()
        h = ("disk", path, None)
        if t.dtype == torch.float32:
            flat = a.reshape(-1)
            nz = flat.view(np.uint32) != 0                      # (bit pattern: -0.0 stays a value)
            if int(nz.sum()) * 4 < flat.size * 3:
                np.save(path, flat[nz])
                np.save(path + ".mask.npy", np.packbits(nz))
                self.on_disk += int(nz.sum()) * 4 + flat.size // 8
                h = ("disk", path, tuple(a.shape))
        if h[2] is None:
            np.save(path, a)
            self.on_disk += a.nbytes
        try:
            self.seen[key] = (h, weakref.ref(t))
        except TypeError:
            pass
        return h

    def unpack(self, h):
        if isinstance(h, tuple) and len(h) == 3 and h[0] == "disk":
            if h[2] is None:
                return torch.from_numpy(np.load(h[1]))
            n = int(np.prod(h[2]))
            nz = np.unpackbits(np.load(h[1] + ".mask.npy"), count=n).astype(bool)
            flat = np.zeros(n, dtype=np.float32)
            flat[nz] = np.load(h[1])
            return torch.from_numpy(flat.reshape(h[2]))
        return h


def gen_full(ref_solver, S, B, offload=False, dis_only=False):
    """One iteration of the imported reference at the shipped network sizes from the seed-1234 initialisation (which the HIP
    Solver reproduces bit for bit, tests/test_host_logic.py) on synth.make_batch(B, S, seed=11): the D-step scalars and sampled D
    gradients (grabbed in front of dis_opt.step), then all 16 scalars and sampled G gradients of the G step."""
    import shutil
    cfg = synth.make_config(image_size=S, lstm_dropout=0.0)
    trainer = build_ref_solver(ref_solver, cfg)
    batch = synth.make_batch(B, S, seed=11)
    a = (batch["x_real"], batch["c_src"], batch["c_trg"], batch["txt"], batch["txt_lens"],
         batch["label_src"], batch["label_trg"], cfg, 0)
    out = {"meta": np.frombuffer(json.dumps({"S": S, "B": B, "seed": 1234, "batch_seed": 11, "lstm_dropout": 0.0,
                                             "torch": torch.__version__, "threads": torch.get_num_threads(),
                                             "sample": FULL_SAMPLE}).encode(), dtype=np.uint8)}
    real_dstep, real_gstep = trainer.dis_opt.step, trainer.gen_opt.step

    def dstep(*args, **kw):
        dp = dict(trainer.dis.named_parameters())
        grad_record(out, "dgrad", [(k, dp[k].grad) for k in FULL_DIS_GRADS])
        return real_dstep(*args, **kw)

    def gstep(*args, **kw):
        gp = dict(trainer.gen.named_parameters())
        grad_record(out, "ggrad", [(k, gp[k].grad) for k in FULL_GEN_GRADS if gp[k].grad is not None])
        return real_gstep(*args, **kw)
    trainer.dis_opt.step, trainer.gen_opt.step = dstep, gstep
    root = "/tmp/dwc_offload_S%d_B%d" % (S, B)
    off = DiskOffload(root) if offload else None
    t0 = time.time()
    ctx = torch.autograd.graph.saved_tensors_hooks(off.pack, off.unpack) if off else __import__("contextlib").nullcontext()
    try:
        with ctx:
            trainer.dis_update(*a)
            out["loss_dis"], out["loss_dis_all"] = np.float64(float(trainer.loss_dis)), np.float64(float(trainer.loss_dis_all))
            print("[full S%d B%d] dis_update %.1fs loss_dis_all %.6f" % (S, B, time.time() - t0, float(trainer.loss_dis_all)), flush=True)
            if off:
                shutil.rmtree(root, ignore_errors=True)
                os.makedirs(root, exist_ok=True)
                off.seen.clear()
            if not dis_only:
                trainer.gen_update(*a)
                losses = read_losses(trainer)
                out["losses_json"] = np.frombuffer(json.dumps(losses).encode(), dtype=np.uint8)
                print("[full S%d B%d] gen_update %.1fs loss_gen_total %.6f" % (S, B, time.time() - t0, losses["loss_gen_total"]), flush=True)
    finally:
        if off:
            print("[full S%d B%d] offloaded %d tensors, %.1f GB (%.1f GB on disk)" % (S, B, off.n, off.bytes / 1e9, off.on_disk / 1e9), flush=True)
            shutil.rmtree(root, ignore_errors=True)
    path = os.path.join(HERE, "full_s%d_b%d%s.npz" % (S, B, "_dis" if dis_only else ""))
    np.savez_compressed(path, **out)
    print(path, "written:", len(out), "arrays")



# ----------------------------------------------------------------------------------------
# Conv2dBlock beyond the shipped configurations (reference networks.py:524-585: pad_type zero / replicate, norm bn, activation prelu / selu)
# ----------------------------------------------------------------------------------------
REACH_CASES = [  # (name, B, Cin, Cout, H, k, s, p, norm, act, pad_type)
    ("zero3_bn_prelu", 3, 8, 16, 10, 3, 1, 1, "bn", "prelu", "zero"),
    ("rep5_none_selu", 2, 8, 8, 12, 5, 1, 2, "none", "selu", "replicate"),
    ("zero3_in_lrelu", 2, 8, 16, 12, 3, 1, 1, "in", "lrelu", "zero"),
    ("rep3_ln_prelu", 2, 16, 8, 8, 3, 1, 1, "ln", "prelu", "replicate"),
    ("zero3_none_relu", 2, 8, 8, 9, 3, 1, 1, "none", "relu", "zero"),
]


def gen_reach(ref_nets):
    g = torch.Generator().manual_seed(31)
    rnd = lambda *shape: torch.randn(*shape, generator=g)
    out = {}
    for (name, B, ci, co, H, k, s, p, norm, act, pad) in REACH_CASES:
        blk = ref_nets.Conv2dBlock(ci, co, k, s, p, norm=norm, activation=act, pad_type=pad)
        with torch.no_grad():
            blk.conv.weight.copy_(rnd(co, ci, k, k) * 0.2)
            blk.conv.bias.copy_(rnd(co) * 0.1)
            if norm == "ln":
                blk.norm.gamma.copy_(torch.rand(co, generator=g))
                blk.norm.beta.copy_(rnd(co) * 0.1)
            if norm == "bn":
                blk.norm.weight.copy_(torch.rand(co, generator=g) + 0.5)
                blk.norm.bias.copy_(rnd(co) * 0.1)
            if act == "prelu":
                blk.activation.weight.fill_(0.3)
        x = rnd(B, ci, H, H).requires_grad_(True)
        y = blk(x)                                  # training mode: batch statistics, running statistics updated once
        gy = rnd(*y.shape)
        (y * gy).sum().backward()
        rec = {"x": x, "w": blk.conv.weight, "b": blk.conv.bias, "y": y, "gy": gy, "dx": x.grad, "dw": blk.conv.weight.grad,
               "db": blk.conv.bias.grad}
        if norm == "ln":
            rec.update({"gamma": blk.norm.gamma, "beta": blk.norm.beta, "dgamma": blk.norm.gamma.grad, "dbeta": blk.norm.beta.grad})
        if norm == "bn":
            rec.update({"bn_w": blk.norm.weight, "bn_b": blk.norm.bias, "dbn_w": blk.norm.weight.grad, "dbn_b": blk.norm.bias.grad,
                        "running_mean": blk.norm.running_mean, "running_var": blk.norm.running_var})
        if act == "prelu":
            rec.update({"dprelu": blk.activation.weight.grad})
        for kk, v in rec.items():
            out["%s/%s" % (name, kk)] = t2n(v)
    np.savez_compressed(os.path.join(HERE, "conv_reach.npz"), **out)
    print("conv_reach.npz:", len(out), "arrays")


def gen_vgg(ref_solver, ref_nets):
    """compute_vgg_loss of the reference (solver.py:242-247) on a seeded, randomly initialised Vgg16 (the trained
    weights cannot be fetched here): loss, gradient w.r.t. the target image, relu5_3 features of the first image.
    The 59 MB of weights are not stored: both sides rebuild them from the seed; per-tensor checksums confirm it."""
    torch.manual_seed(777)
    vgg = ref_nets.Vgg16()
    vgg.eval()
    for prm in vgg.parameters():
        prm.requires_grad = False
    cfg = synth.make_config(image_size=64)
    trainer = build_ref_solver(ref_solver, cfg)           # only for its compute_vgg_loss method / instancenorm
    g = torch.Generator().manual_seed(778)
    out = {}
    for S, B in ((32, 2), (64, 1)):
        img = torch.rand(B, 3, S, S, generator=g) * 2 - 1
        target = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).requires_grad_(True)
        loss = trainer.compute_vgg_loss(vgg, img, target)
        loss.backward()
        with torch.no_grad():
            fea = vgg(ref_solver.vgg_preprocess(img, torch.device("cpu")))
        tag = "s%d" % S
        out[tag + "_img"], out[tag + "_target"] = t2n(img), t2n(target)
        out[tag + "_loss"] = np.float64(loss.item())
        out[tag + "_dtarget"] = t2n(target.grad)
        out[tag + "_fea"] = t2n(fea)
    np.savez_compressed(os.path.join(HERE, "vgg_loss.npz"), **out)
    with open(os.path.join(HERE, "vgg_init_checksums.json"), "w") as f:
        json.dump({"seed": 777, "torch": torch.__version__, "tensors": {k: checksum(v) for k, v in vgg.state_dict().items()}}, f)
    print("vgg_loss.npz written:", {k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items()})


def gen_sample(ref_solver):
    """Solver.sample (reference solver.py:249-289) on the tiny configuration: the five output stacks, with attention on
    (iteration-0 state) and off, from the seeded initial weights and a seeded CPU random stream."""
    cfg = synth.make_config(image_size=32, tiny=True)
    trainer = build_ref_solver(ref_solver, cfg)
    batch = synth.make_batch(3, 32, seed=4321)
    out = {}
    for tag, att in (("att", True), ("noatt", False)):
        trainer.use_attention = att
        torch.manual_seed(555)
        with torch.no_grad():
            res = trainer.sample(batch["x_real"], batch["txt"], batch["txt_lens"])
        out[tag + "/n"] = np.array(len(res))
        for i, r in enumerate(res):
            out["%s/%d" % (tag, i)] = t2n(r)
    np.savez_compressed(os.path.join(HERE, "sample_tiny.npz"), **out)
    print("sample_tiny.npz written:", {k: v.shape for k, v in out.items()})


def gen_text(ref_solver):
    """Input-pipeline fixtures (reference vocab.py, data_ios/celeba_text.py, data_ios/celeba_data.py:40-77):
    the CelebA vocabulary in index order, sentences produced by the reference's labels2text for seeded label pairs
    together with their ListsToTensor encoding, and the train/test split of a synthetic attribute file."""
    import random
    sys.path.insert(0, REF)
    import vocab as ref_vocab
    from data_ios import celeba_text as ref_text
    v = ref_vocab.Vocab(dataset="CelebA")
    random.seed(2024)
    rng = np.random.RandomState(7)
    sentences, pairs = [], []
    for _ in range(400):
        src, trg = (rng.rand(8) < 0.5).astype(np.int64), (rng.rand(8) < 0.5).astype(np.int64)
        if rng.rand() < 0.15:
            trg = src.copy()
        sentences.append(ref_text.labels2text(src.copy(), trg.copy()))
        pairs.append([src.tolist(), trg.tolist()])
    toks, lens = ref_vocab.ListsToTensor([s.split() for s in sentences], v, mx_len=80)
    # split: a synthetic list_attr file with 2500 entries and 40 attribute columns
    names = ["A%02d" % i for i in range(32)] + ["Black_Hair", "Blond_Hair", "Brown_Hair", "Smiling", "Young", "Male",
                                                "Eyeglasses", "No_Beard"]
    r2 = np.random.RandomState(11)
    lines = ["2500", " ".join(names)]
    for i in range(2500):
        vals = np.where(r2.rand(40) < 0.4, "1", "-1")
        lines.append("%06d.jpg %s" % (i + 1, " ".join(vals)))
    attr_txt = "\n".join(lines) + "\n"
    import tempfile
    from data_ios import celeba_data as ref_data
    with tempfile.TemporaryDirectory() as d:
        ap = os.path.join(d, "attr.txt")
        with open(ap, "w") as f:
            f.write(attr_txt)
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            ds = ref_data.CelebA(d, ap, names[32:], None, "train")
    with open(os.path.join(HERE, "text_pipeline.json"), "w") as f:
        json.dump({"itos": v.itos, "pairs": pairs, "sentences": sentences, "tokens": toks.tolist(), "lens": lens.tolist(),
                   "attr_seed": 11, "attr_names": names, "selected": names[32:],
                   "test_head": ds.test_dataset[:50], "train_head": ds.train_dataset[:50],
                   "n_test": len(ds.test_dataset), "n_train": len(ds.train_dataset),
                   "train_tail": ds.train_dataset[-5:]}, f)
    print("text_pipeline.json written:", len(sentences), "sentences, vocab", len(v.itos), "split", len(ds.test_dataset), len(ds.train_dataset))


if __name__ == "__main__":
    what = sys.argv[1:] or ["ops", "tiny", "init"]
    ref_solver, ref_nets, ref_v2, ref_gmm, ref_tools = import_reference()
    if "ops" in what:
        gen_ops(ref_nets, ref_gmm, ref_tools)
    if "tiny" in what:
        gen_tiny(ref_solver)
    if "init" in what:
        gen_init_checksums(ref_solver)
    if "penalties" in what:
        gen_penalties(ref_solver)
    if "reach" in what:
        gen_reach(ref_nets)
    if "vgg" in what:
        gen_vgg(ref_solver, ref_nets)
    if "sample" in what:
        gen_sample(ref_solver)
    if "text" in what:
        gen_text(ref_solver)
    if "traj64" in what:
        gen_traj(ref_solver, 64, 4, 100, None, "s64_b4_default")
        gen_traj(ref_solver, 64, 4, 100, 0.0, "s64_b4_nolstmdrop")
    if "traj64_threads4" in what:
        # the SAME reference run with another CPU thread count (different reduction order inside
        # the library kernels): measures how fast two fp32 evaluations of this GAN drift apart
        gen_traj(ref_solver, 64, 4, 100, None, "s64_b4_default_threads4", threads=4)
    if "traj128" in what:
        gen_traj(ref_solver, 128, 16, 100, 0.0, "s128_b16_nolstmdrop", threads=6)
    if "traj128_threads3" in what:
        # the 128x128 / batch-16 reference run again with another CPU thread count: the envelope the 0.15 / 0.06 bounds of
        # tests/test_trajectory.py::test_hip_trajectory_s128_b16_100_steps are derived from
        gen_traj(ref_solver, 128, 16, 100, 0.0, "s128_b16_nolstmdrop_threads3", threads=3)
    # full-size single iterations (one reference run each; the batch-64 one parks autograd's saved tensors on disk)
    if "full128b2" in what:
        gen_full(ref_solver, 128, 2)
    if "full256b1" in what:
        gen_full(ref_solver, 256, 1)
    if "full256b8" in what:
        gen_full(ref_solver, 256, 8)
    if "full256b2" in what:     # (+ full64b4: the shapes tests/test_bf16_parity.py::test_bf16_full_size_iteration_vs_reference runs)
        gen_full(ref_solver, 256, 2)
    if "full64b4" in what:
        gen_full(ref_solver, 64, 4)
    if "full128b64" in what:
        gen_full(ref_solver, 128, 64, offload=True)
    if "full128b128dis" in what:
        gen_full(ref_solver, 128, 128, offload=True, dis_only=True)
    if "full128b128" in what:   # D step + G step at BASELINE configs[2]'s batch; ~1 h, needs ~100 GB of /tmp
        gen_full(ref_solver, 128, 128, offload=True)
