"""MI355X-native trainer behind the reference's ``solver.Solver`` API (reference solver.py:22-413).

Drop-in for ``from solver import Solver`` in the reference's train.py: same constructor, the
same ``dis_update`` / ``gen_update`` / ``smooth_moving`` / ``update_learning_rate`` /
``update_attention_status`` / ``sample`` / ``copy_nets`` / ``resume`` / ``save`` /
``init_network`` methods and the same ``loss_*`` attributes, with the generator and
discriminator running on the hand-written gfx950 kernels (``networks.networks{,_v2}``).

Differences from the reference that do not change any number it produces:
  * work the reference computes and then discards is not computed: the generator is not
    back-propagated in the D step (its gradients are zeroed at solver.py:153 before use), D's
    weight gradients are not formed in the G step (zeroed at solver.py:319 before use), and the
    ``x_fake2`` decode (only ever used detached, solver.py:181) keeps no tape;
  * images travel as NHWC4 buffers end to end (plane 3 is padding or the attention map).
"""
import contextlib
import copy
import os

import torch
from torch import nn

from gmm import gmm_kl_distance_sp, gmm_earth_mover_distance_sp
from hipdwc import host, ops
from hipdwc.host import get_scheduler, moving_average, weights_init
from hipdwc.optim import FusedAdam, FusedEMA
from networks.networks import MsImageDis
from networks.networks_v2 import AdaINGen_v2, flat_heads
from tools import dist_sampling_split

from vocab import Vocab


@contextlib.contextmanager
def _frozen(module):
    """Run a block with ``module``'s parameters excluded from autograd (their gradients would be discarded)."""
    flags = [(p, p.requires_grad) for p in module.parameters()]
    for p, _ in flags:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in flags:
            p.requires_grad_(f)


SAMPLE_CHUNK = 32                                                  # images per decode pass of Solver.sample()
DP_BUCKET_BYTES = 20 << 20                                          # hipdwc.dp.BUCKET_BYTES: D 3 buckets, G 4
GROUPED_DECODE = int(os.environ.get("DWC_GROUPED_DECODE", "1"))    # 0: always decode torch.cat([content] * groups) (ResBlock._forward_groups off)

class Solver(nn.Module):
    def __init__(self, configs, device=None, pretrained_embed=None):
        super().__init__()
        self.device = device if device is not None else torch.device("cpu")
        self.configs = configs
        self.vocab = Vocab(dataset=configs["dataset"])
        self.gen = AdaINGen_v2(configs["input_dim"], self.vocab, configs["gen"], pretrained_embed=pretrained_embed)
        self.dis = MsImageDis(configs["input_dim"], configs["dis"], self.device)
        self.instancenorm = nn.InstanceNorm2d(512, affine=False)
        self.print_network(self.dis, "D")
        self.print_network(self.gen, "G")

        self.num_cls, self.c_dim = configs["gen"]["num_cls"], configs["c_dim"]
        self.dist_mode = configs["dist_mode"]
        self.use_attention = configs["gen"]["use_attention"]
        self.att_status = self.use_attention
        self.ds_iter = configs["ds_iter"]
        self.display_size = int(configs["display_size"])
        self.dataset, self.stddev = configs["dataset"], configs["stddev"]
        self.sigma = torch.tensor(self.stddev ** 2).to(self.device)
        self.d_reg_every, self.rnd_step = 16, 3
        self.init_ds_w = configs["ds_w"]
        self.lr_policy = configs["lr_policy"]
        self.grad_sync = None      # multi-GPU data parallel: hipdwc.dp.GradAllReduce (simple) or, via enable_data_parallel(),
        self._reducers = None      # ... one hipdwc.dp.OverlappedGradReducer per optimiser (all-reduce overlapped with backward)
        self._ema = None
        self._gen_steps = 0            # bumped whenever G's parameters change: validity of the cached content code
        self._content_cache = None
        # D updates per G update (reference train.py:31,105 keeps it in `opts`, out of the solver's sight).  dis_update tapes
        # enc_content(x_real) for the gen_update that follows ONLY on the iterations that have one: tell the solver with
        # set_n_critic(n) (or dis_update(..., tape_content=bool) per call).  A caller that runs n_critic > 1 without doing
        # either is detected (a taped content code left unconsumed: dropped at once, taping off until the next
        # set_n_critic / resume / init_network) -- never wrong, only one wasted taped encode.
        self.n_critic = 1
        self._tape_content = True
        # (r04 measured the text encoder on a side stream beside the convolution work that does not depend on it: nothing beyond
        # run-to-run spread -- its persistent LSTM workgroups want 84 KB of LDS and wait for convolution workgroups to retire; removed)
        adam = dict(lr=configs["lr"], betas=(configs["beta1"], configs["beta2"]), weight_decay=configs["weight_decay"])
        # torch.optim.Adam subclasses (same param_groups / state_dict / scheduler interface) whose step()
        # is one multi-tensor HIP launch per network
        self.dis_opt = FusedAdam([p for p in self.dis.parameters() if p.requires_grad], **adam)
        self.gen_opt = FusedAdam([p for p in self.gen.parameters() if p.requires_grad], **adam)
        self.dis_scheduler = get_scheduler(self.dis_opt, configs)
        self.gen_scheduler = get_scheduler(self.gen_opt, configs)

        self.apply(weights_init(configs["init"]))      # whole solver first ...
        self.dis.apply(weights_init("gaussian"))       # ... then D again (reference solver.py:73-74)
        self.criterionL1 = torch.nn.L1Loss()

        if configs.get("vgg_w", 0) > 0:                 # reference solver.py:79-83
            from hipdwc.host import load_vgg16
            self.vgg = load_vgg16(configs["vgg_model_path"] + "/models")
            self.vgg = self.vgg.to(device) if device is not None else self.vgg
            self.vgg.eval()
            for param in self.vgg.parameters():
                param.requires_grad = False

    # ---- data parallel -----------------------------------------------------------------------
    def enable_data_parallel(self, group=None, bucket_bytes=DP_BUCKET_BYTES):
        """One process per GPU (torch.distributed initialised by the caller, backend nccl = RCCL): gradients of D / G become
        views of flat buckets that are all-reduced (averaged) as backward completes them (hipdwc.dp.OverlappedGradReducer)."""
        from hipdwc import dp
        self._reducers = {"dis": dp.OverlappedGradReducer(self.dis_opt.param_groups[0]["params"], group, bucket_bytes),
                          "gen": dp.OverlappedGradReducer(self.gen_opt.param_groups[0]["params"], group, bucket_bytes)}
        self.grad_sync = None
        # the persistent LSTM launches need ALL their workgroups resident while RCCL's all-reduce kernels (launched from inside
        # backward on a side stream) hold CUs of their own: keep those launches to half of a 256-CU device; larger grids take the
        # per-step kernels (dwc_lstm_seq_* return DWC_EINVAL above the cap)
        # (half of THIS device's CUs, not a constant: on a part with <= 128 CUs or in CPX mode "128" would cap nothing)
        if self.device.type == "cuda":
            ops.LSTM_SEQ_MAX_WORKGROUPS = max(1, torch.cuda.get_device_properties(self.device).multi_processor_count // 2)
        if os.environ.get("DWC_FORCE_DP") == "1":
            for r in self._reducers.values():
                r.force = True

    def _zero_grad(self, which):
        if self._reducers is None:
            (self.dis_opt if which == "dis" else self.gen_opt).zero_grad()
            return
        skip = ()
        if which == "gen" and not self.use_attention:      # the attention head is taken off the tape (Decoder.forward_nhwc4)
            skip = list(self.gen.dec.image_attention.parameters())
        self._reducers[which].prepare(skip)

    def _sync_grads(self, which):
        if self._reducers is not None:
            self._reducers[which].finish()
        elif self.grad_sync is not None:      # simple form: pack, all-reduce, unpack after backward
            self.grad_sync((self.dis_opt if which == "dis" else self.gen_opt).param_groups[0]["params"])

    # ---- bookkeeping -----------------------------------------------------------------------
    def set_n_critic(self, n):
        """D updates per G update of the caller's loop (reference train.py:105 `(iterations + 1) % opts.n_critic == 0`)."""
        self.n_critic = max(1, int(n))
        self._tape_content, self._content_cache = True, None

    def print_network(self, model, name):
        print("The number of parameters in {}: {}".format(name, sum(p.numel() for p in model.parameters())))

    def copy_nets(self):
        self.gen_copy = copy.deepcopy(self.gen)
        self.dis_copy = copy.deepcopy(self.dis)
        self._ema = None

    def smooth_moving(self):
        """EMA copies <- lerp(param, copy, 0.999) (reference solver.py:355-357, utils.py:52-54), one launch per net."""
        if not next(self.gen.parameters()).is_cuda:
            moving_average(self.gen, self.gen_copy)
            moving_average(self.dis, self.dis_copy)
            return
        if self._ema is None or not all(e.still_valid() for e in self._ema):
            self._ema = (FusedEMA(self.gen, self.gen_copy), FusedEMA(self.dis, self.dis_copy))
        for e in self._ema:
            e.step(0.999)

    def update_learning_rate(self):
        if self.lr_policy == "cosa":
            floor = self.configs["eta_min"]
            if self.dis_opt.param_groups[0]["lr"] == floor or self.gen_opt.param_groups[0]["lr"] == floor:
                self.configs["step_size"] *= self.configs["t_mult"]
                self.dis_scheduler = get_scheduler(self.dis_opt, self.configs)
                self.gen_scheduler = get_scheduler(self.gen_opt, self.configs)
        for sched in (self.dis_scheduler, self.gen_scheduler):
            if sched is not None:
                sched.step()

    def update_attention_status(self, iters):
        if self.att_status:
            self.use_attention = iters >= 10000

    # ---- small losses ------------------------------------------------------------------------
    def recon_criterion(self, x, y):
        if ops.is_image(x) and ops.is_image(y) and x.dtype == y.dtype:
            return ops.l1_mean(x, y, image=True)
        if x.dim() == 4:
            return ops.l1_mean(x, y)
        return torch.mean(torch.abs(x - y))

    def compute_vgg_loss(self, vgg, img, target):
        """mean((IN(vgg(img)) - IN(vgg(target)))^2) on relu5_3 (reference solver.py:242-247).  Images may be NCHW
        3-channel or the internal NHWC4 form (plane 3 ignored)."""
        from hipdwc.host import vgg_preprocess
        img_fea = vgg(vgg_preprocess(img))
        target_fea = vgg(vgg_preprocess(target))
        return torch.mean((ops.instance_norm(img_fea).float() - ops.instance_norm(target_fea).float()) ** 2)

    def criterion_l1(self, a, z):
        a, z = flat_heads(a), flat_heads(z)
        if a.is_cuda and a.dtype == z.dtype == torch.float32:
            return ops.l1_mean(a, z)                           # (r06: nn.L1Loss on [B, 64] was 3 + 8 launches per term)
        return self.criterionL1(a, z)

    def style_replace(self, c_src, c_trg, z_src, z_trg):
        keep = (c_src == c_trg).repeat_interleave(self.c_dim, dim=1)
        return torch.where(keep, z_src, z_trg)

    def _decode(self, content, style, x_real4, groups=1):
        """decode + (when enabled) the attention blend x*a + x_real*(1-a); NHWC4 in and out.  ``groups`` > 1: ``content`` is one copy
        of the batch, decoded with ``groups`` styles per sample (``style`` and ``x_real4`` hold groups * B samples, group-major): the
        decoder's first convolution then runs once instead of once per group (networks.ResBlock._forward_groups)."""
        # (r05, same box, alternating runs: c2 / bf16 1 344-1 346 images/s grouped against 1 338-1 340; c1 / fp32 343-346 against 345-352
        # -- there the batch-16 launch of the shared convolution is a contraction-split one, no cheaper per image than the batch-48
        # launch it replaces, and the repeat / gradient sums come on top: grouped on the bf16 path only)
        heads = self._decode_heads(content, style, groups, self.use_attention)
        return ops.attention_blend(heads, x_real4) if self.use_attention else heads

    def _decode_heads(self, content, style, groups=1, attention_used=True):
        """The decoder's fused heads for ``groups`` styles per content code; the grouped form only where it is the faster one (see
        _decode), the concatenated decode elsewhere."""
        if groups > 1 and not (GROUPED_DECODE and ops.PRECISION == "bf16"):
            content, groups = torch.cat([content] * groups), 1
        return self.gen.decode_nhwc4(content, style, attention_used=attention_used, groups=groups)

    def forward(self, x_real, txt_src2trg, txt_lens):
        x4 = ops.pack_image(x_real)
        content, style_src, _ = self.gen.encode(x4)
        style_txt, _ = self.gen.encode_txt(flat_heads(style_src), txt_src2trg, txt_lens)
        return self._decode(content, flat_heads(style_txt), x4)[:, :3].float()

    # ---- penalties (reference solver.py:291-315) --------------------------------------------------
    def gradient_penalty(self, y, x):
        """(||dy/dx||_2 - 1)^2, mean over the batch (reference solver.py:291-304)."""
        dydx = torch.autograd.grad(outputs=y, inputs=x, grad_outputs=torch.ones_like(y), retain_graph=True, create_graph=True,
                                   only_inputs=True)[0]
        dydx = dydx.reshape(dydx.size(0), -1)
        return torch.mean((torch.sqrt(torch.sum(dydx ** 2, dim=1)) - 1) ** 2)

    def r1_penalty(self, y, x):
        """mean over the batch of (||dy/dx||_2^2)^2 -- the reference squares the squared norm (solver.py:306-315); kept."""
        dydx = torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y), create_graph=True, only_inputs=True)[0]
        dydx = dydx.reshape(dydx.size(0), -1)
        return torch.mean(torch.sum(dydx ** 2, dim=1) ** 2)

    # ---- D step (reference solver.py:317-353) ---------------------------------------------------
    def dis_update(self, x_real, c_src, c_trg, txt_src2trg, txt_lens, label_src, label_trg, configs, iters, tape_content=None):
        """``tape_content`` (not in the reference's signature): True / False = whether a gen_update on this batch follows this
        call (its enc_content(x_real) is then computed once, here, on the tape); None = decide from set_n_critic()."""
        self._zero_grad("dis")
        x4 = ops.pack_image(x_real)
        ops.lstm_status_poll(x4.device)     # persistent text-encoder kernels: raise if a hand-off of an earlier step timed out
        ops.ksplit_status_poll()            # contraction-split convolutions: the same for their tile hand-offs
        B = x4.shape[0]
        with torch.no_grad():
            style_real, _ = self.gen.enc_style(x4)              # draw: mapping dropout (same order as gen.encode)
        # The content code of x_real is a deterministic function of x_real and G, and G does not change between this step
        # and the generator step that follows: computed ONCE, on the tape, and handed to gen_update (the reference
        # encodes x_real again there, solver.py:155, with identical values).
        if self._content_cache is not None:            # the last taped code was never consumed: no gen_update follows every
            self._tape_content = False                 # dis_update (n_critic > 1, reference train.py:105) -- stop taping
            self._content_cache = None                 # ... and release the unconsumed graph now
        tape = bool(tape_content) if tape_content is not None else (
            self._tape_content and (iters + 1) % max(1, int(self.n_critic)) == 0)
        with torch.no_grad():
            style_real = flat_heads(style_real)
            style1 = dist_sampling_split(c_trg, self.c_dim, self.stddev, self.device)      # (drawn before the text encoder's draws, as in the reference)
            style_txt, _ = self.gen.encode_txt(style_real, txt_src2trg, txt_lens)
        if tape:
            content_taped = self.gen.enc_content(x4)
            self._content_cache = (x_real, x_real._version, self._gen_steps, content_taped)
        else:
            self._content_cache = None
            with torch.no_grad():
                content_taped = self.gen.enc_content(x4)
        with torch.no_grad():
            content = content_taped.detach()
            # both fakes in ONE decoder pass (AdaIN parameters are per sample)
            fakes = self._decode(content, torch.cat([flat_heads(style_txt), style1]), torch.cat([x4, x4]), groups=2)
        gw, cw = configs["gan_w"], configs["cls_w"]
        # ONE discriminator pass over [x_fake, x_fake1, x_real]; D(x_real) enters both loss terms as
        # in the reference (which evaluates it twice, with identical values)
        outs = self.dis(torch.cat([fakes, x4]))
        if self.dis.gan_type == "lsgan" and self.dis.dataset in ("CelebA", "CUB200"):
            # calc_dis_loss(x_fake, x_real) + calc_dis_loss(x_fake1, x_real) (reference solver.py:333-336), one tail launch per scale
            self.loss_dis = self.dis.adv_loss(outs, B, label_src, targets=(0.0, 0.0, 1.0), w_src=(gw, gw, 2.0 * gw),
                                              w_cls=(0.0, 0.0, 2.0 * cw))
        else:
            o_fake, o_fake1, o_real = self.dis.split_outputs(outs, [B, B, B])
            self.loss_dis = self.dis.dis_loss_terms(o_fake, o_real, label_src, gw, cw) + \
                self.dis.dis_loss_terms(o_fake1, o_real, label_src, gw, cw)
        self.loss_dis_all = self.loss_dis
        # Gradient / R1 penalties (reference solver.py:337-350; gp_w 0 and use_r1 False in the shipped configuration).  Both differentiate
        # the first scale's src map w.r.t. its INPUT and then that gradient w.r.t. D's weights: a double backward, taken on stock torch
        # device ops over the same parameters (MsImageDis.forward_src_scale0_torch) -- the HIP Functions are once-differentiable.
        # The reference adds them IN PLACE to the tensor both names refer to (`self.loss_dis_all = self.loss_dis; ... += ...`): loss_dis
        # reads the penalised value afterwards too; kept.
        if configs["gp_w"] > 0.0:
            alpha = host.noise().rand((x_real.size(0), 1, 1, 1), x_real.device)
            x_hat = (alpha * x_real.detach()[:, :3].float() + (1 - alpha) * fakes[:B, :3].detach().float()).requires_grad_(True)
            self.loss_gp = self.gradient_penalty(self.dis.forward_src_scale0_torch(x_hat), x_hat) * configs["gp_w"]
            self.loss_dis_all = self.loss_dis = self.loss_dis_all + self.loss_gp
        if configs["use_r1"] and (iters + 1) % self.d_reg_every == 0:
            x_r = x_real.detach()[:, :3].float().requires_grad_(True)
            self.loss_r1 = self.r1_penalty(self.dis.forward_src_scale0_torch(x_r), x_r) * 10. / 2
            self.loss_dis_all = self.loss_dis = self.loss_dis_all + self.loss_r1
        self.loss_dis_all.backward()
        self._sync_grads("dis")             # data parallel: average D's gradients over the ranks
        self.dis_opt.step()

    # ---- G step (reference solver.py:151-240) ---------------------------------------------------
    def gen_update(self, x_real, c_src, c_trg, txt_src2trg, txt_lens, label_src, label_trg, configs, iters):
        self._zero_grad("gen")
        gen, cfg = self.gen, configs
        x4 = ops.pack_image(x_real)
        with _frozen(self.dis):
            # The reference runs its encodes/decodes one after another (solver.py:155-192).  The three
            # decodes of content_real and the three re-encodes are independent across samples, so
            # they run here as ONE 3B-sample pass each (larger GEMMs, a third of the launches); the
            # random draws are still made in the reference's order, the masks just get applied later.
            B = x4.shape[0]
            style_real, logvar = gen.enc_style(x4)                               # draw: mapping dropout
            s_real = flat_heads(style_real)
            mask_rec = gen.draw_encode_mask(B, x4.device)                        # draw of encode(x_real_rec)
            style_txt, logvar_txt = self.gen.encode_txt(s_real, txt_src2trg, txt_lens)     # draws of the text encoder
            cache, self._content_cache = self._content_cache, None
            if cache is not None and cache[0] is x_real and cache[1] == x_real._version and cache[2] == self._gen_steps:
                content_real = cache[3]                                          # taped in dis_update on the same batch and G
            else:
                content_real = gen.enc_content(x4)
            style1 = dist_sampling_split(c_trg, self.c_dim, self.stddev, self.device)
            style2 = dist_sampling_split(c_trg, self.c_dim, self.stddev, self.device)
            mask_rand = gen.draw_encode_mask(B, x4.device)                       # draw of encode(x_fake1)
            mask_fake = gen.draw_encode_mask(B, x4.device)                       # draw of encode(x_fake)

            with torch.no_grad():                                                # only ever used detached; independent of the text style
                x_fake2 = self._decode(content_real, style2, x4)
            s_txt = flat_heads(style_txt)
            # decode [within-domain reconstruction | text-driven fake | random-style fake]
            x_all = self._decode(content_real, torch.cat([s_real, s_txt, style1]), torch.cat([x4] * 3), groups=3)
            x_rec, x_fake, x_fake1 = torch.split(x_all, B)
            self.loss_ds = ops.l1_mean(x_fake1, x_fake2, image=True)
            self.init_ds_w = max(self.init_ds_w - 1 / 1e5, 0.0)
            masks = None if mask_rec is None else torch.cat([mask_rec, mask_fake, mask_rand])
            content_all, style_all, _ = gen.encode(x_all, drop_mask=masks)
            content_rec, content_fake, content_rand = torch.split(content_all, B)
            style_rec, style_fake, style_rand = torch.split(flat_heads(style_all), B)     # [B, K*c_dim] each
            cyc = cfg["recon_x_cyc_w"] > 0
            if cyc:
                x_cycle = self._decode(content_fake, s_real, x4)

            self.loss_gen_recon_x = ops.l1_mean(x_rec, x4, image=True)
            self.loss_gen_recon_c_real = ops.l1_mean(content_rec, content_real)
            self.loss_gen_recon_c_fake = ops.l1_mean(content_fake, content_real)
            self.loss_gen_recon_c_rand = ops.l1_mean(content_rand, content_real)
            self.loss_gen_recon_s_real = self.criterion_l1(style_rec, style_real)
            self.loss_gen_recon_s_fake = self.criterion_l1(style_fake, style_txt)
            self.loss_gen_recon_s_rand = self.criterion_l1(style_rand, style1)
            self.loss_gen_cycrecon_x = ops.l1_mean(x_cycle, x4, image=True) if cyc else 0

            outs = self.dis(x_all[B:])                                          # one pass over [x_fake, x_fake1]
            if self.dis.gan_type == "lsgan" and self.dis.dataset in ("CelebA", "CUB200"):
                self.loss_gen_adv = self.dis.adv_loss(outs, B, label_trg, targets=(1.0, 1.0), w_src=(cfg["gan_w"],) * 2,
                                                      w_cls=(cfg["cls_w"],) * 2)
            else:
                o_fake, o_fake1 = self.dis.split_outputs(outs, [B, B])
                self.loss_gen_adv = self.dis.gen_loss_terms(o_fake, label_trg, cfg["gan_w"], cfg["cls_w"]) + \
                    self.dis.gen_loss_terms(o_fake1, label_trg, cfg["gan_w"], cfg["cls_w"])

            if self.dist_mode == "kls":
                self.loss_kl_x = gmm_kl_distance_sp(style_real, logvar, c_src, self.sigma)
                self.loss_kl_trg = gmm_kl_distance_sp(style_txt, logvar_txt, c_trg, self.sigma)
            else:
                self.loss_kl_x = gmm_earth_mover_distance_sp(style_real, c_src)
                self.loss_kl_trg = gmm_earth_mover_distance_sp(style_txt, c_trg)
            self.loss_gen_vgg = 0
            if cyc and cfg["vgg_w"] > 0:                 # reference solver.py:221-223
                self.loss_gen_vgg = self.compute_vgg_loss(self.vgg, x4, x_cycle)

            # reference solver.py:226-238, as one weighted sum (one dot product forward, one scaled copy backward)
            self.loss_gen_total = ops.weighted_sum([
                (1.0, self.loss_gen_adv),
                (cfg["recon_x_w"], self.loss_gen_recon_x),
                (cfg["recon_c_w"], self.loss_gen_recon_c_real),
                (cfg["recon_c_w"], self.loss_gen_recon_c_fake),
                (cfg["recon_c_w"], self.loss_gen_recon_c_rand),
                (cfg["recon_s_w"], self.loss_gen_recon_s_real),
                (cfg["recon_s_w"], self.loss_gen_recon_s_fake),
                (cfg["recon_s_w"], self.loss_gen_recon_s_rand),
                (cfg["recon_x_cyc_w"], self.loss_gen_cycrecon_x),
                (cfg["kl_w"], self.loss_kl_x),
                (cfg["kl_w"], self.loss_kl_trg),
                (cfg["vgg_w"], self.loss_gen_vgg),
                (-self.init_ds_w, self.loss_ds)])
            self.loss_gen_total.backward()
        self._sync_grads("gen")             # data parallel: average G's gradients over the ranks
        self.gen_opt.step()
        self._gen_steps += 1

    # ---- visualisation path (reference solver.py:249-289) ----------------------------------------
    @torch.no_grad()
    def sample(self, x_real, txt_src2trg, txt_lens):
        """The reference walks the images one by one.  Here the convolutional work is batched -- ONE encode of all images and ONE
        decode of the 3 x B (reconstruction, text target, sampled style) codes; every op on that path is per sample (IN / AdaIN per
        (n, c), LayerNorm per n), so the results are the per-image ones.  What stays per image: the text encoder (its ``view``
        interleaves the samples of a batch, reference networks_v2.py:249 -- the reference calls it with batch 1 here) and the
        style draws (one ``dist_sampling_split`` per image, in the reference's order, so a seeded run consumes the same stream)."""
        self.eval()
        B = x_real.size(0)
        x4 = ops.pack_image(x_real)
        content, style_real, _ = self.gen.encode(x4)
        style_real = flat_heads(style_real)
        style_txt = torch.cat([flat_heads(self.gen.encode_txt(style_real[i:i + 1], txt_src2trg[i:i + 1], txt_lens[i:i + 1])[0])
                               for i in range(B)])
        sign = lambda s: torch.where(s.view(-1, self.num_cls, self.c_dim).mean(2) < 0, -1.0, 1.0)
        mus_real, mus_txt = sign(style_real), sign(style_txt)
        z = torch.cat([dist_sampling_split(mus_txt[i:i + 1], self.c_dim, self.stddev, self.device) for i in range(B)])
        z = self.style_replace(mus_real, mus_txt, style_real, z)
        # (through the same gate as the training step's decodes -- DWC_GROUPED_DECODE, bf16 only -- and SAMPLE_CHUNK images at a
        # time: a display batch of hundreds of images does not become one 3 x B decode)
        outs, atts = [], []
        for i in range(0, B, SAMPLE_CHUNK):
            j = min(B, i + SAMPLE_CHUNK)
            heads = self._decode_heads(content[i:j], torch.cat([style_real[i:j], style_txt[i:j], z[i:j]]), groups=3)
            out = ops.attention_blend(heads, torch.cat([x4[i:j]] * 3)) if self.use_attention else heads
            outs.append(out[:, :3].float().contiguous().view(3, j - i, 3, out.shape[2], out.shape[3]))
            if self.use_attention:
                atts.append((heads[j - i:2 * (j - i), 3:4].float().expand(-1, 3, -1, -1) - 0.5) / 0.5)
        outs = torch.cat(outs, dim=1)
        res = [x_real, outs[0], outs[1], outs[2]]
        if self.use_attention:
            res.append(torch.cat(atts))
        self.train()
        return res

    # ---- checkpoints (reference solver.py:359-413; file names/keys kept) ------------------------
    @staticmethod
    def _latest(dirname, key):
        if not os.path.exists(dirname):
            return None
        files = sorted(os.path.join(dirname, f) for f in os.listdir(dirname)
                       if os.path.isfile(os.path.join(dirname, f)) and key in f and ".pt" in f)
        return files[-1] if files else None

    def resume(self, checkpoint_dir, configs, load_optimizer=False):
        """Reference solver.py:359-376.  ``load_optimizer=True`` additionally restores Adam's moments and step counts from
        ``optimizer.pt`` — the state ``save`` writes and the reference forgets to reload (its lines 370-372 are commented
        out, so a resumed reference run restarts Adam from zero moments); off by default to keep the reference's behaviour."""
        name = self._latest(checkpoint_dir, "gen")
        self.gen.load_state_dict(torch.load(name, map_location="cpu")["a"])
        self._gen_steps, self._content_cache, self._tape_content = self._gen_steps + 1, None, True
        iterations = int(name[-15:-7]) if "avg" in name else int(name[-11:-3])
        name = self._latest(checkpoint_dir, "dis")
        self.dis.load_state_dict(torch.load(name, map_location="cpu")["b"])
        self.dis_scheduler = get_scheduler(self.dis_opt, configs, iterations)
        self.gen_scheduler = get_scheduler(self.gen_opt, configs, iterations)
        for _ in range(iterations):
            self.gen_scheduler.step()
            self.dis_scheduler.step()
        opt_path = os.path.join(checkpoint_dir, "optimizer.pt")
        if load_optimizer:
            if not os.path.exists(opt_path):
                raise FileNotFoundError("load_optimizer=True but %s does not exist" % opt_path)
            lr_g, lr_d = self.gen_opt.param_groups[0]["lr"], self.dis_opt.param_groups[0]["lr"]
            state = torch.load(opt_path, map_location="cpu")
            self.gen_opt.load_state_dict(state["gen"])
            self.dis_opt.load_state_dict(state["dis"])
            # the schedule position is the one derived from the iteration count above, not the saved group's
            self.gen_opt.param_groups[0]["lr"], self.dis_opt.param_groups[0]["lr"] = lr_g, lr_d
        print("Resume from iteration %d" % iterations)
        return iterations

    def init_network(self, gen_path, dis_path):
        gen_dict = torch.load(gen_path, map_location="cpu")["a"]
        dis_dict = torch.load(dis_path, map_location="cpu")["b"]
        sd = self.dis.state_dict()
        sd.update({k: v for k, v in dis_dict.items() if k in sd})
        self.dis.load_state_dict(sd)
        sg = self.gen.state_dict()
        sg.update({k: v for k, v in gen_dict.items() if k in sg and "embed_tokens" not in k})
        self.gen.load_state_dict(sg)
        self._gen_steps, self._content_cache, self._tape_content = self._gen_steps + 1, None, True
        print("Initial model loaded...")

    def save(self, snapshot_dir, iterations):
        tag = "%08d" % (iterations + 1)
        torch.save({"a": self.gen.state_dict()}, os.path.join(snapshot_dir, "gen_%s.pt" % tag))
        torch.save({"b": self.dis.state_dict()}, os.path.join(snapshot_dir, "dis_%s.pt" % tag))
        torch.save({"a": self.gen_copy.state_dict()}, os.path.join(snapshot_dir, "gen_%s_avg.pt" % tag))
        torch.save({"b": self.dis_copy.state_dict()}, os.path.join(snapshot_dir, "dis_%s_avg.pt" % tag))
        torch.save({"gen": self.gen_opt.state_dict(), "dis": self.dis_opt.state_dict()},
                   os.path.join(snapshot_dir, "optimizer.pt"))
