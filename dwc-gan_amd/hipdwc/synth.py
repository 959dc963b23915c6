"""Synthetic CelebA-shaped batches and the config dictionaries used by tests, bench and goldens.

The throughput metric and every parity run use synthetic data (the dataset is not
available offline).  Shapes and value ranges follow what the reference's loader
hands to ``Solver.dis_update/gen_update``:

* images  fp32 [B,3,S,S] in [-1,1]            (reference data_loader.py:16 normalises to that range)
* labels  fp32 [B,8] in {0,1}                  (reference data_ios/celeba_data.py: 8 selected attributes)
* c = 2*label-1                                (reference tools.py:40-47 ``asign_label``)
* tokens  int64 [B,80]: BOS=1, words 4..101, EOS=2, PAD=0   (reference vocab.py:177-185, celeba_data.py:98)
* lengths int64 [B] in 3..40

Everything is drawn from a private ``torch.Generator`` so the global CPU stream
(used for weight init, dropout masks and style samples) is left untouched.
"""
import copy

import torch

MAX_TXT_LEN = 80
VOCAB_SIZE = 102
PAD_IDX, BOS_IDX, EOS_IDX, UNK_IDX = 0, 1, 2, 3


def make_batch(batch_size, image_size, seed=1234, device=None):
    """One training batch as the tuple ``train.py`` builds (reference train.py:92-100)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    x_real = torch.rand(batch_size, 3, image_size, image_size, generator=g) * 2.0 - 1.0
    label_src = (torch.rand(batch_size, 8, generator=g) < 0.5).float()
    label_trg = (torch.rand(batch_size, 8, generator=g) < 0.5).float()
    lens = torch.randint(3, 41, (batch_size,), generator=g, dtype=torch.int64)
    words = torch.randint(4, VOCAB_SIZE, (batch_size, MAX_TXT_LEN), generator=g, dtype=torch.int64)
    pos = torch.arange(MAX_TXT_LEN).unsqueeze(0)
    txt = torch.where(pos < (lens.unsqueeze(1) - 1), words, torch.zeros_like(words))
    txt[:, 0] = BOS_IDX
    txt[torch.arange(batch_size), lens - 1] = EOS_IDX
    batch = {
        "x_real": x_real,
        "label_src": label_src,
        "label_trg": label_trg,
        "c_src": label_src * 2.0 - 1.0,
        "c_trg": label_trg * 2.0 - 1.0,
        "txt": txt,
        "txt_lens": lens,
    }
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch


# The shipped training configuration (reference configs/celeba_faces.yaml), restated as a
# dict so nothing has to read the reference tree at run time.
DEFAULT_CONFIG = {
    "dataset": "CelebA",
    "image_save_iter": 10000, "image_display_iter": 500, "display_size": 8,
    "snapshot_save_iter": 10000, "log_iter": 100,
    "max_iter": 1000000, "batch_size": 1, "weight_decay": 0.0001,
    "beta1": 0.5, "beta2": 0.999, "init": "kaiming", "lr": 0.0001,
    "lr_policy": "step", "step_size": 100000, "ds_iter": 800000,
    "eta_min": 0.0, "t_mult": 1, "gamma": 0.5, "stddev": 0.5,
    "gan_w": 1, "cls_w": 1, "ds_w": 1, "kl_w": 0.1,
    "recon_x_w": 10, "recon_s_w": 1, "recon_c_w": 1, "recon_x_cyc_w": 10,
    "vgg_w": 0.1, "gp_w": 0, "use_r1": False, "dist_w": 0.1, "dist_mode": "kls",
    "c_dim": 8, "v_dim": 1,
    "gen": {
        "dim": 64, "mlp_dim": 256, "c_dim": 8, "num_cls": 8, "activ": "relu",
        "style_downsample": 5, "content_downsample": 2, "n_res": 4,
        "pad_type": "reflect", "use_attention": True,
        "embed_dim": 300, "hidden_size": 300, "num_layers": 2,
        "dropout_in": 0.1, "dropout_out": 0.1, "use_map": True,
    },
    "dis": {
        "dim": 64, "norm": "none", "activ": "lrelu", "n_layer": 5,
        "gan_type": "lsgan", "num_scales": 2, "pad_type": "reflect",
        "num_cls": 8, "image_size": 128, "dataset": "CelebA",
    },
    "input_dim": 3, "num_workers": 2, "image_size": 128, "crop_size": 178,
    "use_pretrain": False,
}


def make_config(image_size=128, vgg_w=0.0, lstm_dropout=None, tiny=False):
    """Config for a parity/bench run.

    ``vgg_w`` defaults to 0 because the VGG weights cannot be fetched offline
    (reference utils.py:180-194).  ``lstm_dropout=0`` switches off the LSTM's internal
    inter-layer dropout (``gen.dropout_out``), the one random draw that cannot be
    replayed through a stock ``nn.LSTM`` on another device.
    """
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    cfg["vgg_w"] = vgg_w
    cfg["image_size"] = image_size
    cfg["dis"]["image_size"] = image_size
    if lstm_dropout is not None:
        cfg["gen"]["dropout_out"] = lstm_dropout
    if tiny:
        cfg["gen"].update({"dim": 8, "mlp_dim": 16, "embed_dim": 12, "hidden_size": 16,
                           "style_downsample": 4, "n_res": 2})
        cfg["dis"].update({"dim": 8, "n_layer": 3})
    return cfg
