"""Label / style-sampling helpers with the reference's names (reference tools.py:40-70)."""
import torch

from hipdwc import host


def label2onehot(labels, dim):
    out = torch.zeros(labels.size(0), dim)
    out[torch.arange(labels.size(0)), labels.long()] = 1
    return out


def asign_label(label, c_dim=None, mode="CelebA", normalize=True):
    """{0,1} attribute labels -> {-1,+1} GMM component centres (one-hot first for index labels)."""
    out = label.clone() if mode in ("CelebA", "CUB200") else label2onehot(label, c_dim)
    return out * 2.0 - 1.0 if normalize else out


def dist_sampling_split(mu, c_dim=8, stddev=0.5, device=None):
    """Style sample from the label-selected mixture component: [B, A] centres -> [B, A*c_dim],
    attribute-major, each entry ~ N(mu[b, a], stddev^2).  The draw goes through the active
    noise source (device generator by default, host generator in parity mode)."""
    z = host.noise().style_sample(mu, c_dim, stddev)
    return z if device is None else z.to(device)


distribution_sampling = dist_sampling_split
