"""GMM style-space losses (same function names as reference gmm.py), evaluated for all
attributes in one batched expression instead of a Python loop over heads.

Each attribute i owns a c_dim-wide slice of the style vector whose target distribution is
N(c_i, sigma^2) with c_i = +-1 taken from the label (reference gmm.py:13-22, :33-41).
"""
import torch

from hipdwc import ops


def _stack(heads):
    """list of K tensors [B, D] -> [B, K, D].  A networks_v2.HeadList carries the [B, K*D] tensor its entries were cut from:
    that one is viewed instead of stacking K slices (one autograd edge instead of K)."""
    flat = getattr(heads, "flat", None)
    if flat is not None:
        return flat.reshape(flat.shape[0], len(heads), -1)
    return torch.stack(list(heads), dim=1)


def gmm_kl_distance(pred_mu, pred_sigma, mus, sigma):
    """KL(N(pred_mu, pred_sigma) || N(mus, sigma)) summed over dims, mean over the batch."""
    return (0.5 * (torch.log(sigma / pred_sigma) + (pred_sigma + (pred_mu - mus) ** 2) / sigma - 1.0)).sum(dim=1).mean()


def gmm_kl_distance_sp(pred_mus, pred_sigma, mus, sigma):
    """Per-attribute KL with log-variance heads: sum_i mean_n sum_d KL(N(mu, e^lv) || N(c_i, sigma))."""
    mu, lv = _stack(pred_mus), _stack(pred_sigma)          # [B, K, D]
    if mu.is_cuda and ops.GMM_FUSED and mu.dtype == torch.float32 and lv.dtype == torch.float32:
        return ops.gmm_kl_sp(mu, lv, mus, sigma)           # one launch each way (hipdwc.ops._GmmKlSp)
    var = lv.exp()
    centre = mus[:, :mu.shape[1]].unsqueeze(-1)            # [B, K, 1]
    kl = 0.5 * (torch.log(sigma / var) + (var + (mu - centre) ** 2) / sigma - 1.0)
    return kl.sum(dim=2).mean(dim=0).sum()


def gmm_earth_mover_distance(pred_mus, mus):
    return torch.abs(pred_mus - mus).sum(dim=1).mean()


def gmm_earth_mover_distance_sp(pred_mus, mus):
    mu = _stack(pred_mus)
    return (mu - mus[:, :mu.shape[1]].unsqueeze(-1)).abs().sum(dim=2).mean(dim=0).sum()
