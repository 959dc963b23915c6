"""Loop-level numpy definitions of the primitives.  TEST INFRASTRUCTURE ONLY (see dwcgan_oracle.py).

The reference delegates its arithmetic to PyTorch (pinned pytorch=0.4.1 in reference
environment.yaml:9; torch 2.10 CPU is what is available here).  These are the published
definitions of those operators written out as explicit loops in float64, used on small
cases to pin the torch-based restatement in dwcgan_oracle.py independently of any
library kernel (tests/test_oracle_golden.py).
"""
import numpy as np


def reflect(i, n):
    """ReflectionPad2d index rule (reference networks.py:530-531)."""
    if i < 0:
        i = -i
    if i >= n:
        i = 2 * (n - 1) - i
    return i


def conv2d_reflect(x, w, b, stride, pad):
    """y[n,o,p,q] = b[o] + sum_{c,r,s} x[n,c,refl(p*stride-pad+r),refl(q*stride-pad+s)] * w[o,c,r,s]
    (reference networks.py:580: conv(pad(x)))."""
    B, C, H, W = x.shape
    O, _, KH, KW = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = np.zeros((B, O, Ho, Wo), dtype=np.float64)
    for n in range(B):
        for p in range(Ho):
            for q in range(Wo):
                patch = np.zeros((C, KH, KW))
                for r in range(KH):
                    for s in range(KW):
                        patch[:, r, s] = x[n, :, reflect(p * stride - pad + r, H), reflect(q * stride - pad + s, W)]
                y[n, :, p, q] = (w.reshape(O, -1) @ patch.reshape(-1)) + (b if b is not None else 0.0)
    return y


def instance_norm(x, eps=1e-5):
    """biased variance, eps inside the sqrt (nn.InstanceNorm2d; reference networks.py:545)."""
    y = np.empty_like(x, dtype=np.float64)
    for n in range(x.shape[0]):
        for c in range(x.shape[1]):
            v = x[n, c].astype(np.float64)
            m = v.mean()
            var = ((v - m) ** 2).mean()
            y[n, c] = (v - m) / np.sqrt(var + eps)
    return y


def layer_norm_munit(x, gamma, beta, eps=1e-5):
    """unbiased std, eps added to the std (reference networks.py:744-751)."""
    y = np.empty_like(x, dtype=np.float64)
    for n in range(x.shape[0]):
        v = x[n].astype(np.float64)
        m = v.mean()
        std = np.sqrt(((v - m) ** 2).sum() / (v.size - 1))
        y[n] = (v - m) / (std + eps) * gamma[:, None, None] + beta[:, None, None]
    return y


def upsample_bilinear2x(x):
    """align_corners=False bilinear x2 (reference networks_v2.py:154): source coordinate
    (o+0.5)/2-0.5 clamped at 0, neighbour clamped at the last index."""
    B, C, H, W = x.shape
    y = np.zeros((B, C, 2 * H, 2 * W), dtype=np.float64)

    def taps(o, n):
        s = max((o + 0.5) / 2.0 - 0.5, 0.0)
        i0 = int(np.floor(s))
        i1 = min(i0 + 1, n - 1)
        return i0, i1, s - i0
    for p in range(2 * H):
        h0, h1, fh = taps(p, H)
        for q in range(2 * W):
            w0, w1, fw = taps(q, W)
            y[:, :, p, q] = (1 - fh) * ((1 - fw) * x[:, :, h0, w0] + fw * x[:, :, h0, w1]) + \
                fh * ((1 - fw) * x[:, :, h1, w0] + fw * x[:, :, h1, w1])
    return y
