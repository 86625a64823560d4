"""Losses of the 4DGS loop (reference utils/loss_utils.py:23-92).  l1_loss runs as one fused HIP pass that also
produces the gradient image and the squared-error sum PSNR needs; ssim keeps the reference's definition
(11x11 gaussian window, sigma 1.5, zero padding, C1=0.01^2, C2=0.03^2)."""
from math import exp

import torch

from .. import ops

_last_sums = {"sums": None, "n": 0, "batch": 1}


def l1_loss(network_output, gt):
    loss, sums = ops.BACKEND.l1_loss_with_sums(network_output, gt)
    _last_sums.update(sums=sums, n=network_output.numel(), batch=network_output.shape[0] if network_output.dim() == 4 else 1)
    return loss


def psnr_from_last_l1():
    """PSNR of the image pair of the most recent l1_loss call (batch of 1), without re-reading the images."""
    s = _last_sums
    mse = s["sums"][1] / s["n"]
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def lpips_loss(img1, img2, lpips_model):
    """utils/loss_utils.py:20-22.  train_4DGS.py imports the name but its call is commented out (:89); the lpips package is not
    part of this path."""
    return torch.mean(lpips_model(img1, img2))


def gaussian(window_size, sigma):
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel):
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1, img2, window_size=11, size_average=True):
    """Reference utils/loss_utils.py:52-56.  The training loop only ever calls it with the defaults (train_4DGS.py:222),
    and that is what libmom4d's fused kernels implement (ops.ssim); other windows / per-image means have no HIP kernel
    and are refused rather than served by a torch fallback."""
    if window_size != 11 or not size_average:
        raise ops.N.MomError("ssim: libmom4d implements the reference's default window (11, sigma 1.5) with size_average=True only")
    return ops.BACKEND.ssim(img1, img2)
