"""Plane regularisers (reference scene/regulation.py:14-28), torch expressions kept for API parity; the
training loop uses the fused HIP kernel through GaussianModel.compute_regulation."""
import torch


def compute_plane_tv(t):
    b, c, h, w = t.shape
    h_tv = torch.square(t[..., 1:, :] - t[..., :h - 1, :]).sum()
    w_tv = torch.square(t[..., :, 1:] - t[..., :, :w - 1]).sum()
    return 2 * (h_tv / (b * c * (h - 1) * w) + w_tv / (b * c * h * (w - 1)))


def compute_plane_smoothness(t):
    """Mean squared second difference along dim -2 (regulation.py:22-28)."""
    h = t.shape[-2]
    d1 = t[..., 1:, :] - t[..., :h - 1, :]
    d2 = d1[..., 1:, :] - d1[..., :h - 2, :]
    return torch.square(d2).mean()
