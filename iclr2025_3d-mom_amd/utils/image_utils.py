"""Mirror of the reference's utils/image_utils.py:14-38."""
import torch


def mse(img1, img2):
    return ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)


@torch.no_grad()
def psnr(img1, img2, mask=None):
    """Per-image PSNR for images in [0,1]; `mask` (broadcast over the 3 channels) restricts the pixels."""
    if mask is not None:
        sel = (mask.flatten(1).repeat(3, 1) != 0)
        img1, img2 = img1.flatten(1)[sel], img2.flatten(1)[sel]
    err = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    out = 20 * torch.log10(1.0 / torch.sqrt(err.float()))
    if mask is not None and torch.isinf(out).any():
        out = out[~torch.isinf(out)]
    return out
