"""Mirror of the reference's utils/general_utils.py (functions on the hot path only), device-agnostic
(the reference hard-codes device="cuda" at general_utils.py:71,89,108)."""
import math
import random
import sys
from datetime import datetime

import numpy as np
import torch


def inverse_sigmoid(x):
    """logit (general_utils.py:18-19)."""
    return torch.log(x / (1 - x))


def PILtoTorch(pil_image, resolution=None):
    """general_utils.py:21-33: HWC uint8 -> CHW float in [0,1] (images whose max is 1 are left unscaled)."""
    img = pil_image if resolution is None else pil_image.resize(resolution)
    arr = np.array(img)
    ten = torch.from_numpy(arr) / 255.0 if arr.max() != 1 else torch.from_numpy(arr)
    return ten.permute(2, 0, 1) if ten.dim() == 3 else ten.unsqueeze(-1).permute(2, 0, 1)


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with optional warm-up (general_utils.py:35-68); fp64 host math."""
    log_a, log_b = (math.log(lr_init), math.log(lr_final)) if lr_init > 0 and lr_final > 0 else (None, None)

    def schedule(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        warm = 1.0
        if lr_delay_steps > 0:
            warm = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * min(max(step / lr_delay_steps, 0), 1))
        t = min(max(step / max_steps, 0), 1)      # (= np.clip(x, 0, 1) of the reference, without its 5 us per call: three schedules per iteration)
        return warm * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)

    schedule.log_endpoints = (log_a, log_b)
    return schedule


def strip_lowerdiag(L):
    """[N,3,3] -> [N,6] upper triangle xx,xy,xz,yy,yz,zz (general_utils.py:70-79)."""
    return torch.stack([L[:, 0, 0], L[:, 0, 1], L[:, 0, 2], L[:, 1, 1], L[:, 1, 2], L[:, 2, 2]], dim=1).float()


def strip_symmetric(sym):
    return strip_lowerdiag(sym)


def build_rotation(r):
    """Rotation matrices of NORMALISED quaternions (r,x,y,z) (general_utils.py:84-105)."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q.unbind(dim=1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def build_scaling_rotation(s, r):
    """L = R * diag(s) (general_utils.py:107-117)."""
    return build_rotation(r) * s[:, None, :]


def safe_state(silent):
    """Seeds python/numpy/torch with 0 and timestamps stdout lines (general_utils.py:119-139)."""
    out = sys.stdout

    class _Stamped:
        def write(self, x):
            if not silent:
                out.write(x.replace("\n", " [{}]\n".format(datetime.now().strftime("%d/%m %H:%M:%S"))) if x.endswith("\n") else x)

        def flush(self):
            out.flush()

    sys.stdout = _Stamped()
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    if torch.cuda.is_available():
        torch.cuda.set_device(torch.device("cuda:0"))
