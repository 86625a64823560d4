"""PyTorch-CPU restatement of the reference's torch-op sequence for the non-rasterizer part of the hot path --
TEST INFRASTRUCTURE ONLY (oracle).  Each function cites the reference lines it follows.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; tests of pure host logic on a
GPU-less machine install `TorchBackend` into the package explicitly (ops.set_backend) -- the product never
falls back to it.

Pinned against the real reference modules imported in the build container: oracle/ref_harness.py generates
tests/golden/*.npz from /root/reference, tests/test_oracle_torch.py checks this file against them.
"""
import itertools

import torch
import torch.nn.functional as F


def normalize_aabb(pts, aabb):
    # scene/hexplane.py:19-20
    return (pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0


def grid_sample_wrapper(grid, coords):
    # scene/hexplane.py:21-46 for 2-D planes
    if grid.dim() == 3:
        grid = grid.unsqueeze(0)
    if coords.dim() == 2:
        coords = coords.unsqueeze(0)
    coords = coords.view([coords.shape[0], 1] + list(coords.shape[1:]))
    B, C = grid.shape[:2]
    n = coords.shape[-2]
    out = F.grid_sample(grid, coords, align_corners=True, mode='bilinear', padding_mode='border')
    return out.view(B, C, n).transpose(-1, -2).squeeze()


def hexplane_features(pts, timestamps, aabb, planes_by_level, order=None, aabb_host=None):
    """HexPlaneField.get_density (scene/hexplane.py:160-175) + interpolate_ms_features (:73-106).
    `order` and `aabb_host` are speed hints of the HIP backend and are ignored here."""
    pts = normalize_aabb(pts, aabb)
    if not torch.is_tensor(timestamps):
        timestamps = torch.full((pts.shape[0], 1), float(timestamps), dtype=pts.dtype, device=pts.device)
    p4 = torch.cat((pts, timestamps.reshape(-1, 1)), dim=-1)
    combs = list(itertools.combinations(range(4), 2))
    feats = []
    for planes in planes_by_level:
        acc = 1.0
        for ci, comb in enumerate(combs):
            C = planes[ci].shape[1]
            acc = acc * grid_sample_wrapper(planes[ci], p4[..., comb]).view(-1, C)
        feats.append(acc)
    return torch.cat(feats, dim=-1)


def l1_loss_with_sums(img, gt):
    # utils/loss_utils.py:23-24 ; utils/image_utils.py:14-15
    d = img - gt
    return torch.abs(d).mean(), torch.stack([torch.abs(d).sum().detach(), (d * d).sum().detach()])


def compute_plane_smoothness(t):
    # scene/regulation.py:22-28
    h = t.shape[-2]
    first = t[..., 1:, :] - t[..., :h - 1, :]
    second = first[..., 1:, :] - first[..., :h - 2, :]
    return torch.square(second).mean()


def plane_regulation(planes, w_smooth, w_l1):
    # scene/gaussian_model.py:730-769 flattened: sum_p w_smooth[p]*smooth(p) + w_l1[p]*mean|1-p|
    total = 0.0
    for p, ws, wl in zip(planes, w_smooth, w_l1):
        if ws != 0:
            total = total + ws * compute_plane_smoothness(p)
        if wl != 0:
            total = total + wl * torch.abs(1 - p).mean()
    return total


def adam_reference(params, grads, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """One torch.optim.Adam (single-tensor path) update on plain tensors; returns nothing, updates in place."""
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        m.lerp_(g, 1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def ssim(img1, img2, window_size=11):
    # utils/loss_utils.py:52-92
    from math import exp
    channel = img1.size(-3)
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    window = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous().to(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    s2 = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    s12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()


def densify_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
    """train_4DGS.py:266 + scene/gaussian_model.py:713-715 (boolean-mask indexing), in place."""
    vis = radii > 0
    max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis].to(max_radii2D.dtype))
    xyz_gradient_accum[vis] += torch.norm(viewspace_grad[vis, :2], dim=-1, keepdim=True).reshape(xyz_gradient_accum[vis].shape)
    denom[vis] += 1


def select_rows(mask, tensors):
    """scene/gaussian_model.py:409-482,511-581: boolean-mask indexing, one tensor at a time."""
    return [t[mask] for t in tensors]


class TorchBackend:
    """Oracle backend for host-logic tests on machines without a GPU (installed explicitly by tests)."""
    name = "torch-oracle"
    hexplane_features = staticmethod(hexplane_features)
    l1_loss_with_sums = staticmethod(l1_loss_with_sums)
    ssim = staticmethod(ssim)
    densify_stats = staticmethod(densify_stats)
    select_rows = staticmethod(select_rows)
    plane_regulation = staticmethod(plane_regulation)
    Adam = torch.optim.Adam


def deform_mlp(feat, xyz, scaling, rotation, scene_flow, flow_coef, params):
    """Deformation.forward_dynamic's MLP part (scene/deformation.py:97-135, shipped config) as plain torch ops."""
    W0, b0, *rest = params
    hidden = F.linear(feat, W0, b0)
    outs = []
    for k in range(3):
        W1, b1, W2, b2 = rest[4 * k:4 * k + 4]
        outs.append(F.linear(torch.relu(F.linear(torch.relu(hidden), W1, b1)), W2, b2))
    return xyz + (outs[0] + flow_coef * scene_flow), scaling + outs[1], rotation + outs[2]


TorchBackend.deform_mlp = staticmethod(deform_mlp)
