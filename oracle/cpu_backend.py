"""Swaps the oracle in for every libmom4d call so that the package's host logic (GaussianModel, render(), Trainer)
runs on a machine without a GPU -- TEST INFRASTRUCTURE ONLY.  Used by tests/ (host-logic and gloo tests) and by
bench.py's cpu_baseline leg, always through the explicit `installed()` context manager; the product path never
reaches this module."""
import contextlib
import importlib

import numpy as np
import torch

from . import raster_oracle as ro
from . import torch_ref as tr


def _np(t):
    return None if t is None or t.numel() == 0 else t.detach().cpu().numpy()


def rasterize_gaussians(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-117) on the CPU oracle; the oracle state rides in place of the
    three byte buffers."""
    st = ro.forward(_np(means3D), _np(opacity), _np(viewmatrix), _np(projmatrix), _np(campos), int(image_width),
                    int(image_height), float(tan_fovx), float(tan_fovy), _np(bg), shs=_np(sh), sh_degree=int(degree),
                    colors_precomp=_np(colors), scales=_np(scales), rotations=_np(rotations),
                    cov3D_precomp=_np(cov3D_precomp), scale_modifier=float(scale_modifier))
    holder = torch.zeros(1, dtype=torch.uint8)
    holder._oracle_state = st
    return (st.num_rendered, torch.from_numpy(st.out_color), torch.from_numpy(st.out_depth), torch.from_numpy(st.radii),
            holder, torch.zeros(1, dtype=torch.uint8), torch.zeros(1, dtype=torch.uint8))


def rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                 projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug):
    st = geomBuffer._oracle_state
    g = ro.backward(st, _np(dL_dout_color), None if dL_dout_depth is None else _np(dL_dout_depth))
    f = torch.from_numpy
    return (f(g["dL_dmeans2D"]), f(g["dL_dcolors"]), f(g["dL_dopacity"]), f(g["dL_dmeans3D"]), f(g["dL_dcov3D"]),
            f(g["dL_dsh"]), f(g["dL_dscales"]), f(g["dL_drotations"]))


def mark_visible(means3D, viewmatrix, projmatrix):
    return torch.from_numpy(ro.mark_visible(_np(means3D), _np(viewmatrix), _np(projmatrix)))


def distCUDA2(points):
    return torch.from_numpy(ro.knn_mean_dist2(_np(points)))


@contextlib.contextmanager
def installed():
    ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
    rc = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization._C")
    knn = importlib.import_module("iclr2025_3d-mom_amd.simple_knn._C")
    saved = (ops.BACKEND, rc.rasterize_gaussians, rc.rasterize_gaussians_backward, rc.mark_visible, knn.distCUDA2)
    ops.set_backend(tr.TorchBackend)
    rc.rasterize_gaussians, rc.rasterize_gaussians_backward, rc.mark_visible = (rasterize_gaussians,
                                                                                 rasterize_gaussians_backward, mark_visible)
    knn.distCUDA2 = distCUDA2
    try:
        yield
    finally:
        ops.set_backend(saved[0])
        rc.rasterize_gaussians, rc.rasterize_gaussians_backward, rc.mark_visible, knn.distCUDA2 = saved[1:]
