"""Helpers for the GPU parity tests: run libmom4d through the reference-shaped `_C` functions and decode its
private scratch buffers (via mom_raster_layout) into numpy for comparison with the oracle."""
import ctypes as C
import importlib

import numpy as np
import torch

PKG = importlib.import_module("iclr2025_3d-mom_amd")
N = importlib.import_module("iclr2025_3d-mom_amd._native")
DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
RC = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization._C")


def t(a, dev="cuda"):
    if a is None:
        return torch.empty(0, device=dev)
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


def _aligned(buf):
    off = (-buf.data_ptr()) % 256
    return buf[off:]


def hip_forward(s, shs=True, colors_precomp=None, cov3D_precomp=None, sh_degree=3, scale_modifier=1.0, debug=False,
                keep_all_tiles=False):
    """keep_all_tiles: bin like the reference (whole rectangles) instead of the default cull; hip_backward of the returned
    state runs under the same setting."""
    dev = "cuda"
    P = s["means3D"].shape[0]
    args = (t(s["bg"]), t(s["means3D"]), t(colors_precomp), t(s["opacities"]),
            t(None if cov3D_precomp is not None else s["scales"]), t(None if cov3D_precomp is not None else s["rotations"]),
            scale_modifier, t(cov3D_precomp), t(s["viewmatrix"]), t(s["projmatrix"]), s["tanfovx"], s["tanfovy"], s["H"],
            s["W"], t(s["shs"] if (shs and colors_precomp is None) else None), sh_degree, t(s["campos"]), False, debug)
    RC.set_keep_all_tiles(keep_all_tiles)
    try:
        R, color, depth, radii, geom, binning, img = RC.rasterize_gaussians(*args)
        torch.cuda.synchronize()
    finally:
        RC.set_keep_all_tiles(False)
    lay = N.MomRasterLayout()
    N.lib().mom_raster_layout(P, s["W"], s["H"], R, C.byref(lay))
    W, H = s["W"], s["H"]
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(R=R, color=color.cpu().numpy(), depth=depth.cpu().numpy(), radii=radii.cpu().numpy(), args=args,
               bufs=(geom, binning, img), keep_all_tiles=keep_all_tiles)
    if P:
        g = _aligned(geom).cpu().numpy()
        rec = g[lay.geom_rec:lay.geom_rec + P * 48].view(np.float32).reshape(P, 12)
        out["means2D"] = rec[:, 0:2].copy()
        out["depths"] = rec[:, 2].copy()
        out["tiles_touched"] = rec[:, 3].copy().view(np.uint32)
        out["conic_opacity"] = rec[:, 4:8].copy()
        out["rgb"] = rec[:, 8:11].copy()
        out["cov3D"] = g[lay.geom_cov3D:lay.geom_cov3D + P * 24].view(np.float32).reshape(P, 6).copy()
        out["clamped"] = g[lay.geom_clamped:lay.geom_clamped + P * 4].reshape(P, 4)[:, :3].copy()
        im = _aligned(img).cpu().numpy()
        out["ranges"] = im[lay.img_ranges:lay.img_ranges + tiles * 8].view(np.uint32).reshape(tiles, 2).copy()
        out["n_contrib"] = im[lay.img_n_contrib:lay.img_n_contrib + W * H * 4].view(np.uint32).copy()
        out["final_T"] = im[lay.img_final_T:lay.img_final_T + W * H * 4].view(np.float32).copy()
        out["tile_counts"] = im[lay.img_tile_counts:lay.img_tile_counts + tiles * 4].view(np.uint32).copy()
        out["tile_walked"] = im[lay.img_tile_walked:lay.img_tile_walked + tiles * 4].view(np.uint32).copy()
        b = _aligned(binning).cpu().numpy()
        out["point_list"] = b[lay.bin_point_list:lay.bin_point_list + R * 4].view(np.uint32).copy()
    return out


def hip_backward(fw, dL_dcolor, dL_ddepth=None):
    a = fw["args"]
    (bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tanx, tany, H, W,
     sh, degree, campos, _, debug) = a
    geom, binning, img = fw["bufs"]
    dd = torch.zeros((1, H, W), device="cuda") if dL_ddepth is None else t(dL_ddepth)
    RC.set_keep_all_tiles(fw["keep_all_tiles"])
    try:
        res = RC.rasterize_gaussians_backward(bg, means3D, t(fw["radii"]), colors, scales, rotations, scale_modifier,
                                              cov3D_precomp, viewmatrix, projmatrix, tanx, tany, t(dL_dcolor), dd, sh, degree,
                                              campos, geom, fw["R"], binning, img, debug)
        torch.cuda.synchronize()
    finally:
        RC.set_keep_all_tiles(False)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    return {n: r.cpu().numpy() for n, r in zip(names, res)}
