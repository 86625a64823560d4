"""How many (splat, tile) instances of the benchmark frame can reach alpha >= 1/255 somewhere in their 16x16 tile?

Reads the rasterizer's own buffers after one fused step (records, ranges, sorted point list) and evaluates the continuous
rectangle bound of csrc/raster_render.hip (strip_reach_mask) per instance in torch.  Measurement helper only."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = importlib.import_module("iclr2025_3d-mom_amd._native")


def main(cfg_name="c2"):
    cfg = bench.CONFIGS[cfg_name]
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
    fs = trainer.fused
    fs.exact_next()
    fs.forward_backward(trainer.cams[7], 1)
    torch.cuda.synchronize()
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    R = int(fs.nr_host[0])
    lay = N.MomRasterLayout()
    fs.lib.mom_raster_layout(P, W, H, fs.cap, C.byref(lay))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    geom = fs.geom[(-fs.geom.data_ptr()) % 256:]
    img = fs.img[(-fs.img.data_ptr()) % 256:]
    binb = fs.binning[(-fs.binning.data_ptr()) % 256:]
    rec = geom[lay.geom_rec:lay.geom_rec + P * 48].view(torch.float32).view(P, 3, 4)
    ranges = img[lay.img_ranges:lay.img_ranges + gx * gy * 8].view(torch.int32).view(gx * gy, 2).long()
    plist = binb[lay.bin_point_list:lay.bin_point_list + R * 4].view(torch.int32).long()
    counts = ranges[:, 1] - ranges[:, 0]
    assert int(counts.sum()) == R, (int(counts.sum()), R)
    tile = torch.repeat_interleave(torch.arange(gx * gy, device="cuda"), counts)
    # instance k of the sorted list belongs to tile[k] if ranges are contiguous in tile order
    order = torch.argsort(ranges[:, 0], stable=True)
    tile = torch.repeat_interleave(order, counts[order])
    r = rec[plist]
    cx, cy, a, b, c, opac = r[:, 0, 0], r[:, 0, 1], r[:, 1, 0], r[:, 1, 1], r[:, 1, 2], r[:, 1, 3]
    bound = -torch.log(255.0 * opac) - 2e-3

    def edge_max(fixed, lo, hi, qf, qt, bb):
        t = torch.minimum(torch.maximum(-bb * fixed / qt, lo), hi)
        return -0.5 * (qf * fixed * fixed + qt * t * t) - bb * fixed * t

    def reach(x0, y0, w, h):
        xa, xb, ya, yb = x0, x0 + (w - 1), y0, y0 + (h - 1)
        dxl, dxh, dyl, dyh = cx - xb, cx - xa, cy - yb, cy - ya
        inside = (cx >= xa) & (cx <= xb) & (cy >= ya) & (cy <= yb)
        best = torch.maximum(torch.maximum(edge_max(dxl, dyl, dyh, a, c, b), edge_max(dxh, dyl, dyh, a, c, b)),
                             torch.maximum(edge_max(dyl, dxl, dxh, c, a, b), edge_max(dyh, dxl, dxh, c, a, b)))
        best = torch.where(inside, torch.zeros_like(best), best)
        return ~(best < bound)

    tx0, ty0 = (tile % gx).float() * 16, (tile // gx).float() * 16
    whole = reach(tx0, ty0, 16, 16)
    strips = torch.stack([reach(tx0, ty0 + 4 * w, 16, 4) for w in range(4)], 1)
    print(f"R = {R}   P = {P}   tiles = {gx * gy}")
    print(f"instances that can reach their tile:          {int(whole.sum())}  ({float(whole.float().mean()):.3f})")
    print(f"mean strips reached per instance:             {float(strips.float().sum(1).mean()):.3f} of 4")
    print(f"instances reaching no strip:                  {float((~strips.any(1)).float().mean()):.3f}")
    per_g = torch.bincount(plist, minlength=P)
    per_g_reach = torch.bincount(plist[whole], minlength=P)
    vis = per_g > 0
    print(f"visible Gaussians {int(vis.sum())}; with no reachable tile at all: {int((vis & (per_g_reach == 0)).sum())}")
    print(f"tiles per visible Gaussian: {float(per_g[vis].float().mean()):.2f} -> {float(per_g_reach[vis].float().mean()):.2f}")
    print(f"opacity < 1/255 among visible: {int((rec[:, 1, 3][vis] < 1 / 255).sum())}")


if __name__ == "__main__":
    main(*sys.argv[1:])
