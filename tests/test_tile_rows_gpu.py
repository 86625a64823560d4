"""Tile-row shard at the library level (BASELINE config 4): two virtual ranks on one GPU, each binning /
compositing / back-propagating only its tile rows (MomRasterArgs.tile_row0/1), exchanging the per-Gaussian record
`gacc` between the two halves of the backward (mom_raster_backward_render / _geometry), must reproduce the unsharded
run: sorted per-tile lists and image rows bit for bit, gradients to float-atomic rounding."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

import scenes
from hip_helpers import N, RC, _aligned, t

pytestmark = pytest.mark.gpu
SENTINEL = -123.0


def _run(s, rows=None, dcol=None, ddep=None):
    """Forward (+ render half of the backward) through the C ABI with an optional tile-row range."""
    lib = N.lib()
    P, H, W = s["means3D"].shape[0], s["H"], s["W"]
    a, keep = RC._args(t(s["bg"]), t(s["means3D"]), t(None), t(s["opacities"]), t(s["scales"]), t(s["rotations"]), 1.0, t(None),
                       t(s["viewmatrix"]), t(s["projmatrix"]), s["tanfovx"], s["tanfovy"], H, W, t(s["shs"]), 3, t(s["campos"]),
                       False, False)
    if rows is not None:
        a.tile_row0, a.tile_row1 = rows
    dev = "cuda"
    color = torch.full((3, H, W), SENTINEL, device=dev)
    depth = torch.full((1, H, W), SENTINEL, device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    geom = torch.empty(lib.mom_raster_geom_bytes(P), dtype=torch.uint8, device=dev)
    img = torch.empty(lib.mom_raster_image_bytes(W, H), dtype=torch.uint8, device=dev)
    nr_dev = torch.zeros(2, dtype=torch.int32, device=dev)
    nr_host = torch.empty(1, dtype=torch.int32).pin_memory()
    st = N.current_stream()
    N.check(lib.mom_raster_forward_geometry(C.byref(a), geom.data_ptr(), img.data_ptr(), radii.data_ptr(), nr_dev.data_ptr(),
                                            nr_host.data_ptr(), st), "geometry")
    torch.cuda.synchronize()
    R = int(nr_host[0])
    binning = torch.empty(lib.mom_raster_binning_bytes(P, W, H, R), dtype=torch.uint8, device=dev)
    N.check(lib.mom_raster_forward_render(C.byref(a), geom.data_ptr(), binning.data_ptr(), R, img.data_ptr(), color.data_ptr(),
                                          depth.data_ptr(), nr_dev[1:].data_ptr(), st), "render")
    torch.cuda.synchronize()
    lay = N.MomRasterLayout()
    lib.mom_raster_layout(P, W, H, R, C.byref(lay))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    im = _aligned(img).cpu().numpy()
    b = _aligned(binning).cpu().numpy()
    out = dict(a=a, keep=keep, R=R, color=color, depth=depth, radii=radii, geom=geom, binning=binning, img=img, lay=lay,
               ranges=im[lay.img_ranges:lay.img_ranges + tiles * 8].view(np.uint32).reshape(tiles, 2).copy(),
               point_list=b[lay.bin_point_list:lay.bin_point_list + R * 4].view(np.uint32).copy())
    if dcol is not None:
        N.check(lib.mom_raster_backward_render(C.byref(a), geom.data_ptr(), binning.data_ptr(), R, img.data_ptr(), dcol.data_ptr(),
                                               ddep.data_ptr(), st), "backward_render")
        torch.cuda.synchronize()
    return out


def _gacc(run, P):
    off = run["lay"].geom_gacc
    return _aligned(run["geom"])[off:off + P * 4 * N.GACC_FLOATS].view(torch.float32).view(P, N.GACC_FLOATS)


def _geometry_backward(run, P, act_rotations_raw=None, copies=False):
    lib = N.lib()
    z = lambda *sh: torch.zeros(*sh, device="cuda")
    bufs = dict(dL_dmeans2D=z(P, 3), dL_dcolors=z(P, 3), dL_dopacity=z(P, 1), dL_dmeans3D=z(P, 3), dL_dcov3D=z(P, 6),
                dL_dsh=z(P, 16, 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4))
    gr = N.MomRasterGrads()
    for k, v in bufs.items():
        setattr(gr, k, v.data_ptr())
    if act_rotations_raw is not None:
        gr.act_rotations_raw = act_rotations_raw.data_ptr()
    if copies:
        bufs["dL_dscales_copy"], bufs["dL_drotations_copy"] = z(P, 3) + 7, z(P, 4) + 7
        gr.dL_dscales_copy, gr.dL_drotations_copy = bufs["dL_dscales_copy"].data_ptr(), bufs["dL_drotations_copy"].data_ptr()
    N.check(lib.mom_raster_backward_geometry(C.byref(run["a"]), run["radii"].data_ptr(), run["geom"].data_ptr(), C.byref(gr),
                                             N.current_stream()), "backward_geometry")
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in bufs.items()}


@pytest.mark.parametrize("W,H,split", [(203, 117, 3), (320, 240, 7), (128, 96, 1)])
def test_two_row_shards_reproduce_the_unsharded_pass(W, H, split):
    P = 12000
    s = scenes.random_gaussians(P, seed=W + H, W=W, H=H, scale=(-3.8, -1.8))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    assert 0 < split < gy
    rng = torch.Generator().manual_seed(4)
    dcol = torch.randn(3, H, W, generator=rng).cuda()
    ddep = torch.randn(1, H, W, generator=rng).cuda()
    full = _run(s, None, dcol, ddep)
    parts = [_run(s, (0, split), dcol, ddep), _run(s, (split, gy), dcol, ddep)]

    # --- binning: every instance lands on exactly one rank, per-tile sorted lists are the unsharded ones
    assert parts[0]["R"] + parts[1]["R"] == full["R"] and min(parts[0]["R"], parts[1]["R"]) > 0
    for r, (r0, r1) in enumerate(((0, split), (split, gy))):
        rg = parts[r]["ranges"]
        for tile in range(gx * gy):
            lo, hi = full["ranges"][tile]
            mine = parts[r]["point_list"][rg[tile][0]:rg[tile][1]]
            if r0 <= tile // gx < r1:
                np.testing.assert_array_equal(mine, full["point_list"][lo:hi])
            else:
                assert mine.size == 0
        np.testing.assert_array_equal(parts[r]["radii"].cpu().numpy(), full["radii"].cpu().numpy())   # projection is global

    # --- images: local rows bit-identical, other rows untouched
    y = split * 16
    for k in ("color", "depth"):
        f, p0, p1 = full[k].cpu().numpy(), parts[0][k].cpu().numpy(), parts[1][k].cpu().numpy()
        np.testing.assert_array_equal(p0[:, :y], f[:, :y])
        np.testing.assert_array_equal(p1[:, y:], f[:, y:])
        assert (p0[:, y:] == SENTINEL).all() and (p1[:, :y] == SENTINEL).all()

    # --- backward: the per-Gaussian records of the ranks sum to the unsharded record ...
    g_full = _gacc(full, P).clone()
    g_sum = _gacc(parts[0], P) + _gacc(parts[1], P)
    scale = float(g_full.abs().max())
    assert float((g_sum - g_full).abs().max()) <= 2e-5 * scale, (float((g_sum - g_full).abs().max()), scale)
    assert float(_gacc(parts[0], P)[:, 10:].abs().max()) == 0.0
    # ... and after the exchange (here: written back into rank 0's scratch) the geometry half gives the same gradients
    want = _geometry_backward(full, P)
    _gacc(parts[0], P).copy_(g_sum)
    got = _geometry_backward(parts[0], P)
    for k in want:
        sc = max(1e-20, float(np.abs(want[k]).max()))
        assert np.abs(got[k] - want[k]).max() <= 3e-5 * sc, (k, float(np.abs(got[k] - want[k]).max()), sc)


def test_activation_backward_inside_the_projection_backward_is_the_separate_launch_bit_for_bit():
    """MomRasterGrads.act_rotations_raw: dL_dscales / dL_drotations / dL_dopacity leave the projection backward already through
    exp / normalize / sigmoid (gaussian_renderer/__init__.py:134-137) -- the same numbers mom_activations_backward makes of the plain
    gradients in its own launch, bit for bit, a clamped quaternion (|q| < eps) included."""
    lib = N.lib()
    P, W, H = 9000, 176, 112
    s = dict(scenes.random_gaussians(P, seed=11, W=W, H=H, scale=(-3.8, -1.8)))
    g = torch.Generator().manual_seed(3)
    dev = lambda x: torch.as_tensor(x).float().cuda().contiguous()
    raw_s = torch.log(dev(s["scales"]))
    raw_r = dev(s["rotations"]) * (0.25 + 3 * torch.rand(P, 1, generator=g)).cuda()
    raw_r[5] = 0
    raw_r[6] *= 1e-14
    raw_o = torch.logit(dev(s["opacities"]).clamp(1e-4, 1 - 1e-4))
    sc, rot, op = torch.empty_like(raw_s), torch.empty_like(raw_r), torch.empty_like(raw_o)
    st = N.current_stream()
    N.check(lib.mom_activations_forward(P, raw_s.data_ptr(), raw_r.data_ptr(), raw_o.data_ptr(), sc.data_ptr(), rot.data_ptr(),
                                        op.data_ptr(), st), "act_fwd")
    torch.cuda.synchronize()
    s["scales"], s["rotations"], s["opacities"] = (x.cpu().numpy() for x in (sc, rot, op))
    dcol = torch.randn(3, H, W, generator=g).cuda()
    ddep = torch.randn(1, H, W, generator=g).cuda()
    run = _run(s, None, dcol, ddep)
    plain = _geometry_backward(run, P)
    fused = _geometry_backward(run, P, act_rotations_raw=raw_r, copies=True)
    # (dL_dscales_copy / dL_drotations_copy: the same values a second time, for a caller that reduces the first pair in place)
    np.testing.assert_array_equal(fused.pop("dL_dscales_copy"), fused["dL_dscales"])
    np.testing.assert_array_equal(fused.pop("dL_drotations_copy"), fused["dL_drotations"])
    want = [torch.empty(P, k, device="cuda") for k in (3, 4, 1)]
    ds, dr, do = (torch.from_numpy(plain[k]).cuda() for k in ("dL_dscales", "dL_drotations", "dL_dopacity"))
    N.check(lib.mom_activations_backward(P, sc.data_ptr(), raw_r.data_ptr(), op.data_ptr(), ds.data_ptr(), dr.data_ptr(),
                                         do.data_ptr(), want[0].data_ptr(), want[1].data_ptr(), want[2].data_ptr(), st), "act_bwd")
    torch.cuda.synchronize()
    assert np.abs(plain["dL_dscales"]).max() > 0 and np.abs(plain["dL_drotations"]).max() > 0
    for k, w in zip(("dL_dscales", "dL_drotations", "dL_dopacity"), want):
        np.testing.assert_array_equal(fused[k], w.cpu().numpy(), err_msg=k)
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh"):        # everything else is untouched
        np.testing.assert_array_equal(fused[k], plain[k], err_msg=k)
    # the option needs the inputs it differentiates through
    a = run["a"]
    gr = N.MomRasterGrads()
    gr.act_rotations_raw = raw_r.data_ptr()
    keep_scales = a.scales
    a.scales = None
    rc = lib.mom_raster_backward_geometry(C.byref(a), run["radii"].data_ptr(), run["geom"].data_ptr(), C.byref(gr), st)
    a.scales = keep_scales
    assert rc == N.MOM_EINVAL


def test_row_range_edge_cases():
    s = scenes.random_gaussians(3000, seed=2, W=96, H=80)
    gy = 5
    empty = _run(s, (2, 2))            # a rank may own no rows (more ranks than rows)
    assert empty["R"] == 0 and (empty["color"] == SENTINEL).all()
    whole = _run(s, (0, gy))
    base = _run(s, None)
    assert whole["R"] == base["R"]
    np.testing.assert_array_equal(whole["color"].cpu().numpy(), base["color"].cpu().numpy())
    np.testing.assert_array_equal(whole["point_list"], base["point_list"])
    a = base["a"]
    a.tile_row0, a.tile_row1 = 3, 2     # inverted / out of range ranges are refused before anything is launched
    lib = N.lib()
    rc = lib.mom_raster_forward_geometry(C.byref(a), base["geom"].data_ptr(), base["img"].data_ptr(), base["radii"].data_ptr(),
                                         torch.empty(2, dtype=torch.int32, device="cuda").data_ptr(),
                                         torch.empty(1, dtype=torch.int32).pin_memory().data_ptr(), N.current_stream())
    assert rc == -1
    a.tile_row0, a.tile_row1 = 0, gy + 1
    rc = lib.mom_raster_forward_geometry(C.byref(a), base["geom"].data_ptr(), base["img"].data_ptr(), base["radii"].data_ptr(),
                                         torch.empty(2, dtype=torch.int32, device="cuda").data_ptr(),
                                         torch.empty(1, dtype=torch.int32).pin_memory().data_ptr(), N.current_stream())
    assert rc == -1
