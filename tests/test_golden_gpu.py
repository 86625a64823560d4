"""The committed golden vectors (tests/golden/g1..g8: inputs and outputs of the REAL reference modules, written by
oracle/ref_harness.py) fed straight through the HIP path on the GPU -- through the package's MODULES (HexPlaneField,
deform_network, utils.loss_utils, GaussianModel), not just the raw ops, so that a regression in how they are composed
(aabb flip, time axis, frame_num * scene_flow, dead heads, optimizer surgery) is caught on the GPU box, where
/root/reference does not exist.  CPU counterparts of the same assertions: tests/test_golden_cpu.py.

Tolerances: the reference is fp32 on the CPU (ATen); the HIP kernels are fp32 with a different (but fixed-order) summation
inside a Gaussian and float atomics across Gaussians.  Values: 2e-6 relative to the tensor's scale; gradients that are sums
over hundreds of points: 2e-5 of the tensor's scale.
"""
import argparse
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pkg = "iclr2025_3d-mom_amd"
DEV = "cuda"


def load(name):
    return np.load(os.path.join(G, name))


class HP:
    net_width = 64; timebase_pe = 4; defor_depth = 0; posebase_pe = 10; scale_rotation_pe = 2; opacity_pe = 2
    timenet_width = 64; timenet_output = 32; bounds = 1.6; plane_tv_weight = 0.0001; time_smoothness_weight = 0.01
    l1_time_planes = 0.0001
    kplanes_config = {'grid_dimensions': 2, 'input_coordinate_dim': 4, 'output_coordinate_dim': 32, 'resolution': [8, 8, 8, 5]}
    multires = [1, 2]; no_dx = False; no_grid = False; no_ds = False; no_dr = False; no_do = True; no_dshs = True
    empty_voxel = False; grid_pe = 0; static_mlp = False; apply_rotation = False


def close(got, want, rel, what=""):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape and err <= rel * scale, (what, err, scale)


def _field_with_g1_planes(d):
    HexPlaneField = importlib.import_module(pkg + ".scene.hexplane").HexPlaneField
    f = HexPlaneField(1.6, HP.kplanes_config, HP.multires).to(DEV)
    f.set_aabb([1.0, 1.2, 1.4], [-1.0, -1.2, -1.4])
    with torch.no_grad():
        for l in range(2):
            for i in range(6):
                f.grids[l][i].copy_(torch.tensor(d[f"plane_{l}_{i}"]))
    return f


def test_g1_hexplane_field_module_on_hip():
    """G1: 257 points incl. out-of-box and exact-corner ones, t in {0, 0.3, 1}: features, d/d points, d/d every plane."""
    d = load("g1_hexplane.npz")
    f = _field_with_g1_planes(d)
    np.testing.assert_array_equal(f.aabb.cpu().numpy(), d["aabb"])
    w = torch.tensor(d["w"], device=DEV)
    for t in (0.0, 0.3, 1.0):
        p = torch.tensor(d["pts"], device=DEV).requires_grad_(True)
        f.zero_grad(set_to_none=True)
        feat = f(p, t)
        (feat * w).sum().backward()
        close(feat, d[f"feat_t{t}"], 2e-6, f"feat t={t}")
        close(p.grad, d[f"dpts_t{t}"], 2e-5, f"dpts t={t}")
        for l in range(2):
            for i in range(6):
                close(f.grids[l][i].grad, d[f"dplane_{l}_{i}_t{t}"], 2e-5, f"dplane {l} {i} t={t}")
        # per-point timestamps (the reference passes a [P,1] tensor) must give the same features as the scalar
        feat2 = f(p.detach(), torch.full((p.shape[0], 1), t, device=DEV))
        close(feat2, d[f"feat_t{t}"], 2e-6, f"feat (tensor t) t={t}")


def test_g2_deform_network_module_on_hip():
    """G2: the reference's seeded deform_network (state_dict from the fixture) on its inputs, frame_num in {0, 7},
    delta_scale in {0, 1}: five outputs, gradients w.r.t. inputs and every live parameter, dead heads without gradient."""
    d = load("g2_deform.npz")
    deform_network = importlib.import_module(pkg + ".scene.deformation").deform_network
    net = deform_network(HP).to(DEV)
    net.deformation_net.set_aabb([1.0, 1.2, 1.4], [-1.0, -1.2, -1.4])
    sd = {k[4:]: torch.tensor(d[k]) for k in d.files if k.startswith("sd__")}
    assert sorted(sd) == sorted(net.state_dict().keys())
    net.load_state_dict(sd)
    assert net.deformation_net._fusable()                      # the fused HIP MLP is what runs
    ws = [torch.tensor(d[f"w{i}"], device=DEV) for i in range(3)]
    op, sh, flow = (torch.tensor(d[k], device=DEV) for k in ("opacity", "shs", "scene_flow"))
    for frame_num, delta_scale, t in ((0, 0, 0.0), (7, 1, 0.4)):
        tag = f"f{frame_num}_d{delta_scale}"
        x = torch.tensor(d["xyz"], device=DEV).requires_grad_(True)
        s = torch.tensor(d["scaling"], device=DEV).requires_grad_(True)
        r = torch.tensor(d["rotation"], device=DEV).requires_grad_(True)
        net.zero_grad(set_to_none=True)
        pts, sc, ro_, op_o, sh_o = net(x, s, r, op, sh, t, flow, frame_num, delta_scale)
        ((pts * ws[0]).sum() + (sc * ws[1]).sum() + (ro_ * ws[2]).sum()).backward()
        close(pts, d[f"pts_{tag}"], 2e-6, "pts " + tag)
        close(sc, d[f"scales_{tag}"], 2e-6, "scales " + tag)
        close(ro_, d[f"rots_{tag}"], 2e-6, "rots " + tag)
        assert torch.equal(op_o, op) and torch.equal(sh_o, sh)                    # pass-through (no_do, no_dshs)
        close(x.grad, d[f"dxyz_{tag}"], 2e-5, "dxyz " + tag)
        close(s.grad, d[f"dscal_{tag}"], 2e-6, "dscal " + tag)
        close(r.grad, d[f"drot_{tag}"], 2e-6, "drot " + tag)
        for k, p in net.named_parameters():
            ref = d[f"grad_{tag}__{k}"]
            if ref.size == 0:
                assert p.grad is None, k
            else:
                close(p.grad, ref, 3e-5, f"grad {tag} {k}")


def test_g3_losses_on_hip():
    """G3: 3x37x53 images (odd sizes: the zero-padded borders of the SSIM window): l1, ssim, psnr, d/d image of
    L1 + 0.2 (1 - SSIM)."""
    d = load("g3_loss.npz")
    L = importlib.import_module(pkg + ".utils.loss_utils")
    I = importlib.import_module(pkg + ".utils.image_utils")
    img, gt = torch.tensor(d["img"], device=DEV), torch.tensor(d["gt"], device=DEV)
    x = img.clone().requires_grad_(True)
    l1 = L.l1_loss(x, gt)
    ss = L.ssim(x, gt)
    (l1 + 0.2 * (1.0 - ss)).backward()
    np.testing.assert_allclose(float(l1), float(d["l1"]), rtol=2e-6)
    np.testing.assert_allclose(float(ss), float(d["ssim"]), rtol=2e-6)
    close(x.grad, d["dimg"], 2e-5, "dimg")
    np.testing.assert_allclose(float(L.psnr_from_last_l1()), float(d["psnr"].reshape(-1)[0]), rtol=2e-6)
    np.testing.assert_allclose(I.psnr(img, gt).cpu().numpy(), d["psnr"], rtol=2e-6)


def test_g7_compute_regulation_on_hip():
    """G7: GaussianModel.compute_regulation(0.01, 1e-4, 1e-4) on G1's planes: value and every plane's gradient."""
    d, g1 = load("g7_regulation.npz"), load("g1_hexplane.npz")
    GaussianModel = importlib.import_module(pkg + ".scene.gaussian_model").GaussianModel
    gm = GaussianModel(3, HP, device=DEV)
    gm._deformation = gm._deformation.to(DEV)
    grids = gm._deformation.deformation_net.grid.grids
    with torch.no_grad():
        for l in range(2):
            for i in range(6):
                grids[l][i].copy_(torch.tensor(g1[f"plane_{l}_{i}"]))
    v = gm.compute_regulation(0.01, 1e-4, 1e-4)
    v.backward()
    np.testing.assert_allclose(float(v), float(d["value"]), rtol=2e-6)
    for l in range(2):
        for i in range(6):
            close(grids[l][i].grad, d[f"dplane_{l}_{i}"], 2e-5, f"dplane {l} {i}")


def test_g8_adam_densify_prune_reset_on_hip():
    """G8: 500 seeded Gaussians: one Adam step with fixed gradients (one-launch FusedAdam), densification statistics,
    densify (clone + split; the HIP row selection), prune, opacity reset, with the Adam-moment surgery."""
    d = load("g8_densify.npz")
    GaussianModel = importlib.import_module(pkg + ".scene.gaussian_model").GaussianModel
    ops = importlib.import_module(pkg + ".ops")
    gm = GaussianModel(3, HP, device=DEV)
    gm._deformation = gm._deformation.to(DEV)
    P = torch.nn.Parameter
    names = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
    for k in names:
        setattr(gm, k, P(torch.tensor(d[k], device=DEV)))
    gm._scene_flow = torch.tensor(d["_scene_flow"], device=DEV)
    n = gm._xyz.shape[0]
    gm._deformation_table = torch.ones(n, dtype=torch.bool, device=DEV)
    gm.max_radii2D = torch.zeros(n, device=DEV)
    gm.spatial_lr_scale = 0.29
    opt = argparse.Namespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                             position_lr_delay_mult=0.01, position_lr_max_steps=20000, deformation_lr_init=1.6e-4,
                             deformation_lr_final=1.6e-6, deformation_lr_delay_mult=0.01, grid_lr_init=1.6e-3,
                             grid_lr_final=1.6e-5, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
    gm.training_setup(opt)
    assert isinstance(gm.optimizer, ops.FusedAdam)
    assert [g["name"] for g in gm.optimizer.param_groups] == ["xyz", "deformation", "grid", "f_dc", "f_rest", "opacity",
                                                                "scaling", "rotation"]
    for k in names:
        getattr(gm, k).grad = torch.tensor(d["grad" + k], device=DEV)
    gm.optimizer.step()
    for k in names:
        # eps = 1e-15: the first step moves every element by +-lr (m / sqrt(v) = +-1); division rounding only
        close(getattr(gm, k), d["after_step" + k], 1e-6, "after_step" + k)
    # statistics through the one-kernel path the trainer uses (radii > 0 <=> the reference's visibility filter)
    radii = torch.tensor(d["vis"], device=DEV).to(torch.int32) * 3
    gm.update_densification_stats(radii, torch.tensor(d["vsp"], device=DEV))
    close(gm.xyz_gradient_accum, d["accum"], 1e-6, "accum")
    np.testing.assert_array_equal(gm.denom.cpu().numpy(), d["denom"])
    gm.max_radii2D.zero_()
    torch.manual_seed(33)
    gm.densify(2e-4, 0.005, 5.0, None, 5, 5)
    assert gm._xyz.shape[0] == int(d["dens_P"])
    n_split = 0
    for k in ("_features_dc", "_scaling", "_rotation", "_opacity", "_scene_flow"):
        close(getattr(gm, k), d["dens" + k], 2e-6, "dens" + k)
    # positions: clones are bit-equal; the split's samples come from torch.normal on another device's generator, so their
    # positions are compared in distribution: (x - parent) rotated back and divided by the scale must be ~N(0,1)
    ref_xyz = d["dens_xyz"]
    got_xyz = gm._xyz.detach().cpu().numpy()
    same = np.isclose(got_xyz, ref_xyz, rtol=1e-6, atol=1e-7).all(axis=1)
    n_split = int((~same).sum())
    # the split's 2k children are the last rows (k parents were removed in front of them); everything before is bit-equal
    assert n_split > 0 and n_split % 2 == 0 and (~same)[-n_split:].all() and same[:-n_split].all(), n_split
    k = n_split // 2
    par_scale = np.exp(d["dens_scaling"][-n_split:]) * 1.6        # children carry log(s / 1.6): undo it -> the parents' scales
    assert np.allclose(par_scale[:k], par_scale[k:])              # repeat(N=2): child i and child k+i share a parent
    # both children of a parent are drawn around the parent's position: their offsets are bounded by a few sigma of it
    off = np.abs(got_xyz[-n_split:][:k] - got_xyz[-n_split:][k:])
    assert float((off / (par_scale[:k].max(axis=1, keepdims=True) + 1e-12)).max()) < 12.0
    st = gm.optimizer.state[gm._xyz]
    np.testing.assert_array_equal(st["exp_avg"].cpu().numpy() != 0, d["dens_exp_avg_xyz"] != 0)       # zero-extended moments
    close(st["exp_avg"], d["dens_exp_avg_xyz"], 1e-6, "exp_avg")
    close(st["exp_avg_sq"], d["dens_exp_avg_sq_xyz"], 1e-6, "exp_avg_sq")
    assert float(gm.xyz_gradient_accum.abs().sum()) == 0 and float(gm.max_radii2D.abs().sum()) == 0
    # prune / reset on the reference's own post-densify state (the random positions do not enter the prune mask)
    with torch.no_grad():
        gm._xyz.copy_(torch.tensor(ref_xyz, device=DEV))
    gm.max_radii2D = torch.tensor(d["maxr"], device=DEV)
    gm.prune(2e-4, 0.005, 5.0, 20)
    assert gm._xyz.shape[0] == int(d["prune_P"])
    close(gm._xyz, d["prune_xyz"], 1e-6, "prune_xyz")
    gm.reset_opacity()
    close(gm._opacity, d["reset_opacity"], 2e-6, "reset_opacity")
    st = gm.optimizer.state[gm._opacity]
    assert float(st["exp_avg"].abs().sum() + st["exp_avg_sq"].abs().sum()) == 0.0


def test_activations_kernels_against_the_reference_torch_ops():
    """a5: exp / F.normalize / sigmoid (gaussian_renderer/__init__.py:130-132) and their backward as the two fused launches."""
    import ctypes as C
    N = importlib.import_module(pkg + "._native")
    lib = N.lib()
    g = torch.Generator().manual_seed(5)
    P = 4097
    sr = (torch.randn(P, 3, generator=g) * 2 - 3)
    rr = torch.randn(P, 4, generator=g)
    rr[7] = 0                                          # |q| < eps: F.normalize clamps the norm
    rr[8] *= 1e-14
    orr = torch.randn(P, 1, generator=g) * 4
    ds, dr, do = torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g), torch.randn(P, 1, generator=g)
    a, b, c = (t.clone().requires_grad_(True) for t in (sr, rr, orr))
    s_ref, r_ref, o_ref = torch.exp(a), torch.nn.functional.normalize(b), torch.sigmoid(c)
    ((s_ref * ds).sum() + (r_ref * dr).sum() + (o_ref * do).sum()).backward()
    dev = lambda t: t.to(DEV).contiguous()
    srd, rrd, ord_, dsd, drd, dod = map(dev, (sr, rr, orr, ds, dr, do))
    s, r, o = torch.empty_like(srd), torch.empty_like(rrd), torch.empty_like(ord_)
    st = N.current_stream()
    N.check(lib.mom_activations_forward(P, srd.data_ptr(), rrd.data_ptr(), ord_.data_ptr(), s.data_ptr(), r.data_ptr(),
                                        o.data_ptr(), st), "act_fwd")
    gs, gr, go = torch.empty_like(srd), torch.empty_like(rrd), torch.empty_like(ord_)
    N.check(lib.mom_activations_backward(P, s.data_ptr(), rrd.data_ptr(), o.data_ptr(), dsd.data_ptr(), drd.data_ptr(),
                                         dod.data_ptr(), gs.data_ptr(), gr.data_ptr(), go.data_ptr(), st), "act_bwd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(s.cpu().numpy(), s_ref.detach().numpy(), rtol=3e-7, atol=0)
    np.testing.assert_allclose(r.cpu().numpy(), r_ref.detach().numpy(), rtol=3e-7, atol=1e-37)
    np.testing.assert_allclose(o.cpu().numpy(), o_ref.detach().numpy(), rtol=3e-7, atol=1e-30)
    np.testing.assert_allclose(gs.cpu().numpy(), a.grad.numpy(), rtol=1e-6, atol=1e-30)
    # rows 7 and 8 sit below F.normalize's eps: the clamped norm is a constant there, so the gradient is g / eps (ATen)
    np.testing.assert_allclose(gr.cpu().numpy(), b.grad.numpy(), rtol=2e-5, atol=2e-6)
    assert torch.isfinite(gr).all()
    np.testing.assert_allclose(go.cpu().numpy(), c.grad.numpy(), rtol=2e-6, atol=1e-30)
