"""Whole training iterations on the HIP path against the same iterations on the CPU oracle path.

The oracle path is the package's own Trainer with every libmom4d call swapped for the oracle (oracle.cpu_backend: the C
restatement of the reference rasterizer + the reference's torch-op sequence for HexPlane / MLP / loss / Adam).  The HIP side
runs twice: the autograd path (render() + loss.backward()) and the fused launch sequence (fused_step.py).

What can differ: fp32 summation order (float atomics on the GPU, OpenMP-free serial sums in the oracle), v_exp_f32 vs
expf in the compositing exponent (~1.5 ulp), and therefore -- for a handful of (pixel, splat) pairs whose alpha sits
within an ulp of 1/255 or of the 0.99 cap -- a different branch.  Those pairs move single Gaussians' gradients visibly,
which is why the comparison is per ELEMENT with a stated quantile instead of a whole-tensor norm."""
import contextlib
import importlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")
LIVE = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "plane_0_0", "plane_0_2", "plane_1_3", "plane_1_5", "w0", "b0",
        "w_pos1", "w_pos3", "w_sc1", "w_rot3", "b_rot3")


def _tensors(g):
    dn = g._deformation.deformation_net
    return {"xyz": g._xyz, "f_dc": g._features_dc, "f_rest": g._features_rest, "scaling": g._scaling, "rotation": g._rotation,
            "opacity": g._opacity, "plane_0_0": dn.grid.grids[0][0], "plane_0_2": dn.grid.grids[0][2],
            "plane_1_3": dn.grid.grids[1][3], "plane_1_5": dn.grid.grids[1][5], "w0": dn.feature_out[0].weight,
            "b0": dn.feature_out[0].bias, "w_pos1": dn.pos_deform[1].weight, "w_pos3": dn.pos_deform[3].weight,
            "w_sc1": dn.scales_deform[1].weight, "w_rot3": dn.rotations_deform[3].weight, "b_rot3": dn.rotations_deform[3].bias}


def _one_step(device, fused, lambda_dssim, cam_index=1, cfg=None):
    """One fine-stage iteration from the seeded benchmark state; returns loss, every live gradient (Adam's first moment
    after ONE step is 0.1 * gradient exactly), the statistics and the radii."""
    import bench
    from oracle import cpu_backend
    ctx = cpu_backend.installed() if device == "cpu" else contextlib.nullcontext()
    with ctx:
        scene, g, trainer, op = bench.build_state(cfg or CFG, torch.device(device), fused=fused, lambda_dssim=lambda_dssim)
        assert (trainer.fused is not None) == fused
        loss = float(trainer.step(5001, cams=[trainer.cams[cam_index]]))
        if device != "cpu":
            trainer.drain()
            torch.cuda.synchronize()
        t = _tensors(g)
        grads = {k: (g.optimizer.state[t[k]]["exp_avg"].detach().float().cpu().numpy() * 10.0) for k in LIVE}
        stats = {"accum": g.xyz_gradient_accum.detach().cpu().numpy().copy(), "denom": g.denom.detach().cpu().numpy().copy(),
                 "maxr": g.max_radii2D.detach().cpu().numpy().copy()}
        dead = [n for n, p in g._deformation.named_parameters()
                if ("opacity_deform" in n or "shs_deform" in n or "timenet" in n) and p in g.optimizer.state
                and "exp_avg" in g.optimizer.state[p] and float(g.optimizer.state[p]["exp_avg"].abs().sum()) != 0]
    return loss, grads, stats, dead


def _cfg(name):
    import bench
    return CFG if name == "tiny" else bench.CONFIGS[name]


@pytest.mark.parametrize("cfg_name,lambda_dssim", [("tiny", 0.0), ("tiny", 0.2), ("c2", 0.0), ("c2", 0.2)])
def test_one_iteration_hip_vs_cpu_oracle(cfg_name, lambda_dssim):
    """"tiny": 6 k Gaussians, 160x96.  "c2": BASELINE configs[1] at full size -- bench.py's own scene and model state (200 k
    Gaussians, 960x540, 60 frames, HexPlane [64, 64, 64, 50]) -- one whole iteration (field, rasterizer, loss, backward,
    statistics, Adam) on the HIP paths against the CPU oracle Trainer running on 16 threads (VERDICT r3 missing 3)."""
    cfg = _cfg(cfg_name)
    if cfg_name != "tiny":
        from oracle import raster_oracle as ro
        ro.set_threads(16)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref_loss, ref_g, ref_s, _ = _one_step("cpu", False, lambda_dssim, cfg=cfg)
    for fused in (False, True):
        loss, grads, stats, dead = _one_step("cuda", fused, lambda_dssim, cfg=cfg)
        assert not dead, ("dead heads received a gradient", dead)
        assert abs(loss - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss)), (fused, loss, ref_loss)
        # statistics: visibility (radius > 0) is integer work and must agree exactly; the accumulated gradient norm is float
        np.testing.assert_array_equal(stats["denom"], ref_s["denom"])
        # radius = ceil(3 sqrt(lambda_max)) of a covariance built from the FIELD's outputs, which the HIP field forward and the
        # torch ops round differently (2e-7 relative): with 200 k Gaussians one or two sit within that of an integer and land on
        # the other side of the ceil.  (The rasterizer alone, on identical inputs, gives identical radii at this size:
        # tests/test_raster_gpu.py::test_forward_parity[...200000-960-540].)  Counted, and off by exactly one.
        dr = np.abs(stats["maxr"] - ref_s["maxr"])
        assert int((dr != 0).sum()) <= max(0 if cfg_name == "tiny" else 2, int(1e-5 * dr.size)) and float(dr.max(initial=0.0)) <= 1.0, \
            (int((dr != 0).sum()), float(dr.max()))
        for k in LIVE:
            a, b = grads[k], ref_g[k]
            assert a.shape == b.shape
            scale = max(float(np.abs(b).max()), 1e-30)
            err = np.abs(a - b) / scale
            # 99.9 % of the elements within 1e-4 of the tensor's scale, every element within 2e-3 -- at config-2 size (12 M
            # elements in f_rest) a counted handful, at most 1e-6 of the tensor and never fewer than 2 allowed, may reach 5e-3:
            # Gaussians with a (pixel, splat) pair within an ulp of the 1/255 / 0.99 / 1e-4 thresholds (measured: 2.6e-3)
            # A Gaussian whose gradient moved with a threshold pair hands the difference on to everything behind it: through the MLP
            # to the four texels x 32 channels it samples in each of six planes.  At config-2 size a dozen such Gaussians put ~0.1-0.2 %
            # of a 64 x 64 plane's elements beyond 1e-4 of the plane's scale (measured 1.1e-3 and 1.9e-3, none beyond 2.6e-3): the
            # planes get 3e-3 there, every other tensor keeps 1e-3.
            # The MLP's weight gradients are sums over ALL Gaussians (200 k terms of both signs per element at config 2): two fp32
            # summation orders of such a sum differ by ~1e-4 of the tensor's largest element by themselves (measured: every element
            # within 1.8e-4; the one-kernel backward against the two f32 kernels on identical inputs: 2e-5, tests/test_ops_gpu.py),
            # so "loose" means beyond 5e-4 for them there.
            weight = k.startswith(("w", "b")) and cfg_name != "tiny"
            frac_loose = float((err > (5e-4 if weight else 1e-4)).mean())
            n_far = int((err > 2e-3).sum())
            far_ok = 0 if cfg_name == "tiny" else max(2, int(1e-6 * err.size))
            loose_ok = 3e-3 if (cfg_name != "tiny" and k.startswith("plane_")) else 1e-3
            assert frac_loose <= loose_ok and n_far <= far_ok and float(err.max()) <= 5e-3, (fused, k, frac_loose, n_far, float(err.max()))
        acc_scale = max(float(np.abs(ref_s["accum"]).max()), 1e-30)
        e = np.abs(stats["accum"] - ref_s["accum"]) / acc_scale
        assert float((e > 1e-4).mean()) <= 1e-3 and float(e.max()) <= 2e-3, (fused, float(e.max()))


def test_loss_curve_of_config_1_replayed_on_hip():
    """tests/golden/g10_loss_curve.npz (oracle/make_curve_fixture.py): 50 coarse + 50 fine iterations of BASELINE config 1 on
    the CPU oracle path.  The same 100 iterations on the HIP path (coarse: autograd path; fine: fused step) must follow the
    same curve.  Adam with eps = 1e-15 turns a sign flip of a vanishing gradient into a full learning-rate step, so two
    correct fp32 implementations drift apart slowly: the loss is held to 2e-3 relative over all 100 iterations (1e-4 over the
    first 5 of each stage), the Gaussian count exactly, the parameter sums to 1e-3 of their absolute sums (rotations: 3e-3, see below)."""
    from oracle.make_curve_fixture import run, N_COARSE
    d = np.load(os.path.join(ROOT, "tests", "golden", "g10_loss_curve.npz"))
    losses, points, cs, xyz = run("cuda", fused_fine=True)
    np.testing.assert_array_equal(points, d["points"])
    ref = d["losses"]
    rel = np.abs(losses - ref) / np.abs(ref)
    assert float(rel[:5].max()) <= 1e-4 and float(rel[N_COARSE:N_COARSE + 5].max()) <= 1e-4, (rel[:5], rel[N_COARSE:N_COARSE + 5])
    assert float(rel.max()) <= 2e-3, (int(rel.argmax()), float(rel.max()))
    # Measured over 40 runs (tools/probe/curve_spread.py): every sum within 1.2e-4 of its absolute sum (10 x inside the bound) --
    # except the rotations', 9.5e-4 in the median and up to 1.04e-3: the gradient of a quaternion along its own direction vanishes
    # (the rasterizer normalises it), its sign is rounding noise, and Adam turns each sign into a full learning-rate step, so
    # that component random-walks for 100 iterations on either implementation.  It gets the bound that goes with that.
    for k in cs:
        if k.startswith("sum_"):
            tot = float(d["abs_" + k[4:]])
            bound = 3e-3 if k == "sum_rotation" else 1e-3
            assert abs(cs[k] - float(d[k])) <= bound * max(tot, 1e-12), (k, cs[k], float(d[k]), tot)
    np.testing.assert_allclose(xyz, d["xyz_sample"], rtol=0, atol=2e-3)
