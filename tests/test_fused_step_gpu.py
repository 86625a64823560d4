"""The autograd-free launch sequence (fused_step.py) must reproduce render() + loss.backward() + Adam."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(fused, steps=3, lambda_dssim=0.0):
    import bench
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=fused, lambda_dssim=lambda_dssim)
    assert (trainer.fused is not None) == fused
    losses = []
    for it in range(steps):
        cam = trainer.cams[(3 * it + 1) % len(trainer.cams)]
        losses.append(float(trainer.step(5001 + it, cams=[cam])))
    torch.cuda.synchronize()
    dn = g._deformation.deformation_net
    out = {"xyz": g._xyz, "f_dc": g._features_dc, "f_rest": g._features_rest, "scaling": g._scaling, "rotation": g._rotation,
           "opacity": g._opacity, "plane_xy": dn.grid.grids[1][0], "plane_zt": dn.grid.grids[0][5],
           "w0": dn.feature_out[0].weight, "b_rot": dn.rotations_deform[3].bias, "w_pos1": dn.pos_deform[1].weight,
           "accum": g.xyz_gradient_accum, "denom": g.denom, "maxr": g.max_radii2D}
    return losses, {k: v.detach().float().cpu().numpy().copy() for k, v in out.items()}


@pytest.mark.parametrize("lambda_dssim", [0.0, 0.2])
def test_fused_step_matches_autograd_path(lambda_dssim):
    """lambda_dssim 0 is the reference's default loss, 0.2 adds the SSIM term (train_4DGS.py:222-223)."""
    la, pa = _run(False, lambda_dssim=lambda_dssim)
    lf, pf = _run(True, lambda_dssim=lambda_dssim)
    np.testing.assert_allclose(lf, la, rtol=2e-5)
    for k in pa:
        a, b = pf[k], pa[k]
        scale = max(1e-12, float(np.abs(b).max()))
        # Adam normalises every update to ~lr, so a gradient that differs by rounding moves a parameter by at most ~lr
        assert np.abs(a - b).max() <= 2e-4 * scale + 1e-6, (k, float(np.abs(a - b).max()), scale)
    np.testing.assert_array_equal(pf["denom"], pa["denom"])
    np.testing.assert_array_equal(pf["maxr"], pa["maxr"])
