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


class _VirtualRank:
    """Stands in for parallel.DistContext on one GPU: captures each buffer at start() and then poisons it, which is what
    a concurrent in-place all-reduce does to it from the point of view of any later kernel that still reads it."""
    def __init__(self, world):
        self.world, self.captured = world, []

    def start(self, tensor, op="sum"):
        assert tensor.is_contiguous()
        self.captured.append((op, tensor.clone()))
        tensor.fill_(float("nan") if tensor.is_floating_point() else -7)

    def finish(self):
        pass


def test_camera_batch_buckets_of_two_virtual_ranks_sum_to_the_batch_mean():
    """Multi-GPU exchange of the fused step, checked on one GPU: with world = 2 each rank's buckets carry 1/2, so the
    sum of what two ranks hand to start() must equal the mean of two plain single-GPU backward passes -- the
    reference's batch_size = 2 gradient (train_4DGS.py:189-229) -- and the radii their maximum.  Poisoning the buffers
    at start() proves no later kernel of the step reads a bucket that is being reduced."""
    import bench
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.2)
    fs, cams = trainer.fused, [trainer.cams[1], trainer.cams[4]]

    def snapshot():
        torch.cuda.synchronize()
        return {"radii": fs.radii.clone(), "g2d": fs.g2d.clone(), "early": fs.early.clone(), "late": fs._dg_flat.clone()}

    plain = []
    for cam in cams:                                   # no optimiser step in between: both see the same model
        fs.forward_backward(cam, 1)
        plain.append(snapshot())
    ranks = []
    for cam in cams:
        fs.dist = _VirtualRank(2)
        fs.forward_backward(cam, 1)
        torch.cuda.synchronize()
        ops_seen = [o for o, _ in fs.dist.captured]
        assert ops_seen == ["max", "sum", "sum", "sum"], ops_seen          # radii, g2d, early bucket, late bucket
        ranks.append(dict(zip(("radii", "g2d", "early", "late"), [t for _, t in fs.dist.captured])))
    fs.dist = None
    assert ranks[0]["early"].numel() == 56 * 6000 and ranks[0]["late"].numel() == fs._dg_flat.numel()
    torch.testing.assert_close(torch.maximum(ranks[0]["radii"], ranks[1]["radii"]),
                               torch.maximum(plain[0]["radii"], plain[1]["radii"]), rtol=0, atol=0)
    for k in ("g2d", "early", "late"):
        got = ranks[0][k] + ranks[1][k]
        want = 0.5 * (plain[0][k] + plain[1][k])
        assert torch.isfinite(got).all(), k
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * scale + 1e-9, (k, float((got - want).abs().max()), scale)
