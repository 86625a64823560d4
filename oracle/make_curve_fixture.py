"""Writes tests/golden/g10_loss_curve.npz: SURVEY 8(c)'s end-to-end fixture -- 50 coarse + 50 fine iterations of BASELINE
config 1 (5 000 Gaussians, 8 frames, 256x256) through the package's own Trainer with every libmom4d call replaced by the CPU
oracle (oracle.cpu_backend: the C restatement of the rasterizer + the reference's torch-op sequence), fixed seeds, a fixed
camera cycle: per-iteration loss, Gaussian count, and checksums of the final parameters.  tests/test_whole_step_gpu.py replays
the same 100 iterations on the HIP path and compares.  TEST INFRASTRUCTURE (run in the build container or anywhere with
the oracle built):

    python oracle/make_curve_fixture.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = dict(P=5000, F=8, W=256, H=256, time_res=8, name="c1")
N_COARSE, N_FINE = 50, 50


def checksums(g):
    dn = g._deformation.deformation_net
    t = {"xyz": g._xyz, "f_dc": g._features_dc, "f_rest": g._features_rest, "scaling": g._scaling, "rotation": g._rotation,
         "opacity": g._opacity, "plane_0_0": dn.grid.grids[0][0], "plane_1_5": dn.grid.grids[1][5],
         "w0": dn.feature_out[0].weight, "w_pos1": dn.pos_deform[1].weight, "accum": g.xyz_gradient_accum, "denom": g.denom}
    out = {}
    for k, v in t.items():
        v = v.detach().double().cpu()
        out["sum_" + k] = float(v.sum())
        out["abs_" + k] = float(v.abs().sum())
    return out


def run(device, fused_fine, lambda_dssim=0.2):
    """The 100 iterations on `device`; returns (losses [100], points [100], checksums dict, sample of final xyz)."""
    import importlib
    A = importlib.import_module("iclr2025_3d-mom_amd.arguments")
    S = importlib.import_module("iclr2025_3d-mom_amd.scene")
    T = importlib.import_module("iclr2025_3d-mom_amd.train")
    args, lp, op, pp, hp = A.default_args(time_resolution=CFG["time_res"])
    op.lambda_dssim = lambda_dssim
    torch.manual_seed(6666)
    scene = S.SyntheticScene(CFG["P"], CFG["F"], CFG["W"], CFG["H"], seed=6666)
    g = S.GaussianModel(lp.sh_degree, hp, device=device)
    scene.init_gaussians(g)                     # create_from_pcd: the "init" state of SURVEY 8(d)
    losses, points = [], []
    coarse = T.Trainer(scene, g, op, hp, pp, stage="coarse", delta_scale=1, sync_every_step=False)
    for i in range(N_COARSE):
        cam = coarse.cams[(3 * i + 1) % len(coarse.cams)]
        losses.append(float(coarse.step(1 + i, cams=[cam])))
        points.append(g.get_xyz.shape[0])
    fine = T.Trainer(scene, g, op, hp, pp, stage="fine", delta_scale=1, sync_every_step=False, fused=fused_fine)
    for i in range(N_FINE):
        cam = fine.cams[(5 * i + 2) % len(fine.cams)]
        losses.append(float(fine.step(1 + i, cams=[cam])))
        points.append(g.get_xyz.shape[0])
    if hasattr(fine, "drain"):
        fine.drain()
    cs = checksums(g)
    return np.array(losses), np.array(points), cs, g._xyz.detach().cpu().numpy()[::97].copy()


def main():
    from oracle import cpu_backend
    from oracle import raster_oracle as ro
    ro.build()
    ro.set_threads(1)                           # deterministic accumulation order
    torch.set_num_threads(1)
    with cpu_backend.installed():
        losses, points, cs, xyz = run("cpu", fused_fine=False)
    out = os.path.join(ROOT, "tests", "golden", "g10_loss_curve.npz")
    np.savez_compressed(out, losses=losses, points=points, xyz_sample=xyz, **{k: np.float64(v) for k, v in cs.items()})
    print("wrote", out, "loss", losses[0], "->", losses[N_COARSE - 1], "|", losses[N_COARSE], "->", losses[-1])


if __name__ == "__main__":
    main()
