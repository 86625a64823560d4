"""Measured HIP-vs-oracle errors of the rasterizer parity cases (tests/test_raster_gpu.py), to set the regression gates from
what the kernels achieve rather than from the north star's ceiling: python tools/parity_stats.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from hip_helpers import hip_backward, hip_forward  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402
from scenes import random_gaussians  # noqa: E402
import test_raster_gpu as T  # noqa: E402

ro.set_threads(16)
print("forward: case | mean L1 | max | pixels > 1e-5 | n_contrib mismatches | last-contributor mismatches | final_T mean")
for seed, P, W, H, kw in [(0, 2000, 128, 96, {}), (1, 5000, 256, 256, {}), (2, 700, 100, 50, dict(scale=(-3.0, -0.5))), (3, 64, 33, 17, {}),
                          (4, 20000, 320, 180, dict(scale=(-5.0, -3.0))), (5, 100000, 480, 270, dict(scale=(-5.5, -3.5)))]:
    s = random_gaussians(P, seed=seed, W=W, H=H, **kw)
    st = T._oracle(s)
    for keep in (True, False):
        fw = hip_forward(s, keep_all_tiles=keep)
        dc = np.abs(fw["color"] - st.out_color)
        dd = np.abs(fw["depth"] - st.out_depth)
        a = T._last_contributor(fw["n_contrib"], fw["ranges"], fw["point_list"], W, H)
        b = T._last_contributor(st.n_contrib, st.ranges, st.point_list, W, H)
        nc = int((fw["n_contrib"] != st.n_contrib).sum()) if keep else -1
        print(f"  P={P} {W}x{H} keep={int(keep)} | {dc.mean():.2e} | {dc.max():.2e} | {int((dc.max(0) > 1e-5).sum())} of {W * H} | {nc} | "
              f"{int((a != b).sum())} | {np.abs(fw['final_T'] - st.final_T).mean():.2e} | depth mean {dd.mean():.2e} max {dd.max():.2e}")
print("backward: case | tensor | rel err | rows > 2e-5 | rows > 1e-4 | worst row")
for seed, P, W, H, kw in [(10, 1500, 128, 96, {}), (11, 400, 70, 45, dict(scale=(-3.0, -1.0))), (12, 5000, 256, 256, dict(scale=(-5.0, -3.0)))]:
    s = random_gaussians(P, seed=seed, W=W, H=H, **kw)
    rng = np.random.default_rng(seed)
    dcol = rng.normal(size=(3, H, W)).astype(np.float32)
    ddep = (rng.normal(size=(1, H, W)) * 0.2).astype(np.float32)
    fw, st = hip_forward(s), T._oracle(s)
    g, go = hip_backward(fw, dcol, ddep), ro.backward(st, dcol, ddep)
    for name in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        a, b = g[name], go[name].reshape(g[name].shape)
        scale = max(float(np.abs(b).max()), 1e-30)
        row = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1) / scale
        print(f"  P={P} | {name:14s} | {T._relerr(a, b):.2e} | {int((row > 2e-5).sum())} | {int((row > 1e-4).sum())} | {row.max():.2e}")
