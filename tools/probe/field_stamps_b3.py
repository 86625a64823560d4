"""In-kernel cycle stamps of the bf16x3 fused deformation forward (diagnostic build lib/var/stamps.so, -DMOM_FIELD_STAMPS)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = importlib.import_module("iclr2025_3d-mom_amd._native")
scene, g, trainer, op = bench.build_state(bench.CONFIGS["c2"], torch.device("cuda"), fused=True, lambda_dssim=0.0)
for i in range(30):
    trainer.step(5001 + i, cams=[trainer.cams[i % len(trainer.cams)]])
torch.cuda.synchronize()
buf = np.zeros((256, 32, 4), np.uint64)
assert N.lib().mom_debug_field_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
b = buf.astype(np.float64)
w, e = b[:, :12], b[:, 16:28]
tiles = w[..., 3].sum()
print("waves: tiles %.2f (max %d)  total %.0f cyc (max %.0f)  prologue %.0f" % (w[..., 3].mean(), w[..., 3].max(), w[..., 2].mean(), w[..., 2].max(), e[..., 2].mean()))
print("per tile: gather %.0f  bounce+split %.0f  3 head layers (bias, MFMA, relu) %.0f  out layers+stores %.0f" %
      (w[..., 0].sum() / tiles, w[..., 1].sum() / tiles, e[..., 0].sum() / tiles, e[..., 1].sum() / tiles))
