"""In-kernel cycle stamps of the fused deformation forward (diagnostic build lib/var/stamps.so, -DMOM_FIELD_STAMPS):
where the MFMA waves and the gather waves spend their time.  MOM4D_LIB=.../stamps.so python tools/probe/field_stamps.py"""
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
cfg = bench.CONFIGS["c2"]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
cams = trainer.cams
for i in range(30):
    trainer.step(5001 + i, cams=[cams[i % len(cams)]])
torch.cuda.synchronize()
buf = np.zeros((256, 32, 4), np.uint64)
lib = N.lib()
assert lib.mom_debug_field_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
b = buf.astype(np.float64)
NM = int(os.environ.get("NM", "2"))
mf, ga = b[:, :4 * NM], b[:, 4 * NM:16]
ga = ga[ga[..., 3] > 0].reshape(-1, 4)
print("MFMA waves:   tiles %.2f  total %.0f cyc  waiting for tiles %.0f  until first tile %.0f  -> per tile busy %.0f"
      % (mf[..., 3].mean(), mf[..., 2].mean(), mf[..., 0].mean(), mf[..., 1].mean(),
         ((mf[..., 2] - mf[..., 0]).sum() / mf[..., 3].sum())))
print("gather waves: tiles %.2f  total %.0f cyc  waiting for buffer %.0f  gathering %.0f  -> per tile %.0f"
      % (ga[:, 3].mean(), ga[:, 2].mean(), ga[:, 0].mean(), ga[:, 1].mean(), ga[:, 1].sum() / ga[:, 3].sum()))
print("max MFMA-wave total %.0f cyc, max gather total %.0f" % (mf[..., 2].max(), ga[:, 2].max()))
ex = b[:, 16:16 + 4 * NM]
tiles = mf[..., 3].sum()
print("per tile: trunk layer %.0f cyc, head hidden layers %.0f (3 layers), out layers + relu %.0f (3)" % (ex[..., 0].sum() / tiles, ex[..., 1].sum() / tiles, ex[..., 2].sum() / tiles))
