"""How many lanes are still compositing (render_fwd: `live`) / contribute (render_bwd: all per-lane tests passed) when a wave walks a
list entry -- csrc/raster_render.hip built with "-DFWD_STAMPS -DFWD_LANE_HIST" (tools/variants.sh; MOM4D_LIB names that build).  Per camera class: entries walked and
entries that pass the wave-level reject, by lane-count bucket.  Decides whether a "few live pixels" mode for the tail of a list pays."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
N = importlib.import_module("iclr2025_3d-mom_amd._native")
cfg = bench.CONFIGS["c2"]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
cams = trainer.cams
for i in range(20):
    trainer.step(5001 + i, cams=[cams[(17 * i) % len(cams)]])
trainer.drain(); torch.cuda.synchronize()
real = C.CDLL(N.LIB_PATH)
buf = np.zeros(28, dtype=np.uint64)
names = ["0", "1-2", "3-4", "5-8", "9-16", "17-32", "33-64"]
for label, idxs in (("hemisphere views (the last five cameras)", list(range(len(cams) - 5, len(cams)))), ("ordinary cameras", list(range(0, 60, 4)))):
    real.mom_debug_lane_hist(buf.ctypes.data_as(C.c_void_p), 1)
    for k, ci in enumerate(idxs):
        trainer.step(5100 + k + 1, cams=[cams[ci]])
    trainer.drain(); torch.cuda.synchronize()
    real.mom_debug_lane_hist(buf.ctypes.data_as(C.c_void_p), 1)
    h = buf.reshape(2, 7, 2).astype(np.float64) / len(idxs)
    for kern, kn in ((0, "render_fwd (live lanes)"), (1, "render_bwd (valid lanes)")):
        tot = h[kern, :, 0].sum()
        print(f"{label} | {kn}: {tot / 1e6:.2f} M (wave, entry) pairs walked per frame; share by lanes: " +
              ", ".join(f"{n}: {100 * h[kern, b, 0] / tot:.1f}% (pass {100 * h[kern, b, 1] / max(h[kern, b, 0], 1):.0f}%)" for b, n in enumerate(names)))
