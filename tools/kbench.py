"""Per-kernel times of the fused training step at config c2 from the library's HIP-event slots (csrc/profile.hip), for the
library named by MOM4D_LIB (default: lib/libmom4d.so).  Used to A/B kernel variants in one gpurun call:
    for v in lib/var/*.so; do MOM4D_LIB=$v python tools/kbench.py hexplane_bwd mlp_bwd; done"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

prof = importlib.import_module("iclr2025_3d-mom_amd.profiling")
slots = [a for a in sys.argv[1:] if not a.startswith("--")] or ["hexplane_fwd", "hexplane_bwd", "mlp_fwd", "mlp_bwd", "render_fwd", "render_bwd"]
steps = 60
cfg = bench.CONFIGS[os.environ.get("KBENCH_CONFIG", "c2")]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
cams = trainer.cams
if os.environ.get("KBENCH_FREEZE"):   # the model never moves: variants whose gradients differ (experiments) still see one scene
    g.optimizer.step = lambda *a, **k: None
for i in range(70):
    trainer.step(5001 + i % 90, cams=[cams[i % len(cams)]])
torch.cuda.synchronize()
out = []
for s in slots:                       # one slot at a time: the event pairs of several slots would serialise the stream more
    prof.enable(s)
    for i in range(steps):
        trainer.step(5001 + i % 90, cams=[cams[i % len(cams)]])
    trainer.drain()
    torch.cuda.synchronize()
    ms, n = prof.read(s)
    prof.enable(s, False)
    out.append(f"{s} {ms / max(n, 1) * 1e3:.1f} us")
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(200):
    trainer.step(5001 + i % 90, cams=[cams[i % len(cams)]])
trainer.drain()
t1.record()
torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('MOM4D_LIB', 'default')):28s} step {t0.elapsed_time(t1) / 200 * 1e3:.0f} us | " + " | ".join(out), flush=True)
