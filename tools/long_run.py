"""A longer training run of the benchmark scene through the Trainer's own cadence (densify / prune / opacity reset boundaries
included): steps/s over the whole window, Gaussian count, loss, overflow replays.  Stability check, not a bench line."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

n = int(os.environ.get("LONG_STEPS", "1200"))
cfg = bench.CONFIGS[os.environ.get("KBENCH_CONFIG", "c2")]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=float(os.environ.get("LONG_LAMBDA", "0.0")))
op.pruning_interval = int(os.environ.get("LONG_PRUNE", "100"))
print("densify_until", op.densify_until_iter, "interval", op.densification_interval, "prune", op.pruning_interval, "reset", op.opacity_reset_interval)
it0 = 2901                      # crosses an opacity reset (3000), an SH bump (3000) and a dozen densify / prune rounds
for i in range(20):
    trainer.step(it0 + i)
trainer.drain(); torch.cuda.synchronize()
t0 = time.perf_counter()
losses = []
for i in range(20, 20 + n):
    l = trainer.step(it0 + i)
    if i % 100 == 0:
        losses.append((it0 + i, float(l), g.get_xyz.shape[0], trainer.replayed))
trainer.drain(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
for row in losses:
    print("iter %d  loss %.5f  points %d  replayed %d" % row)
print(f"{n} iterations in {dt:.2f} s = {n / dt:.1f} steps/s; points {g.get_xyz.shape[0]}; replayed {trainer.replayed}; finite {bool(torch.isfinite(g._xyz).all())}")
