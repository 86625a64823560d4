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
windows = os.environ.get("LONG_WINDOWS") == "1"        # wall time of every 100-iteration window, split at 10 steps after the boundary
tw, marks = time.perf_counter(), []
for i in range(20, 20 + n):
    l = trainer.step(it0 + i)
    if i % 100 == 0:
        losses.append((it0 + i, float(l), g.get_xyz.shape[0], trainer.replayed))
    if windows and (it0 + i) % 100 in (10, 95):
        trainer.drain(); torch.cuda.synchronize()
        now = time.perf_counter()
        marks.append(((it0 + i) % 100, now - tw))
        tw = now
if windows:
    plain = [t for k, t in marks[1:] if k == 95]      # 85 iterations without a boundary
    bound = [t for k, t in marks[1:] if k == 10]      # 15 iterations around a boundary
    print("boundary windows (ms):", " ".join("%.1f" % (1e3 * t) for t in bound))
    print("plain 85 iterations: %.1f ms (%.0f steps/s); 15 around a boundary: %.1f ms" % (
        1e3 * sum(plain) / len(plain), 85 * len(plain) / sum(plain), 1e3 * sum(bound) / len(bound)))
trainer.drain(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
for row in losses:
    print("iter %d  loss %.5f  points %d  replayed %d" % row)
print(f"{n} iterations in {dt:.2f} s = {n / dt:.1f} steps/s; points {g.get_xyz.shape[0]}; replayed {trainer.replayed}; finite {bool(torch.isfinite(g._xyz).all())}")
