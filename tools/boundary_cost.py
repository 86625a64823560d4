"""Where does a densify / prune boundary iteration's time go?  Per-iteration wall time (synchronised) around boundaries."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import cProfile, pstats

cfg = bench.CONFIGS[os.environ.get("KBENCH_CONFIG", "c2")]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
op.pruning_interval = 100
it0 = 3050
for i in range(30):
    trainer.step(it0 + i)
trainer.drain(); torch.cuda.synchronize()
rows = []
worst = (0.0, None)
for i in range(30, 30 + 540):
    it = it0 + i
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prof = None
    if it % 100 == 0 and it > 3100:
        prof = cProfile.Profile(); prof.enable()
    trainer.step(it)
    torch.cuda.synchronize()
    if prof is not None:
        prof.disable()
    dt = (time.perf_counter() - t0) * 1e3
    if prof is not None and dt > worst[0]:
        worst = (dt, prof)
    if it % 100 in (0, 1) :
        rows.append((it, dt, g.get_xyz.shape[0]))
for r in rows:
    print("iter %d  %.2f ms  points %d" % r)
print("slowest boundary: %.1f ms" % worst[0])
pstats.Stats(worst[1]).sort_stats("tottime").print_stats(14)
