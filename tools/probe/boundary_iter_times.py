"""Wall time of single iterations around a densify / prune boundary (synchronised after every iteration), early Adam on / off:
where do the extra milliseconds after a boundary go?   MOM_EARLY_ADAM=0|1 python tools/probe/boundary_iter_times.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

cfg = bench.CONFIGS["c2"]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True)
it0 = 3081
for i in range(10):
    trainer.step(it0 + i)
trainer.drain(); torch.cuda.synchronize()
rows = {}
for i in range(10, 250):
    it = it0 + i
    t = time.perf_counter()
    trainer.step(it)
    trainer.drain(); torch.cuda.synchronize()
    rows.setdefault(it % 100, []).append(1e3 * (time.perf_counter() - t))
for k in (98, 99, 0, 1, 2, 3, 4, 5, 50):
    v = rows.get(k, [])
    print("iteration %% 100 == %2d: %s ms" % (k, " ".join("%.2f" % x for x in v)))
