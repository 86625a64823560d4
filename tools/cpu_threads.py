import sys, time, os
sys.path.insert(0, '.')
import torch
from oracle import cpu_backend, raster_oracle as ro
import bench
cfg = bench.CONFIGS["c2"]
print("cpu_count", os.cpu_count())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' ")
with cpu_backend.installed():
    torch.set_num_threads(32); ro.set_threads(32)
    scene, g, trainer, op = bench.build_state(cfg, "cpu")
    trainer.step(5001)
    for nt in (8, 16, 32, 64, 128):
        torch.set_num_threads(nt); ro.set_threads(nt)
        t=time.time(); trainer.step(5002); dt=time.time()-t
        print("threads", nt, "step s", round(dt,2), flush=True)
