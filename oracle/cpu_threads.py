"""Thread scaling of the CPU baseline (the oracle timed by bench.py's cpu_baseline leg): prints the step time of the same
fine-stage step at 8..128 threads on this host.  It chose the 16 threads bench.py uses (8: 2.4 s, 16: 1.4 s, 32: 1.6 s,
64: 2.5 s, 128: 4.5 s per step on the GPU box's 2 x EPYC 9575F).  Part of the oracle, like everything that runs it.

    python oracle/cpu_threads.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
