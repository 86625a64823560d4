"""Host enqueue time against GPU time of the render() + loss.backward() path (one autograd node): 10 steps enqueued, then a
synchronisation; if the enqueue alone takes as long as the whole, the host sets the pace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
scene, g, trainer, op = bench.build_state(bench.CONFIGS["c2"], torch.device("cuda"), fused=False, gc_freeze=os.environ.get("GC") != "on")
if os.environ.get("PER_OP") == "1":
    trainer.pipe.per_op_autograd = True
DGR = __import__("importlib").import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
DGR.set_sync_mode("async", capacity_hint=2_000_000)
cams = trainer.cams
import gc
if os.environ.get("GC") == "freeze":
    gc.collect(); gc.freeze()
elif os.environ.get("GC") == "off":
    gc.disable()
for i in range(30):
    trainer.step(5001 + i, cams=[cams[i % len(cams)]])
torch.cuda.synchronize()
for n in (40, 40, 40, 40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        trainer.step(5040 + i, cams=[cams[i % len(cams)]])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{n:3d} steps: enqueue {1e3*(t1-t0)/n:.3f} ms/step, with the GPU {1e3*(t2-t0)/n:.3f} ms/step")
