"""How fast can the host enqueue training steps?  Times the Python/ctypes enqueue loop of bench.py's workload WITHOUT the
final synchronize (host time per step) and with it (wall time per step).  If the two are close, the step is bound by
the host's launch rate, not by the GPU.

Use a SHORT run (--steps 10..50) to read the host's own cost: once more work is queued than the launch queue holds,
further launches block until the GPU catches up and "host enqueue" just tracks the GPU step (long runs still tell
host-bound from GPU-bound correctly through the work left in flight at the end).

    python tools/host_rate.py [--config c2] [--steps 200]
"""
import argparse
import importlib.util
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    import torch
    cfg = bench.CONFIGS[a.config]
    dev = torch.device("cuda", 0)
    scene, g, trainer, op = bench.build_state(cfg, dev, fused=True)
    cams = trainer.cams
    for c in cams:
        c.device_tensors(torch.device("cuda", 0))      # as bench.py: inputs resident before timing

    def one(i):
        return trainer.step(5001 + (i % 90), cams=[cams[i % len(cams)]])

    for i in range(20):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host, wall = (t1 - t0) / a.steps * 1e3, (t2 - t0) / a.steps * 1e3
    print(f"host enqueue {host:.3f} ms/step   wall {wall:.3f} ms/step   GPU still busy after the last enqueue: {(t2 - t1) * 1e3:.2f} ms")
    print("host-bound" if (t2 - t1) * 1e3 < 2 * wall else "GPU-bound (the host runs ahead)")


if __name__ == "__main__":
    main()
