"""What does one densify + prune round cost (scene/gaussian_model.py's optimizer surgery, every 100 iterations)?
Two rounds: the first one pays one-time library initialisation (measured: 400-575 ms), the second is the steady cost
(measured at 200k Gaussians: densify 4.7 ms, prune 2.5 ms, i.e. 0.07 ms per step amortised).

    python tools/time_densify.py [--config c2]
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
    a = ap.parse_args()
    import torch
    cfg = bench.CONFIGS[a.config]
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda", 0), fused=True)
    cams = trainer.cams
    for c in cams:
        c.device_tensors(torch.device("cuda", 0))
    for i in range(30):                       # accumulate densification statistics
        trainer.step(5001 + i, cams=[cams[i % len(cams)]])
    torch.cuda.synchronize()
    n0 = g.get_xyz.shape[0]
    t0 = time.perf_counter()
    g.densify(0.0002, 0.005, scene.cameras_extent, 20, 5, 5, scene.model_path, 5100, "fine")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n1 = g.get_xyz.shape[0]
    g.prune(0.0002, 0.005, scene.cameras_extent, 20)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n2 = g.get_xyz.shape[0]
    t3 = time.perf_counter()
    for i in range(3):                        # the steps right after: buffers and caches are rebuilt
        trainer.step(5101 + i, cams=[cams[i % len(cams)]])
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    # a second round: separates one-time costs (library initialisation on first use) from the steady cost
    for i in range(30):
        trainer.step(5110 + i, cams=[cams[i % len(cams)]])
    torch.cuda.synchronize()
    m0 = g.get_xyz.shape[0]
    u0 = time.perf_counter()
    g.densify(0.0002, 0.005, scene.cameras_extent, 20, 5, 5, scene.model_path, 5200, "fine")
    torch.cuda.synchronize()
    u1 = time.perf_counter()
    m1 = g.get_xyz.shape[0]
    g.prune(0.0002, 0.005, scene.cameras_extent, 20)
    torch.cuda.synchronize()
    u2 = time.perf_counter()
    print(f"second round: densify {1e3 * (u1 - u0):.1f} ms ({m0} -> {m1}), prune {1e3 * (u2 - u1):.1f} ms ({m1} -> {g.get_xyz.shape[0]})")
    print(f"first round (includes one-time library initialisation, e.g. the first torch.bmm): densify {1e3 * (t1 - t0):.1f} ms "
          f"({n0} -> {n1}), prune {1e3 * (t2 - t1):.1f} ms ({n1} -> {n2}); the 3 steps after it {1e3 * (t4 - t3):.1f} ms")
    print(f"steady state, amortised over 100 iterations: {(1e3 * (u2 - u0)) / 100:.3f} ms per step for the surgery itself")


if __name__ == "__main__":
    main()
