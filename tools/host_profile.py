"""cProfile of the host side of the fused training step (bench.py workload): where does the enqueue time go?

    python tools/host_profile.py [--config c2] [--steps 200] [--top 30]
"""
import argparse
import cProfile
import importlib.util
import os
import pstats
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--by-cum", action="store_true", help="sort by cumulative time")
    ap.add_argument("--async-mode", action="store_true", help="autograd path in the async sync mode (bench.py via_render_api)")
    ap.add_argument("--autograd", action="store_true", help="the render() + loss.backward() path instead of the fused step")
    ap.add_argument("--freeze", action="store_true", help="gc.collect(); gc.freeze() after setup, as bench.py's legs do")
    a = ap.parse_args()
    import torch
    if a.autograd:
        torch.autograd.set_multithreading_enabled(False)        # the backward's Python runs on this thread: the profiler sees it
    cfg = bench.CONFIGS[a.config]
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda", 0), fused=not a.autograd)
    cams = trainer.cams
    for c in cams:
        c.device_tensors(torch.device("cuda", 0))      # as bench.py: inputs resident before timing

    if a.autograd and a.async_mode:
        import importlib
        DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
        DGR.set_sync_mode("async", capacity_hint=2_000_000)
        import gc; gc.collect(); gc.freeze()

    if a.freeze:
        import gc; gc.collect(); gc.freeze()

    def one(i):
        return trainer.step(5001 + (i % 90), cams=[cams[i % len(cams)]])

    for i in range(20):
        one(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(a.steps):
        one(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.strip_dirs()
    total = sum(v[2] for v in st.stats.values())
    print(f"profiled {a.steps} steps, {total / a.steps * 1e3:.3f} ms/step of host time under the profiler")
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][3 if a.by_cum else 2])[:a.top]
    print(f"{'tottime ms/step':>16s} {'cumtime ms/step':>16s} {'calls/step':>11s}  function")
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print(f"{tt / a.steps * 1e3:16.4f} {ct / a.steps * 1e3:16.4f} {nc / a.steps:11.1f}  {fn}:{line}({name})")


if __name__ == "__main__":
    main()
