"""What does the refresh of the processing orders (one Morton sort + six plane sorts every REORDER_EVERY steps) cost the step, and how
fast does an old order lose its value?  Config 2, fused step: steps/s over 512 steps for several cadences, in one process, alternating.
    python tools/probe/reorder_cost.py"""
import importlib, importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
H = importlib.import_module("iclr2025_3d-mom_amd.scene.hexplane")
cfg = bench.CONFIGS[os.environ.get("KBENCH_CONFIG", "c2")]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, gc_freeze=True)
cams = trainer.cams
for c in cams:
    c.device_tensors(torch.device("cuda"))
n = 0
def run(k):
    global n
    for _ in range(k):
        trainer.step(5001 + n % 90, cams=[cams[n % len(cams)]]); n += 1
run(100)
for rnd in range(2):
    for every in (64, 256, 1024, 10**9):
        H.HexPlaneField.REORDER_EVERY = every
        run(8)
        trainer.drain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(512)
        trainer.drain(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"round": rnd, "reorder_every": every, "steps_per_s": round(512 / dt, 1), "us_per_step": round(dt / 512 * 1e6, 1)}), flush=True)
