"""Does the length of bench.py's timed window change the figure?  Config 2, fused step, the headline's protocol (synchronize, K steps, drain,
loss, synchronize) for K = 20 / 50 / 100 / 200, three rounds in one process.    python tools/probe/window_len.py"""
import importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
cfg = bench.CONFIGS["c2"]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, gc_freeze=True)
cams = trainer.cams
for c in cams:
    c.device_tensors(torch.device("cuda"))
n = 0
def one():
    global n
    loss = trainer.step(5001 + n % 90, cams=[cams[n % len(cams)]]); n += 1
    return loss
for _ in range(300):
    one()
for rnd in range(3):
    for K in (20, 50, 100, 200, 20):
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            loss = one()
        t1 = time.perf_counter()
        trainer.drain()
        t2 = time.perf_counter()
        loss = loss.tensor()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"round": rnd, "K": K, "steps_per_s": round(K / dt, 1), "enqueue_ms": round((t1 - t0) * 1e3, 2), "drain_ms": round((t2 - t1) * 1e3, 2),
                          "tail_ms": round((dt - (t2 - t0)) * 1e3, 3), "first_step_n": n - K}), flush=True)
