"""Where does a 20-step window lose 5 % against the steady state?  bench.py's own sequence of calls (sizing, the two passes over the
cameras, 50 + 200 steady steps, 5 warm-up steps), then the 20-step window with an event behind every step: per-step GPU time, the host's
enqueue time per step, and the window's edges.     python tools/probe/window20.py"""
import importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
cfg = bench.CONFIGS["c2"]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, gc_freeze=True)
cams = trainer.cams
for c in cams:
    c.device_tensors(torch.device("cuda"))
def one(i):
    return trainer.step(5001 + (i % 90), cams=[cams[i % len(cams)]])
one(0)
for keep in (True, False):
    trainer.fused.keep_all_tiles = keep
    trainer.fused.exact_next()
    for i in range(1, 1 + len(cams)):
        one(i); torch.cuda.synchronize()
trainer.fused.exact_next()
nxt = 1 + len(cams)
for i in range(250):
    one(nxt + i)
nxt += 250
for i in range(5):
    one(nxt + i)
nxt += 5
MODES = {"drain": lambda: trainer.drain(),
         "device_sync_then_drain": lambda: (torch.cuda.synchronize(), trainer.drain()),
         "stream_sync_then_drain": lambda: (torch.cuda.current_stream().synchronize(), trainer.drain())}
for mode in ("drain", "device_sync_then_drain", "stream_sync_then_drain", "drain", "device_sync_then_drain", "stream_sync_then_drain"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        loss = one(nxt + i)
    t1 = time.perf_counter()
    MODES[mode]()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nxt += 20
    print(json.dumps({"events": None, "end": mode, "steps_per_s": round(20 / dt, 1), "window_ms": round(dt * 1e3, 2), "enqueue_ms": round((t1 - t0) * 1e3, 2),
                      "end_ms": round((t2 - t1) * 1e3, 2)}), flush=True)
    for i in range(5):
        one(nxt + i)
    nxt += 5
for with_events in (False, True, False):
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    host = []
    t0 = time.perf_counter()
    if with_events:
        evs[0].record()
    for i in range(20):
        h0 = time.perf_counter()
        loss = one(nxt + i)
        if with_events:
            evs[i + 1].record()
        host.append((time.perf_counter() - h0) * 1e3)
    t1 = time.perf_counter()
    trainer.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nxt += 20
    out = {"events": with_events, "steps_per_s": round(20 / dt, 1), "window_ms": round(dt * 1e3, 2), "enqueue_ms": round((t1 - t0) * 1e3, 2),
           "host_ms_per_step": [round(h, 2) for h in host]}
    if with_events:
        out["gpu_ms_per_step"] = [round(evs[i].elapsed_time(evs[i + 1]), 3) for i in range(20)]
    print(json.dumps(out), flush=True)
    for i in range(5):
        one(nxt + i)
    nxt += 5
