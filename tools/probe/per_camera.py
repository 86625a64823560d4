"""GPU time of one fused training step per camera of the config-2 scene (60 video frames + the 5 hemisphere views), an event behind every step,
two cycles:   python tools/probe/per_camera.py"""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
scene, g, trainer, op = bench.build_state(bench.CONFIGS["c2"], torch.device("cuda"), fused=True, gc_freeze=True)
cams = trainer.cams
for c in cams:
    c.device_tensors(torch.device("cuda"))
n = len(cams)
for i in range(2 * n):
    trainer.step(5001 + i % 90, cams=[cams[i % n]])
torch.cuda.synchronize()
for cyc in range(2):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        trainer.step(5001 + i % 90, cams=[cams[i]])
        evs[i + 1].record()
    trainer.drain(); torch.cuda.synchronize()
    ms = [round(evs[i].elapsed_time(evs[i + 1]), 3) for i in range(n)]
    print(json.dumps({"cycle": cyc, "mean_ms": round(sum(ms) / n, 4), "video_frames_mean": round(sum(ms[:60]) / 60, 4), "last_five": ms[60:], "first_five": ms[:5],
                      "max": max(ms), "argmax": ms.index(max(ms)), "replayed": trainer.replayed}), flush=True)
