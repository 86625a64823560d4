"""bench.py's two API-path legs alone (the reference's loop shape through render() + loss.backward() + optimizer.step()), plus
the fused headline path for the same box:  python tools/api_legs.py [repeats]"""
import importlib.util, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c2"]
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for name, kw in (("fused", dict(path="fused")), ("via_render_api", dict(path="autograd")), ("via_render_api_exact", dict(path="autograd", sync_mode="exact"))):
        o = bench.side_leg(cfg, dev, kw.pop("path"), 200, 30, **kw)
        print(name, "%.1f steps/s, host enqueue %.3f ms/step" % (o["value"], o["host_enqueue_ms_per_step"]), flush=True)
