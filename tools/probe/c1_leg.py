"""bench.py's c1 and c2 fused legs, `n` times each in one process (host-paced at c1: 5 k Gaussians, 256 x 256):
    python tools/probe/c1_leg.py [n]"""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
dev = torch.device("cuda", 0)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for c, steps in (("c1", 400), ("c2", 200)):
        o = bench.side_leg(bench.CONFIGS[c], dev, "fused", steps, 50)
        print(json.dumps({"repeat": rep, "config": c, "steps_per_s": round(o["value"], 1), "host_enqueue_ms": round(o["host_enqueue_ms_per_step"], 3)}), flush=True)
