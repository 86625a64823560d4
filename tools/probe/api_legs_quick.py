"""bench.py's two render()-path legs (async / exact) and the fused headline, 150 steps each: one line.  For same-call A/B of host-side changes."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
dev = torch.device("cuda", 0)
out = {}
for mode in ("async", "exact"):
    o = bench.side_leg(bench.CONFIGS["c2"], dev, "autograd", 150, 30, sync_mode=mode)
    out[mode] = (round(o["value"], 1), round(o["host_enqueue_ms_per_step"], 3))
o = bench.side_leg(bench.CONFIGS["c1"], dev, "fused", 300, 50)
out["c1"] = round(o["value"], 1)
print(sys.argv[1] if len(sys.argv) > 1 else "", json.dumps(out))
