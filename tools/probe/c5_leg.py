import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
dev = torch.device("cuda", 0)
for wd in (True, False):
    o = bench.side_leg(bench.CONFIGS["c5"], dev, "fused", 120, 10, with_densify=wd)
    print("c5 with_densify" if wd else "c5 steady", round(o["value"], 1), o.get("densify_in_window", {}).get("gaussians_after"))
