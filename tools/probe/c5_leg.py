"""bench.py's c5 leg alone (4 M Gaussians, 1080p, the prune round at iteration 5100 inside the window), `n` times in one process:
    python tools/probe/c5_leg.py [n]
prints value and the three segments of every repeat -- to tell a box effect from a code effect (repeat 1 pays first-use costs)."""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
dev = torch.device("cuda", 0)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    o = bench.side_leg(bench.CONFIGS["c5"], dev, "fused", 121, 10, with_densify=True)
    seg = o["densify_in_window"]["segments"]
    print(json.dumps({"repeat": rep, "value": round(o["value"], 1), "before": round(seg["before"]["steps_per_s"], 1), "boundary_ms": round(seg["boundary"]["ms"], 1),
                      "after": round(seg["after"]["steps_per_s"], 1), "gaussians_after": seg["after"]["gaussians"], "host_enqueue_ms": round(o["host_enqueue_ms_per_step"], 2)}))
