"""bench.py's via_render_api legs (render() + torch loss + loss.backward() + optimizer.step() at config 2) with the second-stream
overlap of the API path switched on and off IN ONE PROCESS, alternating -- boxes differ by up to 2x on host-bound legs, so only a
same-process A/B says what ops.API_OVERLAP is worth:   python tools/probe/api_leg.py [rounds]"""
import importlib
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import torch  # noqa: E402

ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
dev = torch.device("cuda", 0)
rows = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for overlap in (True, False):
        ops.API_OVERLAP = overlap
        for mode in ("async", "exact"):
            o = bench.side_leg(bench.CONFIGS["c2"], dev, "autograd", 150, 30, sync_mode=mode)
            rows.append({"round": rnd, "overlap": overlap, "sync_mode": mode, "steps_per_s": round(o["value"], 1),
                         "host_enqueue_ms": round(o["host_enqueue_ms_per_step"], 3)})
            print(json.dumps(rows[-1]), flush=True)
