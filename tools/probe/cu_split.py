"""Steps/s of one config for the MOM_B3F_BLOCKS the environment names (the MLP backward's workgroup count: 256 - that many CUs stay free
for the appearance Adam launch on the second stream).  One process per setting (the library reads the variable once):
    for b in 224 208 192; do MOM_B3F_BLOCKS=$b python tools/probe/cu_split.py c5; done"""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
c = sys.argv[1] if len(sys.argv) > 1 else "c5"
o = bench.side_leg(bench.CONFIGS[c], torch.device("cuda", 0), "fused", 60 if c == "c5" else 120, 10)
print(json.dumps({"config": c, "blocks": os.environ.get("MOM_B3F_BLOCKS", "default (224)"), "steps_per_s": round(o["value"], 2)}))
