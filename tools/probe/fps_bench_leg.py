import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
scene, g, trainer, op = bench.build_state(bench.CONFIGS["c2"], torch.device("cuda",0), fused=True, gc_freeze=True)
for i in range(20): trainer.step(5001+i, cams=[trainer.cams[i]])
trainer.drain()
for rep in range(2):
    r = bench.render_fps(scene, g, trainer.pipe, trainer.background, trainer.delta_scale)
    print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ("value","frames","frames_rendered_again")}) for k,v in r.items() if k not in ("mode","trajectory")}))
