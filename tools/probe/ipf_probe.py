import importlib.util, os, sys, time, gc
ROOT="/root/repo"
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch, importlib
ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
cfg = bench.CONFIGS["c2"]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda", 0), fused=False)
cams = trainer.cams
for c in cams: c.device_tensors(torch.device("cuda", 0))
orig = ops.in_place_flags
acc=[0.0,0]
def wrapped(held, planes):
    t=time.perf_counter(); r=orig(held, planes); acc[0]+=time.perf_counter()-t; acc[1]+=1; return r
ops.in_place_flags = wrapped
def one(i): return trainer.step(5001 + (i % 90), cams=[cams[i % len(cams)]])
for i in range(30): one(i)
torch.cuda.synchronize()
for mode in ("gc on", "gc off"):
    if mode=="gc off": gc.disable()
    acc[0]=0;acc[1]=0
    t=time.perf_counter()
    for i in range(300): one(i)
    th=time.perf_counter()-t
    torch.cuda.synchronize()
    tt=time.perf_counter()-t
    print(mode, "host %.3f ms/step, wall %.3f ms/step, in_place_flags %.1f us per call (%d calls)" % (th/300*1e3, tt/300*1e3, acc[0]/max(1,acc[1])*1e6, acc[1]))
