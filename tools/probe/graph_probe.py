"""Feasibility probe: capture one fused forward_backward (no optimizer step) in a torch.cuda.CUDAGraph (hipGraph) and replay it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

cfg = bench.CONFIGS[os.environ.get("KBENCH_CONFIG", "c2")]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
fs, cams = trainer.fused, trainer.cams
for i in range(10):
    trainer.step(5001 + i, cams=[cams[i % len(cams)]])
trainer.drain()
torch.cuda.synchronize()
cam = cams[3]
fs.exact_next()
fs.forward_backward(cam, 1)          # sizes buffers, builds descriptors
fs.forward_backward(cam, 1)
torch.cuda.synchronize()
ref = {"early": fs.early.clone(), "late": fs._dg_flat.clone(), "sums": fs.sums.clone()}
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    fs.forward_backward(cam, 1)
torch.cuda.synchronize()
fs.early.zero_(); fs._dg_flat.zero_()
graph.replay()
torch.cuda.synchronize()
for k in ("early", "late"):
    a, b = getattr(fs, "early" if k == "early" else "_dg_flat"), ref[k]
    print(k, float((a - b).abs().max()), float(b.abs().max()))
print("sums", fs.sums.tolist(), ref["sums"].tolist(), "flag", int(fs.flags[0]), "R", int(fs.nr_host[0]))
def timeit(f, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("eager  us/iter", timeit(lambda: fs.forward_backward(cam, 1)))
print("replay us/iter", timeit(graph.replay))
def host_only(f, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t = (time.perf_counter() - t0) / n * 1e6; torch.cuda.synchronize(); return t
print("eager host enqueue us", host_only(lambda: fs.forward_backward(cam, 1), 20), " replay host us", host_only(graph.replay, 20))
