"""Would rendering consecutive frames on two alternating streams pay?  Frame k + 1's deformation field, projection and binning are
short latency-bound kernels; frame k's compositing is bound by vector issue.  Two FusedRender instances (their own scratch), frames
dealt alternately to two streams, against the one-stream loop: aggregate frames per second."""
import importlib, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
FR = importlib.import_module("iclr2025_3d-mom_amd.fused_render")
DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["c2"]
scene, g, trainer, op = bench.build_state(cfg, dev, fused=True)
cams = trainer.cams
for c in cams: c.device_tensors(dev)
bg = trainer.background
DGR.set_sync_mode("async", capacity_hint=2_000_000)
frs = [FR.FusedRender(g) for _ in range(3)]
sts = [torch.cuda.Stream() for _ in range(3)]
keep = []
def run(nstreams, frames):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(frames):
            cam = cams[i % len(cams)]
            if nstreams == 1:
                out = frs[0].render(cam, bg, trainer.delta_scale)
            else:
                k = i % nstreams
                with torch.cuda.stream(sts[k]):
                    out = frs[k].render(cam, bg, trainer.delta_scale)
            keep.append(out[0]); 
            if len(keep) > 8: keep.pop(0)
    torch.cuda.synchronize()
    return frames / (time.perf_counter() - t0)
for n in (1, 2, 3):
    run(n, 60)
for rep in range(2):
    for n in (1, 2, 3):
        print(f"{n} stream(s): {run(n, 600):.0f} frames/s", flush=True)
