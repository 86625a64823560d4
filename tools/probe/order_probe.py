"""Does the binning suffer from the model's Gaussian order?  The synthetic scene's Gaussians are in random order, so a workgroup's
256 consecutive Gaussians touch ~950 distinct tiles and the per-workgroup LDS histograms aggregate nothing: one global atomic per
instance.  This probe permutes the model's rows into 3-D Morton order once (in place, before any step) and compares the binning
kernels' times and the step rate with the unpermuted model."""
import importlib, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
prof = importlib.import_module("iclr2025_3d-mom_amd.profiling")
ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
for permute in (False, True, False, True):
    scene, g, trainer, op = bench.build_state(cfg, dev, fused=True)
    if permute:
        with torch.no_grad():
            order = ops.morton_order(g._xyz.detach()).long()
            for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
                p = getattr(g, name)
                p.data.copy_(p.data[order].clone())
    cams = trainer.cams
    for c in cams: c.device_tensors(dev)
    for i in range(30): trainer.step(5001 + i, cams=[cams[i % len(cams)]])
    trainer.drain(); torch.cuda.synchronize()
    res = {}
    for k in ("tile_hist", "tile_scatter", "render_fwd"):
        prof.enable(k, True, period=3)
    t0 = time.perf_counter()
    n = 300
    for i in range(n): trainer.step(5031 + (i % 60), cams=[cams[i % len(cams)]])
    trainer.drain(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for k in ("tile_hist", "tile_scatter", "render_fwd"):
        ms, cnt = prof.read(k); prof.enable(k, False); res[k] = ms / max(cnt, 1) * 1e3
    print("morton-ordered model" if permute else "model order as generated", "%.1f steps/s" % (n / dt), {k: round(v, 1) for k, v in res.items()}, flush=True)
    del scene, g, trainer
