"""bench.py's pure render-FPS leg at several lengths (passes over the 59-pose trajectory) and stream counts."""
import importlib, importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
R = importlib.import_module("iclr2025_3d-mom_amd.gaussian_renderer")
DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
scene, g, trainer, op = bench.build_state(bench.CONFIGS["c2"], torch.device("cuda", 0), fused=True, gc_freeze=True)
if len(sys.argv) > 1:
    for i in range(int(sys.argv[1])): trainer.step(5001 + i, cams=[trainer.cams[i]])
    trainer.drain(); torch.cuda.synchronize()
cams = scene.getVideoCameras_side()
for c in cams: c.device_tensors(g._xyz.device)
DGR.set_sync_mode("async")
kw = dict(stage="fine", cam_type=scene.dataset_type, delta_scale=trainer.delta_scale)
with torch.no_grad():
    for streams in [int(x) for x in os.environ.get("FPS_STREAMS", "1,3,1,3").split(",")]:
        R.set_render_streams(streams)
        for c in cams[:12]: R.render(c, g, trainer.pipe, trainer.background, **kw)
        torch.cuda.synchronize()
        for passes in (2, 8):
            t0 = time.perf_counter()
            for _ in range(passes):
                for c in cams: out = R.render(c, g, trainer.pipe, trainer.background, **kw)
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n = passes * len(cams)
            print(f"streams {streams} passes {passes}: {n / dt:.0f} frames/s (host loop alone {n / th:.0f})", flush=True)
R.set_render_streams(1); DGR.set_sync_mode("exact")
if len(sys.argv) > 2:
    import cProfile, pstats
    R.set_render_streams(3); DGR.set_sync_mode("async")
    with torch.no_grad():
        for c in cams[:12]: R.render(c, g, trainer.pipe, trainer.background, **kw)
        torch.cuda.synchronize()
        pr = cProfile.Profile(); pr.enable()
        for _ in range(4):
            for c in cams: R.render(c, g, trainer.pipe, trainer.background, **kw)
        pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.strip_dirs()
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:22]
    nfr = 4 * len(cams)
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print(f"{tt / nfr * 1e6:8.1f} us {ct / nfr * 1e6:8.1f} us {nc / nfr:6.1f}  {fn}:{line}({name})")
    R.set_render_streams(1); DGR.set_sync_mode("exact")
