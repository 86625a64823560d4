#!/usr/bin/env python
"""bench.py -- 4DGS fine-stage training steps/s on MI355X (BASELINE.json metric), config
"200k Gaussians, 60 frames, 960x540, HexPlane on" (BASELINE.json configs[1]) on the synthetic scene of
SURVEY.md section 8(d).  One step = one full iteration of the reference loop (train_4DGS.py:119-301):
LR update, render (HexPlane + MLP deformation -> rasterizer), L1 + plane regularisers, backward, densification
statistics, Adam on every Gaussian parameter and the deformation field.

    python bench.py --gpus N --steps K --warmup W
prints ONE JSON line (rank 0).  N > 1: one process per GPU -- started by `python -m torch.distributed.run`, or, when no
launcher set RANK / WORLD_SIZE, by this script itself (launch.py: the parent spawns N ranks before anything touches the GPU
and relays rank 0's line).  --shard camera (default): every rank renders a different camera of the same step and the parameter
gradients are all-reduced over RCCL (weak scaling: per-GPU work fixed); --shard tile-row: one camera split by tile rows
(strong scaling).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c2": dict(P=200_000, F=60, W=960, H=540, time_res=50, name="200k Gaussians, 60 frames, 960x540, HexPlane on"),
    "c3": dict(P=1_000_000, F=120, W=1920, H=1080, time_res=100, name="1M Gaussians, 120 frames, 1920x1080"),
    "c1": dict(P=5_000, F=8, W=256, H=256, time_res=50, name="5k Gaussians, 8 frames, 256x256"),
    # BASELINE configs[4]'s per-GPU model size (the camera-batch shard replicates the model); few frames keep the host's
    # share of the ground-truth images small.  A scale check, not a bench line.
    # BASELINE configs[4]'s per-GPU model (the camera-batch shard replicates the model): SURVEY 8d's S(4 000 000, 240, 1920, 1080)
    # with time resolution 250.  The 245 cameras share a bank of 8 ground-truth images (content does not affect timing).
    "c5": dict(P=4_000_000, F=240, W=1920, H=1080, time_res=250, name="4M Gaussians, 240 frames, 1920x1080"),
}


CAMERA_STRIDE = 17


def build_state(cfg, device, fused=False, lambda_dssim=0.0, gc_freeze=False):
    import torch
    pkg = importlib.import_module("iclr2025_3d-mom_amd")
    A = importlib.import_module("iclr2025_3d-mom_amd.arguments")
    S = importlib.import_module("iclr2025_3d-mom_amd.scene")
    T = importlib.import_module("iclr2025_3d-mom_amd.train")
    args, lp, op, pp, hp = A.default_args(time_resolution=cfg["time_res"])
    op.lambda_dssim = lambda_dssim      # 0 is the reference's default; 0.2 is the "SSIM/L1" loss of the north star
    torch.manual_seed(6666)
    scene = S.SyntheticScene(cfg["P"], cfg["F"], cfg["W"], cfg["H"], seed=6666)
    g = S.GaussianModel(lp.sh_degree, hp, device=device)
    scene.init_gaussians(g)
    scene.make_trained_like(g)
    trainer = T.Trainer(scene, g, op, hp, pp, stage="fine", delta_scale=1, sync_every_step=False, fused=fused, gc_freeze=gc_freeze)
    return scene, g, trainer, op


def cpu_baseline(cfg, budget_s=45.0):
    """The oracle (C rasterizer restatement + the reference's torch-op sequence on the CPU) timed on this host's
    cores on a bounded sample of the same workload."""
    import torch
    from oracle import cpu_backend
    from oracle import raster_oracle as ro
    # 16 threads is the fastest setting measured on the 2 x EPYC 9575F host of the GPU box (8: 2.4 s, 16: 1.4 s,
    # 32: 1.6 s, 64: 2.5 s, 128: 4.5 s per step -- float `omp atomic` and torch's small ops stop scaling)
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    ro.set_threads(cores)
    with cpu_backend.installed():
        scene, g, trainer, op = build_state(cfg, "cpu")
        for i in range(3):                      # median of >= 20 steps after 3 warm-up
            trainer.step(5001 + i)
        times = []
        t_all = time.time()
        while len(times) < 20 and (len(times) < 5 or time.time() - t_all < budget_s):
            t0 = time.time()
            trainer.step(5004 + len(times))
            times.append(time.time() - t0)
        times.sort()
        dt = times[len(times) // 2]
    return {"value": 1.0 / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample_short": f"median of {len(times)} steps of the same scene after 3 warm-up, {dt:.2f} s/step",
            "sample": f"median of {len(times)} fine-stage steps of the same scene after 3 warm-up ({dt:.2f} s/step; "
                      f"min {times[0]:.2f}, max {times[-1]:.2f})",
            "protocol": "BASELINE.md section 2: median of >= 20 steps after 3 warm-up" + ("" if len(times) >= 20 else
                        f"; cut at {len(times)} steps by the {budget_s:.0f} s budget of the default run"),
            "host": host_description(),
            "note": "threads = the fastest setting measured on this host class, not its core count: the restatement's "
                    "backward uses float `omp atomic` and torch's small CPU ops stop scaling (8: 2.4, 16: 1.4, 32: 1.6, 64: 2.5, "
                    "128: 4.5 s per step on 2 x EPYC 9575F)"}


def host_description():
    """lscpu model / sockets / cores of the host the CPU baseline ran on (BASELINE.md protocol: stated with every result)."""
    import subprocess
    d = {"logical_cpus": os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
        for line in txt.splitlines():
            k, _, v = line.partition(":")
            k = k.strip()
            if k in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core"):
                d[k] = v.strip()
    except Exception:
        pass
    return d


def _newest_profile(suffix, cfg):
    """The newest committed profiles/*<suffix> collected on exactly this workload, and whether the library it was collected on is
    the one running now (mom_version() carries a hash of the kernel sources)."""
    import glob
    N = importlib.import_module("iclr2025_3d-mom_amd._native")
    running = N.lib().mom_version().decode()
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*" + suffix)))
    for f in reversed(files):
        with open(f) as fh:
            doc = json.load(fh)
        if doc.get("workload") != cfg["name"]:
            continue
        return doc, os.path.basename(f), doc.get("lib_version") == running
    return None, None, False


def measured_traffic(kernel, cfg):
    """(HBM bytes per launch of `kernel`, note) from the committed rocprofv3 PMC summary (profiles/*_pmc_traffic.json: two
    separate --pmc passes, unit and gfx950 corrections applied there) -- PMC counters cannot be collected from inside this
    process.  The figure is reported only while the running library is the build the counters were collected on; otherwise
    (None, {"traffic_stale": True, ...})."""
    doc, name, fresh = _newest_profile("_pmc_traffic.json", cfg)
    if doc is None:
        return None, {"traffic_source": None}
    k = doc.get("kernels", {}).get(kernel)
    if not fresh or not k:
        return None, {"traffic_stale": True, "traffic_source": name, "traffic_collected_on": doc.get("lib_version"),
                      "traffic_of_that_build": None if not k else k["traffic_bytes"]}
    return k["traffic_bytes"], {"traffic_source": name, "traffic_collected_on": doc.get("lib_version")}



def sq_counters(kernel, cfg):
    """(entry, note): that kernel's means from the committed SQ / GRBM counter summary (profiles/*_sq.json: SQ_INSTS_VALU per
    launch, the clock measured in that pass, the issue fraction inside that pass) -- reported as live only while the running
    library is the build the counters were collected on (same rule as the traffic)."""
    doc, name, fresh = _newest_profile("_sq.json", cfg)
    if doc is None:
        return None, {"sq_source": None}
    k = doc.get("kernels", {}).get(kernel + "_kernel")
    note = {"sq_source": name, "sq_collected_on": doc.get("lib_version")}
    if not k or not fresh:
        note["sq_stale"] = True
        return None, note
    return k, note


def step_bytes(P, R, npix, lambda_dssim=0.0):
    return importlib.import_module("iclr2025_3d-mom_amd.profiling").step_bytes(P, R, npix, lambda_dssim)


def step_roofline(P, R_binned, R_ref, npix, seconds_per_step, lambda_dssim=0.0):
    return importlib.import_module("iclr2025_3d-mom_amd.profiling").step_roofline(P, R_binned, R_ref, npix, seconds_per_step, lambda_dssim)


def metric_name():
    """BASELINE.json's metric string, verbatim (it travels with the repo); the literal is the same text."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, KeyError, ValueError):
        return "4DGS train-steps/sec @200k Gaussians, 960\u00d7540, 60 frames; render FPS"


def pairs_evaluated(fs):
    """SURVEY 8(d)'s Q for the frame the fused step `fs` rendered last (call after a synchronisation): evaluated pixel-Gaussian pairs
    = sum over tiles of 256 x (list entries the tile's block walks before it exits).
      q_fwd     the forward: its block exits at the start of the 256-entry round in which every pixel is done (forward.cu:305-312;
                render_fwd keeps that round structure and stores the count per tile: MomRasterLayout.img_tile_walked);
      q_bwd     the backward as THIS library runs it: a tile's list ends at its last contributor (max n_contrib over the tile);
      q_bwd_no_exit  256 x the instances: the reference's backward has no block exit and walks every entry (backward.cu:497-505);
      alpha_pairs    sum over pixels of n_contrib: the pairs that lie in front of their pixel's last contributor, i.e. the ones
                the backward's per-thread test (backward.cu:519-520) lets through to the exponent.
    With keep_all_tiles the lists are the reference's, so q_fwd and q_bwd_no_exit are the reference's own Q."""
    import ctypes as C
    import torch
    N = importlib.import_module("iclr2025_3d-mom_amd._native")
    P = fs.P
    W, H = fs._wh
    lay = N.MomRasterLayout()
    N.lib().mom_raster_layout(P, W, H, 0, C.byref(lay))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    base = fs.img[(-fs.img.data_ptr()) % 256:]
    walked = base[lay.img_tile_walked:lay.img_tile_walked + gx * gy * 4].view(torch.int32)
    nc = base[lay.img_n_contrib:lay.img_n_contrib + W * H * 4].view(torch.int32).view(H, W)
    pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int32, device=nc.device)
    pad[:H, :W] = nc
    last = pad.view(gy, 16, gx, 16).amax(dim=(1, 3))
    return {"q_fwd": 256 * int(walked.long().sum()), "q_bwd": 256 * int(last.long().sum()),
            "q_bwd_no_exit": 256 * int(fs.nr_host[0]), "alpha_pairs": int(nc.long().sum())}


Q_IS = ("SURVEY 8d: evaluated pixel-Gaussian pairs = sum over tiles of 256 x (list entries walked before the tile's block exits), mean "
        "over the sampled cameras.  q_fwd: the forward (exit when every pixel is done, rounds of 256, forward.cu:305-312); q_bwd: this "
        "library's backward (the list ends at the tile's last contributor); q_bwd_no_exit: 256 x instances (the reference's backward "
        "walks every entry); alpha_pairs: sum of n_contrib over the pixels.  *_ref: on the reference's lists (keep_all_tiles)")


EXPLAIN = {
    "value": "training steps per second over exactly --steps steps after --warmup, whole job (camera-batch shard: summed over ranks); full "
             "iteration in the timed region: LR -> render (field, projection, binning, compositing) -> L1 (+ plane regularisers) -> backward "
             "-> densification statistics -> Adam; inputs resident in HBM; verified applied by Trainer.drain() before the clock stops",
    "config.instances_R": "the reference's instance count (every tile of every splat's rectangle), mean over the cameras of the timed steps",
    "config.instances_binned": "what the default binning keeps: instances that can reach alpha >= 1/255 in their tile",
    "config.pairs_Q": Q_IS,
    "config.legs": "numbers only; every leg is a reading of the same metric.  steady: 200 steps after 50 (SURVEY 8d); with_ssim: lambda_dssim 0.2; "
                   "keep_all_tiles: the reference-identical binning (integer indices bit-exact); via_render_api / _exact: gaussian_renderer.render() "
                   "+ torch loss + loss.backward() + optimizer.step() as train_4DGS.py:189-297 drives the modules, in async and in the drop-in's "
                   "default exact sync mode (host_ms = wall time of the Python loop that enqueues the window); c1 / c3 / c5: BASELINE configs "
                   "[0] / [2] / [4]'s per-GPU model (c5 with the trainer's own prune round at iteration 5100 inside the window: before / boundary / "
                   "after; c5_cold_allocator: the same without the allocator prewarm, i.e. a process's first round); frac = step-level HBM roofline "
                   "on processed instances",
    "config.legs.render_fps": "no-grad render() over the 59-pose side trajectory (render_4DGS.py:60-71): two_streams / one_stream (async sizing) / "
                              "one_stream_exact (the drop-in's default sync mode), images left on the device; as_scripted = render_set with every "
                              "frame written as PNG through the asynchronous writer, FPS = (frames - 1) / seconds as render_4DGS.py:71 prints it; "
                              "blocking = the reference's order (each PNG encoded inside the loop, render_4DGS.py:64)",
    "roofline": "dominant kernel (render_bwd).  bound valu: achieved = SQ_INSTS_VALU per launch (committed --pmc pass of the same workload and "
                "library build) / live launch duration (HIP events on the launch stream, a leg of its own after the headline region); peak = 1024 "
                "SIMDs x clock / 2 cycles (the issue rate behind the 157 TFLOP/s fp32 vector peak).  hbm_*: 84 R' + 24 N algorithmic bytes per "
                "launch against 8 TB/s; traffic = HBM bytes per launch from the committed PMC summary (null when collected on another build); "
                "useful_flop_frac = 70 FLOP x q_bwd / duration / 157 TFLOP/s (SURVEY 8d); alpha_pair_flop_frac on the pairs that reach the exponent",
    "roofline.step": "whole step: algorithmic bytes (SURVEY 8d per-unit figures x units, profiling.step_bytes) / ms_per_step against 8 TB/s; "
                     "frac_on_reference_R prices the reference's instance count instead of the processed one",
    "cpu_baseline": "oracle Trainer (C restatement of the rasterizer + the reference's torch-op sequence) on this host's cores: median of up to 20 "
                    "fine-stage steps after 3 warm-up; threads = the fastest setting measured on 2 x EPYC 9575F (8: 2.4, 16: 1.4, 32: 1.6, 64: 2.5, "
                    "128: 4.5 s per step -- float `omp atomic` and torch's small ops stop scaling), not the core count",
    "full record": "the complete document with every sub-field goes to stderr behind the tag BENCH_FULL and, when MOM_BENCH_FULL names a path, "
                   "into that file (profiles/r06_bench_full.json is such a file)",
}


def _r(x, n=4):
    """Numbers of the compact line: n significant digits are what a bench figure carries."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if x == x and abs(x) != float("inf") else None
    return x


def compact_line(full):
    """The ONE JSON line of stdout: < 4 KB, numbers only below `config.legs`, so that the driver's record (which keeps `config`,
    `roofline` and `cpu_baseline` whole and truncates strings) carries every figure BASELINE.md's table quotes.  Prose lives in
    EXPLAIN (`bench.py --explain`); the complete document goes to stderr / MOM_BENCH_FULL."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: full[k] for k in keep}
    c = full["config"]
    cfg = {k: c[k] for k in ("workload", "gaussians", "frames", "width", "height", "step_path", "lambda_dssim", "parallelism",
                             "ranks_seen", "host_sync", "gaussians_at_end", "steps_replayed_after_overflow") if k in c}
    cfg["instances_R"] = _r(c.get("instances_R"), 6)
    cfg["instances_binned"] = _r(c.get("instances_binned"), 6)
    cfg["final_loss"] = _r(c.get("final_loss"), 5)
    q = (c.get("pairs_Q") or {}).get("processed")
    if q:
        cfg["pairs_Q"] = {k: _r(v, 4) for k, v in q.items()}

    def leg(d):
        o = {"value": _r(d["value"]), "ms": _r(d["ms_per_step"])}
        if "host_enqueue_ms_per_step" in d:
            o["host_ms"] = _r(d["host_enqueue_ms_per_step"])
        rs = d.get("roofline_step")
        if rs:
            o["frac"] = _r(rs["frac"], 3)
        dw = d.get("densify_in_window")
        if dw:
            sg = dw["segments"]
            o.update(gaussians_before=dw["gaussians_before"], gaussians_after=dw["gaussians_after"],
                     boundary_ms=_r(sg["boundary"]["ms"]), before=_r(sg["before"]["steps_per_s"]), after=_r(sg["after"]["steps_per_s"]),
                     prewarmed=bool(dw.get("allocator_prewarmed_bytes")))
        return o

    legs = {}
    for k in ("steady", "with_ssim", "keep_all_tiles", "via_render_api", "via_render_api_exact"):
        if full.get(k):
            legs[k] = leg(full[k])
    for k, d in (full.get("other_configs") or {}).items():
        legs[k] = leg(d)
    rf = full.get("render_fps")
    if rf:
        legs["render_fps"] = {"two_streams": _r(rf["value"]), "one_stream": _r(rf["one_stream"]["value"]),
                              "one_stream_exact": _r(rf["one_stream_exact"]["value"])}
        if "as_scripted" in rf:
            legs["render_fps"].update(as_scripted=_r(rf["as_scripted"]["value"]), blocking=_r(rf["as_scripted_blocking"]["value"]))
    if legs:
        cfg["legs"] = legs
    out["config"] = cfg
    rl = full.get("roofline")
    if rl:
        o = {k: _r(rl.get(k), 5) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches",
                                            "wave_insts_per_launch", "useful_flop_frac", "alpha_pair_flop_frac")}
        if o["unit"] and len(o["unit"]) > 20:
            o["unit"] = "Ginst/s"
        hb = rl.get("hbm") or {}
        o.update(hbm_achieved=_r(hb.get("achieved")), hbm_peak=hb.get("peak"), hbm_frac=_r(hb.get("frac"), 3),
                 hbm_bytes_per_launch=_r(hb.get("algorithmic_bytes_per_launch"), 5))
        o["counters_live"] = not (rl.get("sq_stale") or rl.get("traffic_stale"))
        o["counters_from"] = rl.get("sq_source")
        out["roofline"] = o
    else:
        out["roofline"] = None
    rs = full.get("roofline_step")
    if rs and out["roofline"] is not None:
        out["roofline"]["step"] = {"bound": "hbm", "achieved": _r(rs["achieved"]), "peak": rs["peak"], "unit": rs["unit"], "frac": _r(rs["frac"], 3),
                                   "frac_on_reference_R": _r(rs["frac_on_reference_R"], 3), "bytes_per_step": _r(rs["algorithmic_bytes_per_step"], 5)}
    elif rs:
        out["roofline_step"] = {"bound": "hbm", "achieved": _r(rs["achieved"]), "peak": rs["peak"], "unit": rs["unit"], "frac": _r(rs["frac"], 3)}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": cb.get("sample_short") or cb["sample"][:110], "host": (cb.get("host") or {}).get("Model name")}
    return out


def mean_q(rows):
    return {k: sum(r[k] for r in rows) / len(rows) for k in rows[0]} if rows else None


def render_fps(scene, g, pp, background, delta_scale, passes=8):
    """Second half of the metric: no-grad gaussian_renderer.render() over the reference's 59-pose `side` trajectory
    (render_4DGS.py:88 -> render_set), three ways: `value` = pure (images left on the device: the reference loop without its PNG
    writer); `as_scripted` = the whole of render_set with every frame written as a PNG, through render.py's asynchronous writer;
    `as_scripted_blocking` = the same with the reference's order (encode each PNG inside the loop, render_4DGS.py:64)."""
    import shutil
    import tempfile
    import torch
    R = importlib.import_module("iclr2025_3d-mom_amd.gaussian_renderer")
    DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
    own = importlib.import_module("iclr2025_3d-mom_amd.render")
    cams = scene.getVideoCameras_side()
    dev = g._xyz.device
    for c in cams:
        c.device_tensors(dev)
    cfg_hw = (int(cams[0].image_height), int(cams[0].image_width))
    DGR.set_sync_mode("async")          # size the binning buffer from earlier frames; overflowed frames are rendered again below

    def pure(streams):
        """`passes` x the trajectory through render(), images left on the device; streams > 1: consecutive frames on alternating
        streams (gaussian_renderer.set_render_streams, fused_render.FusedRenderPool) -- the mode render.py's render_set uses."""
        R.set_render_streams(streams)
        try:
            with torch.no_grad():
                for c in (cams * 2)[:12 * streams]:
                    R.render(c, g, pp, background, stage="fine", cam_type=scene.dataset_type, delta_scale=delta_scale)
                fr = g._fused_render_pool if streams > 1 else g._fused_render
                fr.overflowed()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(passes):
                    for c in cams:
                        out = R.render(c, g, pp, background, stage="fine", cam_type=scene.dataset_type, delta_scale=delta_scale)["render"]
                bad = fr.overflowed()                       # waits for every frame's flag
                R.set_render_streams(1)
                for _ in bad:                               # an overflowed frame counts only once it has been rendered completely
                    DGR.set_sync_mode("exact")
                    out = R.render(cams[0], g, pp, background, stage="fine", cam_type=scene.dataset_type, delta_scale=delta_scale)["render"]
                    DGR.set_sync_mode("async")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            assert torch.isfinite(out).all()
            return passes * len(cams), dt, len(bad)
        finally:
            R.set_render_streams(1)

    def exact_pure(passes=3):
        """The same loop in the drop-in's DEFAULT sync mode (every frame waits for its own instance count, as the reference's
        cudaMemcpy does, rasterizer_impl.cu:282) on the caller's stream: what an unchanged render_4DGS.py gets from render() itself."""
        DGR.set_sync_mode("exact")
        try:
            with torch.no_grad():
                for c in cams[:8]:
                    R.render(c, g, pp, background, stage="fine", cam_type=scene.dataset_type, delta_scale=delta_scale)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(passes):
                    for c in cams:
                        out = R.render(c, g, pp, background, stage="fine", cam_type=scene.dataset_type, delta_scale=delta_scale)["render"]
                torch.cuda.synchronize()
                return passes * len(cams), time.perf_counter() - t0
        finally:
            DGR.set_sync_mode("async")

    try:
        n1, dt1, bad1 = pure(1)
        ne, dte = exact_pure()
        n, dt, bad = pure(own.RENDER_STREAMS)
        res = {"value": n / dt, "unit": "frames/s", "frames": n, "ms_per_frame": 1e3 * dt / n, "frames_rendered_again": bad,
               "streams": own.RENDER_STREAMS,
               "trajectory": "side, 59 poses (test_trajectory/side_{R,t}_list, last pose dropped)",
               "mode": "no-grad render(), deformation on, images kept on the device (no PNG writer); consecutive frames on "
                       f"{own.RENDER_STREAMS} alternating streams (gaussian_renderer.set_render_streams: the mode render.py's "
                       "render_set runs in), every frame complete before the clock stops",
               "one_stream": {"value": n1 / dt1, "unit": "frames/s", "frames": n1, "ms_per_frame": 1e3 * dt1 / n1,
                              "frames_rendered_again": bad1,
                              "what": "every frame on the caller's current stream (the drop-in's default stream use), binning "
                                      "capacity from earlier frames (async sync mode)"},
               "one_stream_exact": {"value": ne / dte, "unit": "frames/s", "frames": ne, "ms_per_frame": 1e3 * dte / ne,
                                    "what": "the drop-in's default sync mode too: every frame waits for its own instance count "
                                            "(polled from pinned memory) before it is composited -- render() as an unchanged "
                                            "render_4DGS.py drives it, without its PNG writer"}}
        tmp = tempfile.mkdtemp(prefix="mom_bench_render_")
        try:
            # one writer for the whole script, as render_sets() would keep it over its four trajectories: pinned ring and encoder
            # threads exist before the first frame
            writer = own.AsyncPNGWriter(cfg_hw[0], cfg_hw[1])
            own.render_set(tmp, "warm", 0, cams, g, pp, background, scene.dataset_type, delta_scale=delta_scale, video=False, writer=writer)
            a = own.render_set(tmp, "side", 0, cams, g, pp, background, scene.dataset_type, delta_scale=delta_scale, video=False,
                               writer=writer)
            writer.close()
            DGR.set_sync_mode("exact")
            b = own.render_set(tmp, "side_blocking", 0, cams, g, pp, background, scene.dataset_type, delta_scale=delta_scale,
                               video=False, scripted=True)
            res["as_scripted"] = {"value": a["fps"], "unit": "frames/s", "frames": a["frames"],
                                  "what": "render_set with every frame written to frame_result/side/%05d.png: quantisation kernel, "
                                          "async copy to pinned host memory, PNG encoder threads; clock stops when the last file is "
                                          "written; FPS = (frames - 1) / seconds as render_4DGS.py:71 prints it"}
            res["as_scripted_blocking"] = {"value": b["fps"], "unit": "frames/s", "frames": b["frames"],
                                           "what": "the reference's order: each PNG encoded inside the render loop (render_4DGS.py:64)"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    finally:
        DGR.set_sync_mode("exact")
    return res


def side_leg(cfg, dev, path, steps, warmup, sync_mode="async", keep_all_tiles=False, with_densify=False, prewarm_allocator=True):
    """One more reading of the metric on a fresh model: `path` fused | autograd on workload `cfg`, `steps` timed steps after
    `warmup`, with the step-level roofline on the instances the steps process and, beside it, on the reference's count (both
    sampled on eight cameras in exact mode).
    keep_all_tiles: the timed steps bin every tile of every splat's rectangle like the reference (rasterizer_impl.cu:70-111), so
    that num_rendered, the tile lists and n_contrib are the reference's bit for bit -- the configuration whose integer indices
    tests/test_raster_gpu.py compares with the oracle.
    sync_mode "exact" (autograd path): the drop-in's DEFAULT -- every forward waits for its own instance count, as the
    reference's cudaMemcpy at rasterizer_impl.cu:282 does; this is what an unchanged train_4DGS.py gets.
    with_densify: consecutive iteration numbers, so that the trainer's own densify / prune round (every 100 iterations,
    train_4DGS.py:264-290) falls inside the timed window."""
    import torch
    DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
    # gc.freeze() after setup, as the headline leg: the cyclic collector's full passes walk the long-lived heap (scene, cameras,
    # optimizer state) every few hundred iterations and cost the host-bound API legs 0.24 ms per step on average
    # (tools/probe/ipf_probe.py: 1.53 against 1.30 ms of host time per step); nothing the steps compute depends on it
    scene, g, trainer, op = build_state(cfg, dev, fused=(path == "fused"), gc_freeze=True)
    cams = trainer.cams
    for c in cams:
        c.device_tensors(dev)
    sample = cams[::max(1, len(cams) // 8)][:8]
    DGR.set_sync_mode("exact")
    counts = {True: [], False: []}
    qs = {True: [], False: []}

    def set_keep(flag):
        if trainer.fused is not None:
            trainer.fused.keep_all_tiles = flag
            trainer.fused.exact_next()
        else:
            DGR.set_keep_all_tiles(flag)

    try:
        for keep in (True, False):
            set_keep(keep)
            for i, c in enumerate(sample):
                if trainer.fused is not None:
                    trainer.fused.exact_next()
                trainer.step(5001 + i, cams=[c])
                torch.cuda.synchronize()
                counts[keep].append(int(trainer.fused.nr_host[0]) if trainer.fused is not None else DGR.last_num_rendered())
                if trainer.fused is not None:
                    qs[keep].append(pairs_evaluated(trainer.fused))
    finally:
        set_keep(keep_all_tiles)
    r_ref = sum(counts[True]) / len(counts[True])
    r_binned = sum(counts[False]) / len(counts[False])
    r_proc = r_ref if keep_all_tiles else r_binned
    # with_densify: BASELINE configs[4]'s "densify/prune every 100 iters" -- the cadence of the reference's argparse default
    # (arguments/__init__.py:146 of the reference; the dnerf_default overlay the scripts load prunes every 8000, which default_args
    # mirrors and which made this leg's round a no-op in round 4).  The window walks consecutive iteration numbers around 5100: `pre`
    # steps, the boundary iteration itself (drain + statistics -> prune -> compaction of every parameter and of Adam's state), `post`
    # steps on the compacted model.  The statistics the round acts on are the ones the steps themselves accumulated (max_radii2D
    # of the cameras seen since the model was built; the trained-like scene has splats above the 20-pixel screen-size limit).
    if with_densify:
        op.pruning_interval = 100
    pre = (steps - 1) // 2 if with_densify else 0
    it0 = 5100 - pre - warmup if with_densify else 5011

    def it(i):
        return it0 + (i if with_densify else i % 80)

    p_start = int(g.get_xyz.shape[0])
    segs = None
    try:
        if trainer.fused is None and sync_mode == "async":
            DGR.set_sync_mode("async", capacity_hint=int(max(counts[keep_all_tiles]) * 1.6) + 65536)
        for i in range(warmup):
            trainer.step(it(i), cams=[cams[(CAMERA_STRIDE * i) % len(cams)]])
        if with_densify:
            # one-time library initialisation out of the window: densify_and_split's torch.bmm is this process's first GEMM (rocBLAS
            # loads its kernels: ~0.2 s, once per process), torch.normal its first random draw
            z = torch.zeros(4, 3, 3, device=dev)
            torch.bmm(z, z)
            torch.normal(mean=torch.zeros(4, 3, device=dev), std=torch.ones(4, 3, device=dev))
            # ... and the caching allocator in the state of a process that has been through a round before (at a 100-iteration
            # cadence: every round but the first).  A round makes new parameter and moment tensors before it lets go of the old
            # ones; in a process whose cache was just emptied (the leg before this one ends with empty_cache()) that is ~3 GB from
            # the driver, 13 ms on one box and 130 on the next (profiles/r05_bench_full.json history), and it says nothing about the
            # round itself.  One allocation of that size, freed again, leaves the cache what round two would find.
            model_bytes = sum(p.numel() * p.element_size() for grp in g.optimizer.param_groups for p in grp["params"]
                              if p.dim() and p.shape[0] == p_start)
            prewarm = int(3.2 * model_bytes) if prewarm_allocator else 0
            if prewarm:
                torch.empty(prewarm, dtype=torch.uint8, device=dev)
        trainer.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = None
        marks = []
        for i in range(steps):
            if with_densify and i in (pre, pre + 1):
                # the segments' edges: the boundary iteration drains by itself (Trainer._boundary), the synchronisations here add
                # nothing to it; the one after it is the price of reporting the boundary's own milliseconds
                trainer.drain()
                torch.cuda.synchronize()
                marks.append((time.perf_counter(), int(g.get_xyz.shape[0])))
            loss = trainer.step(it(warmup + i), cams=[cams[(CAMERA_STRIDE * (warmup + i)) % len(cams)]])
        t_enq = time.perf_counter() - t0           # the host's share: every launch of the window is enqueued (exact mode: its waits included)
        torch.cuda.synchronize()                   # (spinning wait first, as in the headline's timed())
        trainer.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if hasattr(loss, "tensor"):
            loss = loss.tensor()                   # (the value's read-back is not part of a step: behind the clock, as in the headline)
        if with_densify:
            (ta, na), (tb, nb) = marks
            post = steps - pre - 1
            segs = {"before": {"steps": pre, "gaussians": na, "steps_per_s": pre / (ta - t0)},
                    "boundary": {"iteration": it(warmup + pre), "ms": 1e3 * (tb - ta), "gaussians_before": na, "gaussians_after": nb},
                    "after": {"steps": post, "gaussians": nb, "steps_per_s": post / (t0 + dt - tb)}}
    finally:
        DGR.set_sync_mode("exact")
        set_keep(False)
    assert torch.isfinite(loss).all(), f"loss is not finite ({cfg['name']}, {path})"
    out = {"workload": cfg["name"], "step_path": path, "value": steps / dt, "unit": "steps/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt / steps, "host_enqueue_ms_per_step": 1e3 * t_enq / steps,
           "host_enqueue_is": "wall time of the Python loop that enqueues the window's steps, before the final drain: close to "
                              "ms_per_step means the host sets the pace (or, in exact mode, waits for the GPU every step)",
           "instances_R_mean": r_ref, "instances_binned_mean": r_binned,
           "keep_all_tiles": bool(keep_all_tiles), "final_loss": float(loss),
           "time_resolution": cfg["time_res"], "steps_replayed_after_overflow": int(trainer.replayed),
           "host_sync": "device-gated async (fused step)" if path == "fused" else sync_mode,
           "python_gc": "gc.freeze() after setup (Trainer(gc_freeze=True)), as in the headline leg",
           "roofline_step": step_roofline(cfg["P"], r_proc, r_ref, cfg["W"] * cfg["H"], dt / steps)}
    if qs[True]:
        out["pairs_Q"] = {"processed": mean_q(qs[keep_all_tiles]), "reference_lists": mean_q(qs[True]), "what": Q_IS}
    if with_densify:
        p_end = int(g.get_xyz.shape[0])
        out["densify_in_window"] = {"iterations": [it(warmup), it(warmup + steps - 1)], "gaussians_before": p_start,
                                    "gaussians_after": p_end, "pruning_interval": int(op.pruning_interval),
                                    "allocator_prewarmed_bytes": prewarm,
                                    "allocator_prewarm_is": "one allocation of ~3.2 x the Gaussian parameters' bytes made and freed before the window: "
                                                            "the caching allocator as a process finds it in every round but its first",
                                    "segments": segs,
                                    "what": "the trainer's own round at iteration 5100 (train_4DGS.py:264-290 gates, pruning_interval "
                                            "100 = BASELINE configs[4]) is inside the timed window: `value` covers before + boundary + "
                                            "after; 4 M >= 360 000 closes the reference's densify gate (train_4DGS.py:275), so the round "
                                            "prunes (opacity / screen-size / world-size masks, gaussian_model.py:681-692 of the reference)"}
        assert p_end != p_start, f"the densify/prune round at iteration 5100 left the model at {p_start} Gaussians: the leg measured a no-op"
    del scene, g, trainer
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-mode", default="async", choices=["async", "exact"])
    ap.add_argument("--roofline-kernel", default="render_bwd")
    ap.add_argument("--shard", default="camera", choices=["camera", "tile-row"],
                    help="N > 1: camera = one camera per GPU per step (weak scaling, the reference's batch axis); "
                         "tile-row = one camera per step split over the GPUs by rows of 16-pixel tiles (strong scaling)")
    ap.add_argument("--shard-adam", action="store_true",
                    help="camera-batch shard: reduce-scatter the appearance gradients, run Adam on this rank's 1/N slice, all-gather the "
                         "updated parameters (SURVEY 8e's second design: same bytes, 1/N of the Adam pass -- what pays at 4 M Gaussians)")
    ap.add_argument("--with-densify", action="store_true",
                    help="walk consecutive iteration numbers so that the trainer's own densification (every 100 iterations) "
                         "fires inside the timed region; not the headline configuration (SURVEY 8d excludes it)")
    ap.add_argument("--lambda-dssim", type=float, default=0.0,
                    help="weight of the SSIM loss term in the headline value (0 = the reference's default)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra single-GPU legs (steady state, training with lambda_dssim 0.2, render FPS)")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="skip via_render_api and other_configs (profiling runs: their kernels would mix into the per-kernel averages)")
    ap.add_argument("--steady-steps", type=int, default=200, help="steps of the steady-state leg (SURVEY 8d: >= 200)")
    ap.add_argument("--steady-warmup", type=int, default=50, help="warm-up of the steady-state leg (SURVEY 8d: 50)")
    ap.add_argument("--explain", action="store_true", help="print what every key of the JSON line means, and exit")
    ap.add_argument("--path", default="fused", choices=["fused", "autograd", "autograd-per-op"],
                    help="fused: explicit launch sequence (fused_step.py); autograd: render() + loss.backward(), the way the "
                         "reference's own loop drives the modules (render() is one autograd node, fused_autograd.py); "
                         "autograd-per-op: the same through one autograd node per operator")
    return ap.parse_args(argv)


def main():
    a = parse_args()
    if a.explain:
        for k, v in EXPLAIN.items():
            print(f"{k}:\n    {v}\n")
        return
    # N ranks wanted and no launcher started us: become the launcher -- before anything touches the GPU (launch.py).
    # MOM_BENCH_SPAWN=1 forces that route with one rank too (the only way to exercise it on a one-GPU box).
    launch = importlib.import_module("iclr2025_3d-mom_amd.launch")
    force_spawn = os.environ.get("MOM_BENCH_SPAWN") == "1" and not launch.launched_by_a_launcher()
    if force_spawn:
        os.environ["MOM_FORCE_DIST"] = "1"
    launch.main_or_spawn(a.gpus, os.path.abspath(__file__), sys.argv[1:], force=force_spawn)
    # stdout carries the ONE JSON line and nothing else: everything the mirrored modules print (they print what the reference
    # prints) and everything C libraries write to fd 1 (RCCL's banner) goes to stderr from here on
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    cfg = CONFIGS[a.config]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libmom4d has no CPU path)")
    torch.cuda.set_device(local)
    # MOM_FORCE_DIST=1: run the multi-GPU code path (RCCL process group, DistContext, the step's all-reduces) even with one
    # rank -- the only way to exercise it on a box with a single GPU.
    force_dist = world == 1 and os.environ.get("MOM_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import datetime
        # a collective that cannot complete (a rank died, took another branch) aborts the job after this long instead of hanging it:
        # torch's watchdog tears the communicator down and the rank exits non-zero, which makes launch.spawn_ranks stop the others
        dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=float(os.environ.get("MOM_PG_TIMEOUT_S", "300"))))
    dev = torch.device("cuda", local)
    DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
    scene, g, trainer, op = build_state(cfg, dev, fused=(a.path == "fused"), lambda_dssim=a.lambda_dssim, gc_freeze=True)
    if a.path == "autograd-per-op":
        trainer.pipe.per_op_autograd = True
    cams = trainer.cams
    for c in cams:                       # inputs resident in HBM before the timed region: the cameras' matrices and
        c.device_tensors(dev)            # ground-truth images are uploaded here, not on first use inside it
    par = None
    ranks_seen = 1
    if world > 1 or force_dist:
        par = importlib.import_module("iclr2025_3d-mom_amd.parallel")
        par.attach(trainer, rank, world, mode=a.shard, shard_adam=a.shard_adam or None)
        one_t = torch.ones(1, device=dev)
        dist.all_reduce(one_t)
        ranks_seen = int(one_t[0])
    it0 = 5000  # mid-training iteration numbers: densification statistics on, no densify/reset in the window
    npix = cfg["W"] * cfg["H"]

    def cam_of(i):
        # every rank takes a different camera of the cycle (camera-batch shard); N=1 walks all F+5 cameras.  The walk is STRIDED
        # (17 is coprime to the 13 / 65 / 125 / 245 cameras of the configs): any K consecutive steps sample the whole set evenly, as the
        # reference's random draw does (train_4DGS.py:172-187) -- in list order a 20-step window sat on the five hemisphere views and the
        # first frames, whose steps take 0.92-1.09 ms against 0.87 for the average camera (tools/probe/per_camera.py)
        j = (i * world + rank) if a.shard == "camera" else i
        return cams[(CAMERA_STRIDE * j) % len(cams)]

    def one(i):
        return trainer.step(it0 + 1 + (i if a.with_densify else i % 90), cams=[cam_of(i)])

    def instances_of(i):
        """Instance count R of the camera step i draws (read once, outside any timed region)."""
        return r_of_cam.get(id(cam_of(i)))

    DGR.set_sync_mode("exact")
    one(0)                                                   # sizes the binning buffers
    R = DGR.last_num_rendered() if trainer.fused is None else int(trainer.fused.nr_host[0])
    if a.sync_mode == "async" and trainer.fused is None:
        DGR.set_sync_mode("async", capacity_hint=int(R * 1.6) + 65536)
    # per-camera instance counts (R is data dependent: SURVEY 8d asks for it with every timing).  One untimed pass over the
    # cameras this rank will draw; each step's count is read after a synchronisation.
    # R is the REFERENCE's instance count -- every tile of every splat's rectangle (duplicateWithKeys) -- which is what the
    # algorithmic bytes are defined on; the default binning drops the instances that cannot contribute, and that smaller
    # count is reported beside it (instances_binned).  So: two passes, the first with keep_all_tiles.
    r_of_cam, binned_of_cam = {}, {}
    q_of_cam = {True: {}, False: {}}
    if trainer.fused is not None:
        for keep_all, table in ((True, r_of_cam), (False, binned_of_cam)):
            trainer.fused.keep_all_tiles = keep_all
            trainer.fused.exact_next()
            for i in range(1, 1 + len(cams)):    # every rank makes the same number of calls (the steps hold collectives)
                one(i)
                torch.cuda.synchronize()
                table[id(cam_of(i))] = int(trainer.fused.nr_host[0])
                if a.shard == "camera":          # (a tile-row shard's image state covers the rank's rows only)
                    q_of_cam[keep_all][id(cam_of(i))] = pairs_evaluated(trainer.fused)
        trainer.fused.exact_next()
    prof = importlib.import_module("iclr2025_3d-mom_amd.profiling")

    def timed(first, steps, profile_kernel=None):
        """`steps` training steps bracketed by barrier + synchronize on both sides; max over ranks.  Every step enqueued
        inside the region is verified applied before the clock stops (Trainer.drain: an overflowed step is replayed, never
        dropped)."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if profile_kernel and rank == 0:
            # HIP events around every launch of the dominant kernel: only the roofline leg below asks for them (an event pair costs
            # the stream ~13 us of bubbles, tools/gap_stats.py; the headline region carries none, whatever --steps is)
            prof.enable(profile_kernel, period=1)
        t0 = time.perf_counter()
        loss = None
        for i in range(steps):
            loss = one(first + i)
        # (the device synchronisation FIRST: it spins, while drain()'s wait for its last read-back parks the thread in
        # hipEventSynchronize and is woken up to a millisecond late -- 5 % of a 20-step window, tools/probe/window20.py; the same
        # work is verified either way: every step complete, none skipped by an overflow)
        torch.cuda.synchronize()
        trainer.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if hasattr(loss, "tensor"):
            # the fused step forms its loss VALUE on demand, from accumulators the next step overwrites: taken here, behind the
            # clock (no step has been enqueued since) -- two tiny launches and a read-back that no training step contains were
            # 0.15-0.3 ms of a window, 1.5 % of a 20-step one (tools/probe/window_len.py)
            loss = loss.tensor()
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0])
        rs = [instances_of(first + i) for i in range(steps)]
        rs = [r for r in rs if r is not None]
        bs = [binned_of_cam.get(id(cam_of(first + i))) for i in range(steps)]
        bs = [b for b in bs if b is not None]
        binned_mean[0] = sum(bs) / len(bs) if bs else None
        for keep_all in (True, False):
            rows = [q_of_cam[keep_all].get(id(cam_of(first + i))) for i in range(steps)]
            q_mean[keep_all] = mean_q([r for r in rows if r is not None])
        return dt, loss, (sum(rs) / len(rs) if rs else float(R))

    binned_mean = [None]
    q_mean = {True: None, False: None}

    scale = world if a.shard == "camera" else 1
    nxt = 1 + len(cams)
    steady = None
    if not a.no_extra:
        # SURVEY 8(d)'s reading of the metric: >= 200 steps after 50 warm-up.  It runs BEFORE the headline region, so the
        # headline's K steps are measured on a warm device whatever --steps / --warmup the caller chose.
        for i in range(a.steady_warmup):
            one(nxt + i)
        nxt += a.steady_warmup
        dts, loss_s, r_mean = timed(nxt, a.steady_steps)
        nxt += a.steady_steps
        assert torch.isfinite(loss_s).all(), "loss is not finite (steady leg)"
        steady = {"value": a.steady_steps * scale / dts, "unit": "steps/s", "steps": a.steady_steps, "warmup": a.steady_warmup,
                  "ms_per_step": 1e3 * dts / a.steady_steps, "instances_R_mean": r_mean, "instances_binned_mean": binned_mean[0],
                  "roofline_step": step_roofline(cfg["P"], binned_mean[0] or r_mean, r_mean, npix, dts / a.steady_steps, a.lambda_dssim)}
    for i in range(a.warmup):
        one(nxt + i)
    nxt += a.warmup
    dt, loss, r_mean = timed(nxt, a.steps)
    nxt += a.steps
    assert torch.isfinite(loss).all(), "loss is not finite"
    headline_binned, headline_q = binned_mean[0], dict(q_mean)
    # the live roofline's launch durations: a leg of its own, straight after the headline region (same device state), every launch of
    # the kernel bracketed by HIP events on its launch stream -- one full cycle over the cameras, so the sample is the workload's mix
    n_roof = len(cams)
    dt_roof, loss_r, r_mean_roof = timed(nxt, n_roof, profile_kernel=a.roofline_kernel)
    nxt += n_roof
    roof_binned = binned_mean[0]
    assert torch.isfinite(loss_r).all(), "loss is not finite (roofline leg)"
    binned_mean[0] = headline_binned
    out = {
        "metric": metric_name(), "value": a.steps * scale / dt,
        "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak" if a.shard == "camera" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["name"], "gaussians": cfg["P"], "frames": cfg["F"], "width": cfg["W"],
                   "height": cfg["H"], "instances_R": r_mean,
                   "instances_R_is": "the reference's count (every tile of every splat's rectangle), mean over the cameras of the timed steps",
                   "instances_binned": binned_mean[0],
                   "instances_binned_is": "what the default binning keeps: instances that can reach alpha >= 1/255 in their tile",
                   "pairs_Q": {"processed": headline_q[False], "reference_lists": headline_q[True], "what": Q_IS},
                   "sh_degree": 3, "step_path": a.path, "batch_size": 1,
                   "lambda_dssim": a.lambda_dssim, "parallelism": ((f"camera-batch x{world}" + (" sharded-adam" if getattr(trainer.dist, "shard_adam", False) else "")
                                    if a.shard == "camera" else f"tile-row x{world}")
                                   + (" rccl-direct" if getattr(trainer.dist, "direct", None) is not None else " torch.distributed")) if (world > 1 or force_dist) else "single",
                   "ranks_seen": ranks_seen, "host_sync": a.sync_mode, "final_loss": float(loss), "densify_in_window": bool(a.with_densify),
                   "gaussians_at_end": int(g.get_xyz.shape[0]), "steps_replayed_after_overflow": int(trainer.replayed),
                   "inputs": "camera matrices and ground-truth images pre-staged in HBM before the timed region"},
        "roofline_step": step_roofline(cfg["P"], binned_mean[0] or r_mean, r_mean, npix, dt / a.steps, a.lambda_dssim),
    }
    if steady is not None:
        out["steady"] = steady
    if rank == 0:
        traffic, tnote = measured_traffic(a.roofline_kernel, cfg)
        sq, sqnote = sq_counters(a.roofline_kernel, cfg)
        out["roofline"] = prof.roofline(a.roofline_kernel, cfg["P"], roof_binned or r_mean_roof, npix, traffic=traffic, R_ref=r_mean_roof, sq=sq)
        if out["roofline"] is not None:
            rl = out["roofline"]
            rl.update(tnote)
            rl.update(sqnote)
            rl["measured_in"] = (f"a leg of its own after the headline region: {n_roof} steps (one cycle over the cameras), HIP events "
                                 f"around every {a.roofline_kernel} launch ({n_roof / dt_roof:.1f} steps/s with the events' bubbles); "
                                 "the headline region carries no events")
            # SURVEY 8d's secondary ceiling of the raster loops: FLOP on the evaluated pairs against the fp32 vector peak
            qk = {"render_bwd": ("q_bwd", 70.0), "render_fwd": ("q_fwd", 20.0)}.get(a.roofline_kernel)
            qrow = mean_q([q_of_cam[False][id(cam_of(nxt - n_roof + i))] for i in range(n_roof) if id(cam_of(nxt - n_roof + i)) in q_of_cam[False]])
            if qk and qrow:
                sec = rl["avg_launch_us"] * 1e-6
                flop = qk[1] * qrow[qk[0]]
                rl["pairs_Q"] = qrow
                rl["useful_flop_frac"] = flop / sec / 1e12 / prof.FP32_VALU_PEAK_TFLOPS
                rl["useful_flop_is"] = (f"{qk[1]:.0f} FLOP x Q ({qk[0]}: the pairs this kernel's tiles walk) / launch duration / "
                                        f"{prof.FP32_VALU_PEAK_TFLOPS:.0f} TFLOP/s (SURVEY 8d); most walked pairs fail the alpha test and do no "
                                        "arithmetic beyond it, so this prices the walk, not useful lanes")
                rl["alpha_pair_flop_frac"] = qk[1] * qrow["alpha_pairs"] / sec / 1e12 / prof.FP32_VALU_PEAK_TFLOPS
        prof.enable(a.roofline_kernel, False)
        if world == 1 and not a.no_extra:
            # the metric's two other readings, on the same scene and model state: the SSIM/L1 loss of the
            # north star, and render FPS.  Both after the headline region, so they cannot disturb it.
            k2 = max(1, min(a.steps, 50))
            op.lambda_dssim = 0.2
            for i in range(5):
                one(nxt + i)
            dt2, loss2, _ = timed(nxt + 5, k2)
            nxt += 5 + k2
            op.lambda_dssim = a.lambda_dssim
            assert torch.isfinite(loss2).all(), "loss (lambda_dssim 0.2) is not finite"
            out["with_ssim"] = {"lambda_dssim": 0.2, "value": k2 / dt2, "unit": "steps/s", "steps": k2,
                                "ms_per_step": 1e3 * dt2 / k2, "final_loss": float(loss2)}
            out["render_fps"] = render_fps(scene, g, trainer.pipe, trainer.background, trainer.delta_scale)
            # the path the north star names -- gaussian_renderer.render() + a torch loss + loss.backward() + optimizer.step(), as
            # train_4DGS.py:189-297 drives the modules (render() is one autograd node, fused_autograd.py; async binning with the
            # overflow replay of Trainer.step) -- on the same workload; and the other single-GPU configurations of BASELINE.json
            # (parity-test sizes, not bench lines), so that BASELINE.md's table is filled from this record
            if a.config == "c2" and not a.no_side_legs:
                out["via_render_api"] = side_leg(cfg, dev, "autograd", 100, 20)
                # the drop-in's default sync mode: what the reference's unchanged train_4DGS.py gets (it cannot select async)
                out["via_render_api_exact"] = side_leg(cfg, dev, "autograd", 100, 20, sync_mode="exact")
                # the reference-identical binning (every tile of every rectangle): the mode whose integer indices are bit-exact
                out["keep_all_tiles"] = side_leg(cfg, dev, "fused", 100, 20, keep_all_tiles=True)
                out["other_configs"] = {k: side_leg(CONFIGS[k], dev, "fused", 20, 5) for k in ("c1", "c3")}
                # BASELINE configs[4]: "densify/prune every 100 iters" -- the round at iteration 5100 is inside the window
                # first as a process's FIRST round finds the allocator (the leg before it ended with empty_cache(), no prewarm): the
                # boundary then includes whatever the driver takes to hand out ~3 GB of fresh memory on this box; then the leg as
                # every later round finds it (prewarmed) -- in this order, so that the cold leg IS cold
                cold = side_leg(CONFIGS["c5"], dev, "fused", 121, 10, with_densify=True, prewarm_allocator=False)
                out["other_configs"]["c5"] = side_leg(CONFIGS["c5"], dev, "fused", 121, 10, with_densify=True)
                out["other_configs"]["c5_cold_allocator"] = cold
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        sys.stdout.flush()
        full = json.dumps(out)
        print("BENCH_FULL " + full, file=sys.stderr)
        if os.environ.get("MOM_BENCH_FULL"):
            with open(os.environ["MOM_BENCH_FULL"], "w") as fh:
                fh.write(full + "\n")
        line = json.dumps(compact_line(out), separators=(",", ":"))
        assert len(line) < 4096, f"the bench line grew to {len(line)} bytes: the driver's tail would cut it"
        os.write(json_fd, (line + "\n").encode())
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
