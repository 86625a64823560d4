"""Host-side checks of bench.py's contract (no GPU): defaults, the metric string, the workload names, the camera walk, and the
pinned-count wait of the drop-in's exact mode."""
import importlib
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_and_metric_are_the_contracts():
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.config, a.shard, a.path) == (1, 100, 20, "c2", "camera", "fused")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert bench.metric_name() == base["metric"]
    # config.workload of the N = 1 line is BASELINE configs[1]; the parity-size legs name configs[0], [2], [4]
    for key, idx, words in (("c2", 1, ("200k", "60 frames", "960")), ("c3", 2, ("1M", "120 frames", "1920")),
                            ("c1", 0, ("5k", "8 frames", "256")), ("c5", 4, ("4M", "240 frames", "1920"))):
        name, theirs = bench.CONFIGS[key]["name"], base["configs"][idx]
        for w in words:
            assert w in name and w in theirs, (key, w, name, theirs)


def test_the_camera_walk_covers_every_camera_of_every_config_evenly():
    """cam_of() walks the cameras with stride 17: a permutation of the F + 5 cameras of every config, so K consecutive steps sample
    the set evenly (a 20-step window in list order sat on the hemisphere views)."""
    for cfg in bench.CONFIGS.values():
        n = cfg["F"] + 5
        assert math.gcd(bench.CAMERA_STRIDE, n) == 1
        assert sorted((bench.CAMERA_STRIDE * i) % n for i in range(n)) == list(range(n))
        window = sorted((bench.CAMERA_STRIDE * i) % n for i in range(300, 320))
        assert len(set(window)) == min(20, n)
        if n >= 60:           # spread over the whole list, not a run of neighbours
            assert window[-1] - window[0] > n // 2


def test_wait_count_polls_the_pinned_word_and_falls_back_to_a_wait():
    RC = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization._C")
    w = torch.zeros(1, dtype=torch.int32)
    w[0] = 1234
    assert RC.wait_count(w) == 1234                      # already there: no wait of any kind

    class Ev:
        def __init__(self):
            self.waited = 0

        def synchronize(self):
            self.waited += 1
            w[0] = 77                                    # "the stage finished while we were parked"
    w[0] = RC.COUNT_PENDING
    ev = Ev()
    assert RC.wait_count(w, ev, spin_s=0.001) == 77 and ev.waited == 1


def test_wait_count_raises_when_nobody_ever_writes_the_word():
    """A pinned word no kernel writes (the geometry stage's launch failed): after the stream wait the call must raise, not hand
    COUNT_PENDING (-1) to mom_raster_binning_bytes."""
    import pytest
    RC = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization._C")
    N = importlib.import_module("iclr2025_3d-mom_amd._native")
    w = torch.full((1,), RC.COUNT_PENDING, dtype=torch.int32)

    class Ev:
        def synchronize(self):
            pass                                         # the stream drains; the word stays as the host left it
    with pytest.raises(N.MomError, match="instance count"):
        RC.wait_count(w, Ev(), spin_s=0.001)


def test_the_compact_line_carries_every_leg_as_numbers_and_fits_the_drivers_tail():
    """bench.py prints ONE line < 4 KB; every side leg sits inside `config.legs` (the driver keeps `config` whole) as numbers only;
    the prose is behind --explain.  Input: the full record of a real run (profiles/*_bench_full.json)."""
    import glob
    full = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_full.json")))[-1]))
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    legs = line["config"]["legs"]
    for k in ("steady", "with_ssim", "keep_all_tiles", "via_render_api", "via_render_api_exact", "c1", "c3", "c5", "render_fps"):
        assert k in legs, k
    for k in ("two_streams", "one_stream", "one_stream_exact", "as_scripted", "blocking"):
        assert isinstance(legs["render_fps"][k], float)
    assert legs["c5"]["gaussians_after"] < legs["c5"]["gaussians_before"] and legs["c5"]["boundary_ms"] > 0

    def only_numbers(d):
        for v in d.values():
            if isinstance(v, dict):
                only_numbers(v)
            else:
                assert isinstance(v, (int, float, bool)), v
    only_numbers(legs)
    rl = line["roofline"]
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3 and rl["step"]["frac"] < 1
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert set(bench.EXPLAIN) >= {"config.legs", "roofline", "cpu_baseline"}
