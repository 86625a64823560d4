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
