"""f3: how much of frame k's per-tile depth order survives into frame k + 1 along the reference's 59-pose `side`
render trajectory (render_4DGS.py:60-71), i.e. what a cross-frame reuse of the sort could save.

Renders the trajectory with the no-grad path at config 2 (exact sizing), reads every frame's tile ranges and sorted lists back,
and for each pair of consecutive frames reports, over all tiles: the share of frame k + 1's entries that were in the same tile
in frame k (survivors) and of newcomers; the inversions among the survivors when they keep frame k's order but are keyed with
frame k + 1's depths (Kendall distance, as a share of the pairs, and the largest displacement = passes an odd-even / insertion
fix-up would need); and the share of tiles whose list is unchanged.  Prints one JSON document.
    python tools/sort_reuse_stats.py profiles/r03_sort_reuse_stats.json"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

N = importlib.import_module("iclr2025_3d-mom_amd._native")
R = importlib.import_module("iclr2025_3d-mom_amd.gaussian_renderer")
DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")


def lists_of(fr, P, W, H):
    import ctypes as C
    lay = N.MomRasterLayout()
    N.lib().mom_raster_layout(P, W, H, fr.cap, C.byref(lay))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    im = fr.img[(-fr.img.data_ptr()) % 256:].cpu().numpy()
    ranges = im[lay.img_ranges:lay.img_ranges + tiles * 8].view(np.uint32).reshape(tiles, 2).astype(np.int64)
    b = fr.binning[(-fr.binning.data_ptr()) % 256:].cpu().numpy()
    n = int(ranges[:, 1].max(initial=0))
    pl = b[lay.bin_point_list:lay.bin_point_list + n * 4].view(np.uint32).astype(np.int64)
    rec = fr.geom[(-fr.geom.data_ptr()) % 256:].cpu().numpy()
    depth = rec[lay.geom_rec:lay.geom_rec + P * 48].view(np.float32).reshape(P, 12)[:, 2].copy()
    return ranges, pl, depth


def main():
    cfg = bench.CONFIGS["c2"]
    dev = torch.device("cuda")
    scene, g, trainer, op = bench.build_state(cfg, dev, fused=True)
    for i in range(30):                                   # a few training steps: not the untouched initial state
        trainer.step(5001 + i, cams=[trainer.cams[i % len(trainer.cams)]])
    trainer.drain()
    cams = scene.getVideoCameras_side()
    DGR.set_sync_mode("exact")
    frames = []
    with torch.no_grad():
        for c in cams:
            R.render(c, g, trainer.pipe, trainer.background, stage="fine", cam_type=scene.dataset_type, delta_scale=trainer.delta_scale)
            torch.cuda.synchronize()
            frames.append(lists_of(g._fused_render, cfg["P"], cfg["W"], cfg["H"]))
    pairs = []
    for (r0, p0, _), (r1, p1, d1) in zip(frames, frames[1:]):
        surv = new = tot = inv = npairs = same_tiles = 0
        maxdisp = 0
        for t in range(r0.shape[0]):
            a, b = p0[r0[t, 0]:r0[t, 1]], p1[r1[t, 0]:r1[t, 1]]
            tot += len(b)
            if len(a) == len(b) and np.array_equal(a, b):
                same_tiles += 1
            keep = a[np.isin(a, b)]                          # frame k's order, restricted to what is still in the tile
            surv += len(keep)
            new += len(b) - len(keep)
            if len(keep) > 1:
                # position of each survivor in frame k+1's (sorted) list -> inversions of that permutation
                pos = {int(x): i for i, x in enumerate(b)}
                perm = np.array([pos[int(x)] for x in keep])
                rank = np.argsort(np.argsort(perm))
                maxdisp = max(maxdisp, int(np.abs(rank - np.arange(len(rank))).max()))
                # Kendall distance by merge-count is overkill here: lists are short, do it in O(n^2) blocks with numpy
                m = len(rank)
                if m <= 2048:
                    inv += int((rank[:, None] > rank[None, :])[np.triu_indices(m, 1)].sum())
                    npairs += m * (m - 1) // 2
        pairs.append({"entries": tot, "survivor_share": surv / max(tot, 1), "newcomer_share": new / max(tot, 1),
                      "inverted_pair_share": inv / max(npairs, 1), "max_displacement": maxdisp,
                      "unchanged_tile_share": same_tiles / r0.shape[0]})
    mean = lambda k: float(np.mean([p[k] for p in pairs]))
    doc = {"workload": cfg["name"], "trajectory": "side, 59 poses", "lib_version": N.lib().mom_version().decode(),
           "mean": {k: mean(k) for k in ("entries", "survivor_share", "newcomer_share", "inverted_pair_share", "unchanged_tile_share")},
           "max_displacement_over_all_tiles_and_frames": max(p["max_displacement"] for p in pairs), "per_frame_pair": pairs}
    with open(sys.argv[1] if len(sys.argv) > 1 else "sort_reuse_stats.json", "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: doc[k] for k in ("mean", "max_displacement_over_all_tiles_and_frames")}), file=sys.stderr)


if __name__ == "__main__":
    main()
