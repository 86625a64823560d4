"""How many (footprint, splat) pairs the compositing kernels have to walk for different wave footprints, on one camera of
config c2: a pair counts when any pixel of the footprint reaches alpha >= 1/255 (saturation ignored).  Decides whether finer
footprints (lists per half-wave / per 16-lane row) would pay."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, ctypes as C
import bench
N = importlib.import_module("iclr2025_3d-mom_amd._native")
scene, g, trainer, op = bench.build_state(bench.CONFIGS["c2"], torch.device("cuda"), fused=True)
fs = trainer.fused
cam = trainer.cams[7]
fs.exact_next(); fs.forward_backward(cam, 1); torch.cuda.synchronize()
P, W, H = 200000, 960, 540
lay = N.MomRasterLayout(); N.lib().mom_raster_layout(P, W, H, fs.cap, C.byref(lay))
geom = fs.geom[(-fs.geom.data_ptr()) % 256:].cpu().numpy()
rec = geom[lay.geom_rec:lay.geom_rec + P * 48].view(np.float32).reshape(P, 12)
img = fs.img[(-fs.img.data_ptr()) % 256:].cpu().numpy()
tiles = 60 * 34
ranges = img[lay.img_ranges:lay.img_ranges + tiles * 8].view(np.uint32).reshape(tiles, 2)
b = fs.binning[(-fs.binning.data_ptr()) % 256:].cpu().numpy()
R = int(fs.nr_host[0])
pl = b[lay.bin_point_list:lay.bin_point_list + R * 4].view(np.uint32)
print("R", R)
ncontrib = img[lay.img_n_contrib:lay.img_n_contrib + W * H * 4].view(np.uint32).reshape(H, W)
tot = {k: 0 for k in ("inst", "px_valid", "s16x4", "s8x4", "s8x8", "s4x4")}
# the backward's executed (wave, splat) pairs -- a 16x4 strip with at least one lane that passes every test, occlusion included
# (position in the list < the pixel's last contributor) -- by the number of such lanes
lane_hist = np.zeros(65, np.int64)
quarter = {}
rng = np.random.default_rng(0)
for t in rng.choice(tiles, 200, replace=False):
    a, e = ranges[t]
    if e <= a: continue
    ids = pl[a:e]
    r = rec[ids]
    tx, ty = t % 60, t // 60
    px = (tx * 16 + np.arange(16))[None, None, :].astype(np.float32); py = (ty * 16 + np.arange(16))[None, :, None].astype(np.float32)
    dx = r[:, 0][:, None, None] - px; dy = r[:, 1][:, None, None] - py
    power = -0.5 * (r[:, 4][:, None, None] * dx * dx + r[:, 6][:, None, None] * dy * dy) - r[:, 5][:, None, None] * dx * dy
    alpha = np.minimum(0.99, r[:, 7][:, None, None] * np.exp(power))
    v = (power <= 0) & (alpha >= 1 / 255.0)                      # [n,16(y),16(x)]
    n = v.shape[0]
    tot["inst"] += n; tot["px_valid"] += int(v.sum())
    tot["s16x4"] += int(v.reshape(n, 4, 4, 16).any(axis=(2, 3)).sum())
    tot["s8x4"] += int(v.reshape(n, 4, 4, 2, 8).any(axis=(2, 4)).sum())
    tot["s8x8"] += int(v.reshape(n, 2, 8, 2, 8).any(axis=(2, 4)).sum())
    tot["s4x4"] += int(v.reshape(n, 4, 4, 4, 4).any(axis=(2, 4)).sum())
    inside = ((ty * 16 + np.arange(16))[:, None] < H) & ((tx * 16 + np.arange(16))[None, :] < W)
    nc = np.zeros((16, 16), np.int64)
    ys, xs = np.nonzero(inside)
    nc[ys, xs] = ncontrib[ty * 16 + ys, tx * 16 + xs]
    vb = v & (np.arange(n)[:, None, None] < nc[None])             # what render_bwd's `valid` is
    per_strip = vb.reshape(n, 4, 4, 16).sum(axis=(2, 3)).reshape(-1)
    lane_hist += np.bincount(per_strip, minlength=65)
    # two splats per wave iteration, one per half-wave (denser pairs): each half of a strip walks its own list;
    # an iteration serves one entry of each, so a strip needs max(|A|, |B|) iterations instead of |A u B|
    strips = vb.reshape(n, 4, 4, 16)                                  # [n, strip, row in strip, x]
    any_strip = strips.any(axis=(2, 3))                               # [n, 4]
    halves = {"16x2": (strips[:, :, :2, :].any(axis=(2, 3)), strips[:, :, 2:, :].any(axis=(2, 3))),
              "8x4": (strips[:, :, :, :8].any(axis=(2, 3)), strips[:, :, :, 8:].any(axis=(2, 3)))}
    tot.setdefault("it_union", 0)
    tot["it_union"] += int(any_strip.sum())
    for name, (A, B) in halves.items():
        tot.setdefault("it_" + name, 0)
        tot.setdefault("halfpairs_" + name, 0)
        tot["it_" + name] += int(np.maximum(A.sum(axis=0), B.sum(axis=0)).sum())
        tot["halfpairs_" + name] += int(A.sum() + B.sum())
    # quarter-wave lists: the rows of a wave are 16x1 pixel rows of a 16x4 strip (today's lane layout), 4x4 blocks side by side in a
    # 16x4 strip, or the 2x2 4x4 blocks of an 8x8 area.  A list holds the splats that can reach the row's pixels (v) in front of
    # the row's last contributor; today's list: the same test on the whole wave footprint
    pos = np.arange(n)[:, None, None]
    def lists(rows_of_wave):            # rows_of_wave: [waves][4] boolean pixel masks [16,16]
        out = dict(cur_list=0, cur_exec=0, it=0, rows=0, rows_exec=0)
        for rows in rows_of_wave:
            foot = np.any(rows, axis=0)
            wlast = nc[foot].max() if foot.any() else 0
            cur = v[:, foot].any(axis=1) & (np.arange(n) < wlast)
            out["cur_list"] += int(cur.sum())
            out["cur_exec"] += int(vb[:, foot].any(axis=1).sum())
            lens = []
            for m in rows:
                rlast = nc[m].max()
                L = v[:, m].any(axis=1) & (np.arange(n) < rlast)
                lens.append(int(L.sum()))
                out["rows"] += int(L.sum())
                out["rows_exec"] += int(vb[:, m].any(axis=1).sum())
            out["it"] += max(lens)
        return out
    yy, xx = np.mgrid[0:16, 0:16]
    layouts = {
        "16x1 rows of a 16x4 strip": [[(yy == 4 * w + r) for r in range(4)] for w in range(4)],
        "4x4 blocks of a 16x4 strip": [[(yy // 4 == w) & (xx // 4 == r) for r in range(4)] for w in range(4)],
        "4x4 blocks of an 8x8 area": [[(yy // 4 == 2 * (w // 2) + r // 2) & (xx // 4 == 2 * (w % 2) + r % 2) for r in range(4)] for w in range(4)],
    }
    for name, rw in layouts.items():
        d = lists(rw)
        q = quarter.setdefault(name, dict(cur_list=0, cur_exec=0, it=0, rows=0, rows_exec=0))
        for k2 in d: q[k2] += d[k2]
print(tot)
print("quarter-wave lists (one list per 16-lane row; a wave iteration serves one entry of each of its four rows):")
for name, d in quarter.items():
    print(f"  {name}: list entries walked now {d['cur_list']} (executed {d['cur_exec']}); iterations with row lists {d['it']} "
          f"({d['it'] / d['cur_list']:.3f} of the entries walked now, {d['it'] / d['cur_exec']:.3f} of the executed pairs); "
          f"(row, splat) list entries {d['rows']}, of them with a valid lane {d['rows_exec']} = {d['rows_exec'] / d['cur_exec']:.2f} atomic rows per executed pair now")
i = tot["inst"]
print("valid pixels per instance", tot["px_valid"] / i)
for k, lanes in (("s16x4", 64), ("s8x4", 32), ("s8x8", 64), ("s4x4", 16)):
    print(k, "footprint-pairs per instance %.2f" % (tot[k] / i), "lane-slots per instance %.1f" % (tot[k] / i * lanes),
          "useful %.3f" % (tot["px_valid"] / (tot[k] * lanes)))
executed = int(lane_hist[1:].sum())
print("render_bwd: executed (wave, splat) pairs per instance %.2f; valid lanes per executed pair: mean %.1f of 64" %
      (executed / i, float((lane_hist * np.arange(65)).sum()) / executed))
edges = [(1, 1), (2, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 64)]
print("  share of executed pairs by valid lanes: " + ", ".join(f"{a}-{b}: {lane_hist[a:b + 1].sum() / executed:.3f}" for a, b in edges))
for name in ("16x2", "8x4"):
    print(f"half-wave lists {name}: iterations {tot['it_' + name]} against {tot['it_union']} now ({tot['it_' + name] / tot['it_union']:.3f}); "
          f"(half, splat) pairs {tot['halfpairs_' + name]} = {tot['halfpairs_' + name] / tot['it_union']:.2f} per executed pair now (atomic rows)")
