"""CPU simulation of the row flushes of hexplane_bwd4_kernel on bench.py's scene: how many 128-byte gradient rows does the
run-length aggregation flush, per level and plane, for a given processing order and pending-row policy?

    python tools/sim_hexplane_runs.py [--config c2]

Policies: "slot" = the kernel's (corner k of a plane has one pending row; flushed when the row of corner k changes);
"assoc4" = the four pending rows of a plane form one set (a new row only evicts a row that no corner of the new point uses).
Orders: morton3d (the kernel's), xy-major, and the identity.
"""
import argparse
import importlib
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def part(x, stride):
    out = np.zeros_like(x, dtype=np.uint64)
    for b in range(11):
        out |= ((x >> b) & 1).astype(np.uint64) << np.uint64(stride * b)
    return out


def orders(xyz):
    lo = np.minimum(xyz.min(0), 0.0)
    hi = np.maximum(xyz.max(0), 0.0)            # the kernel's box is seeded with the origin
    q = ((xyz - lo) / (hi - lo) * 1023).astype(np.uint64)
    m3 = part(q[:, 0], 3) | (part(q[:, 1], 3) << np.uint64(1)) | (part(q[:, 2], 3) << np.uint64(2))
    lo2, hi2 = xyz.min(0), xyz.max(0)
    q2 = ((xyz - lo2) / (hi2 - lo2) * 2047).astype(np.uint64)
    xy = ((part(q2[:, 0], 2) | (part(q2[:, 1], 2) << np.uint64(1))) << np.uint64(10)) | (q2[:, 2] >> np.uint64(1))
    return {"morton3d": np.argsort(m3, kind="stable"), "xy-major": np.argsort(xy, kind="stable"), "identity": np.arange(len(xyz))}


def flushes(rows, policy):
    """rows: [n, 4] row ids of one plane for the points one half-wave walks, in order (-1 = no row)."""
    if policy == "slot":
        n = 0
        for k in range(4):
            r = rows[:, k]
            r = r[r >= 0]
            n += int((np.diff(r) != 0).sum()) + (1 if len(r) else 0)
        return n
    if policy == "parity":
        # slot = (parity of the row's y, parity of its x): the four corners of a texel always take four different slots, and
        # a row keeps its slot when the walk moves to a neighbouring texel.  rows[:, k] is corner k = 2*dy + dx of texel
        # (y0, x0); its slot is k ^ s with s = 2*(y0 & 1) + (x0 & 1).  `xy` carries (x0, y0) per point.
        raise RuntimeError("parity needs x0/y0: use flushes_parity")
    pend, n = set(), 0
    for rr in rows:
        new = set(int(x) for x in rr if x >= 0)
        keep = pend & new
        need = new - keep
        free = 4 - len(keep)
        evict = list(pend - keep)
        # rows not used by this point can stay if there is room
        stay = evict[:max(0, free - len(need))]
        n += len(evict) - len(stay)
        pend = keep | need | set(stay)
    return n + len(pend)


def flushes_parity(rows, x0, y0):
    sidx = 2 * (y0 & 1) + (x0 & 1)
    n = 0
    for j in range(4):                       # static slot j holds corner k = j ^ s of each point
        k = j ^ sidx
        r = rows[np.arange(len(rows)), k]
        r = r[r >= 0]
        n += int((np.diff(r) != 0).sum()) + (1 if len(r) else 0)
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--blocks", type=int, default=768)
    a = ap.parse_args()
    cfg = bench.CONFIGS[a.config]
    # only the scene's point cloud and its bounding box are needed: no model, no backend
    S = importlib.import_module("iclr2025_3d-mom_amd.scene")
    scene = S.SyntheticScene(cfg["P"], cfg["F"], cfg["W"], cfg["H"], seed=6666)
    xyz = np.asarray(scene.point_cloud.points, dtype=np.float64)
    aabb = np.array([scene.xyz_max, scene.xyz_min], dtype=np.float64)     # HexPlaneField.set_aabb(xyz_max, xyz_min)
    P = len(xyz)
    nchunks = (P + 63) // 64
    waves = a.blocks * 4
    cpw = (nchunks + waves - 1) // waves
    print(f"P={P}, {nchunks} chunks of 64, {waves} waves per level, {cpw} chunk(s) per wave")
    c = (xyz - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0
    for name, order in orders(xyz).items():
        total = {"slot": 0, "assoc4": 0, "parity": 0}
        contrib = 0
        for lvl, res in enumerate((64, 128)):
            ix = np.clip((c + 1) / 2 * (res - 1), 0, res - 1)
            i0 = np.floor(ix).astype(np.int64)
            for (pa, pb) in ((0, 1), (0, 2), (1, 2)):
                x0, y0 = i0[:, pa], i0[:, pb]
                hx, hy = x0 + 1 < res, y0 + 1 < res
                rid = np.stack([y0 * res + x0, np.where(hx, y0 * res + x0 + 1, -1), np.where(hy, (y0 + 1) * res + x0, -1),
                                np.where(hx & hy, (y0 + 1) * res + x0 + 1, -1)], 1)[order]
                contrib += int((rid >= 0).sum())
                for w in range(min(waves, (nchunks + cpw - 1) // cpw)):
                    pts = np.arange(w * cpw * 64, min(P, (w + 1) * cpw * 64))
                    for h in range(2):
                        sel = pts[(pts % 64) // 32 == h]
                        if len(sel) == 0:
                            continue
                        for pol in ("slot", "assoc4"):
                            total[pol] += flushes(rid[sel], pol)
                        total["parity"] += flushes_parity(rid[sel], x0[order][sel], y0[order][sel])
        print(f"{name:10s} contributions {contrib:9d} | flushed rows: slot {total['slot']:9d} (run {contrib / total['slot']:.2f})"
              f" | assoc4 {total['assoc4']:9d} (run {contrib / total['assoc4']:.2f})"
              f" | parity {total['parity']:9d} (run {contrib / total['parity']:.2f})")


if __name__ == "__main__":
    main()
