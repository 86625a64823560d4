"""A/B check of the compositing kernels between two builds of libmom4d.so.

    MOM4D_LIB=/path/to/libA.so python tools/ab_render.py gpurun_out/ab_A.npz
    MOM4D_LIB=/path/to/libB.so python tools/ab_render.py gpurun_out/ab_B.npz
    python tools/ab_render.py --compare gpurun_out/ab_A.npz gpurun_out/ab_B.npz

Dumps the forward images / final_T / n_contrib and the backward gradients of a few seeded scenes.  --compare asserts that
n_contrib is identical (the set of composited splats did not move) and the forward floats agree to 2e-6 (a change that
makes the compiler contract an expression differently moves last bits); --compare-bitwise demands bit-identical forward
arrays, for changes that must not touch the arithmetic at all.  Gradients are accumulated with unordered float atomics,
so for those the largest difference relative to the array's maximum is printed and bounded by 1e-4.
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def dump(path):
    import hip_helpers as hh
    import scenes
    out = {}
    for tag, kw in (("small", dict(P=3000, seed=1, W=128, H=96)), ("dense", dict(P=40000, seed=2, W=320, H=240, scale=(-3.5, -1.5))),
                    ("ragged", dict(P=9000, seed=3, W=203, H=117))):
        s = scenes.random_gaussians(**kw)
        fw = hh.hip_forward(s)
        rng = np.random.default_rng(7)
        dcol = rng.standard_normal(fw["color"].shape).astype(np.float32)
        ddep = rng.standard_normal(fw["depth"].shape).astype(np.float32)
        bw = hh.hip_backward(fw, dcol, ddep)
        for k in ("color", "depth", "final_T", "n_contrib"):
            out[f"{tag}/fwd/{k}"] = np.asarray(fw[k])
        for k, v in bw.items():
            if isinstance(v, np.ndarray):
                out[f"{tag}/bwd/{k}"] = v
    np.savez(path, **out)
    print("wrote", path, len(out), "arrays")


def compare(a, b, bitwise=False):
    A, B = np.load(a), np.load(b)
    assert sorted(A.files) == sorted(B.files), "different array sets"
    worst = 0.0
    for k in sorted(A.files):
        x, y = A[k], B[k]
        if "/fwd/" in k:
            same = x.tobytes() == y.tobytes()
            if x.dtype.kind != "f":
                print(f"{k:28s} identical: {same}")
                assert same, k                       # n_contrib: the set of composited splats must not move
            else:
                ne = x != y
                ulp = np.abs(x.view(np.int32).astype(np.int64) - y.view(np.int32).astype(np.int64))
                worst_abs = float(np.abs(x.astype(np.float64) - y).max())
                print(f"{k:28s} bit-identical: {same}   differing {int(ne.sum())}/{x.size}   max abs {worst_abs:.2e}   "
                      f"max ulp {int(ulp.max())}")
                assert same or (not bitwise and worst_abs <= 2e-6), k
        else:
            den = max(1e-30, float(np.abs(x).max()))
            rel = float(np.abs(x.astype(np.float64) - y).max()) / den
            worst = max(worst, rel)
            print(f"{k:28s} max |diff| / max |x| = {rel:.2e}")
    print("worst gradient difference (relative to the array's max):", f"{worst:.2e}")
    assert worst < 1e-4


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] in ("--compare", "--compare-bitwise"):
        compare(sys.argv[2], sys.argv[3], bitwise=sys.argv[1] == "--compare-bitwise")
    elif len(sys.argv) == 2:
        dump(sys.argv[1])
    else:
        sys.exit(__doc__)
