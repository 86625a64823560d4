"""Run-to-run spread of the 100-iteration loss-curve replay (tests/test_whole_step_gpu.py) against the fixture: the float atomics'
order differs from run to run and Adam (eps 1e-15) amplifies sign flips of vanishing gradients.  python tools/probe/curve_spread.py [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import io, contextlib
import numpy as np
from oracle.make_curve_fixture import run, N_COARSE
d = np.load(os.path.join(ROOT, "tests", "golden", "g10_loss_curve.npz"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worst = {}
for i in range(n):
    with contextlib.redirect_stdout(io.StringIO()):
        losses, points, cs, xyz = run("cuda", fused_fine=True)
    rel = np.abs(losses - d["losses"]) / np.abs(d["losses"])
    row = {"loss": float(rel.max()), "xyz": float(np.abs(xyz - d["xyz_sample"]).max()), "points_equal": bool((points == d["points"]).all())}
    for k in cs:
        if k.startswith("sum_"):
            row[k] = abs(cs[k] - float(d[k])) / max(float(d["abs_" + k[4:]]), 1e-12)
    for k, v in row.items():
        if k != "points_equal":
            worst.setdefault(k, []).append(v)
    if not row["points_equal"]:
        print("run", i, "POINT COUNT DIFFERS")
for k, v in worst.items():
    v = np.array(v)
    print("%-16s median %.2e  max %.2e" % (k, np.median(v), v.max()))
