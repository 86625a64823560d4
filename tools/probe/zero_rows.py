"""per config and camera, the fraction of Gaussians whose compositing-backward record (gacc) is exactly zero after
render_bwd -- culled, fully occluded or binned away.  Such a row gives exactly zero gradients to the projection backward and
therefore contributes nothing to the MLP backward, the HexPlane gather and the scatter.  Also: radii == 0 (not visible at all)."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

out = {}
for name in sys.argv[1:] or ["c2", "c3", "c5"]:
    cfg = bench.CONFIGS[name]
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
    fs = trainer.fused
    cams = trainer.cams
    rows = []
    for i in range(12):
        cam = cams[(17 * i) % len(cams)]
        trainer.step(5001 + i, cams=[cam])
        trainer.drain()
        torch.cuda.synchronize()
        P = g._xyz.shape[0]
        W, H = fs._wh
        rec = fs._gacc_view(P, W, H).view(P, -1)
        zero = (rec == 0).all(dim=1)
        invisible = fs.radii[:P] <= 0
        # what the deformation backward actually receives: d pts (in gxyz before the HexPlane adds), d scales, d rotations
        rows.append({"camera": (17 * i) % len(cams), "zero_record": float(zero.float().mean()), "invisible": float(invisible.float().mean()),
                     "visible_but_zero": float((zero & ~invisible).float().mean())})
    m = lambda k: sum(r[k] for r in rows) / len(rows)
    out[name] = {"workload": cfg["name"], "mean_zero_record": m("zero_record"), "mean_invisible": m("invisible"),
                 "mean_visible_but_zero": m("visible_but_zero"), "min_zero_record": min(r["zero_record"] for r in rows),
                 "max_zero_record": max(r["zero_record"] for r in rows), "per_camera": rows}
    print(name, json.dumps({k: v for k, v in out[name].items() if k != "per_camera"}), flush=True)
    del scene, g, trainer, fs
    torch.cuda.empty_cache()
json.dump(out, open(os.environ.get("ZERO_ROWS_OUT", "gpurun_out/r06_zero_rows.json"), "w"), indent=1)
