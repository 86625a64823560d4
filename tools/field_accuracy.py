"""Error of the three deformation-field forward paths against an fp64 evaluation of the same field and MLP on the same inputs:
  two-kernel (hexplane_fwd4 + f32-MFMA MLP), one-kernel f32 (MOM_FIELD_MODE=f32), one-kernel bf16x3 (default).
Run each mode in its own process (the library reads MOM_FIELD_MODE once):  python tools/field_accuracy.py [out.json]
Prints / writes max and RMS errors of feat, a0, pts, scales, rots relative to the fp64 result's scale."""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(mode):
    import numpy as np
    import torch
    t = importlib.import_module("test_deform_field_gpu")
    tr = importlib.import_module("oracle.torch_ref")
    P, time = 50000, 0.37
    f = t._field((64, 64, 64, 25))
    params_cpu, mk = t._mlp(5)
    xyz = t._points(P)
    scal, rot, flow, opac = mk(P, 3), mk(P, 4), mk(P, 3), mk(P, 1)
    # fp64 reference: the reference's torch ops in double precision
    d = lambda x: x.double()
    feat64 = tr.hexplane_features(d(xyz), time, d(f.aabb.detach()), [[d(p.detach()).contiguous() for p in g] for g in f.grids])
    o64 = tr.deform_mlp(feat64, d(xyz), d(scal), d(rot), d(flow), 0.7, [d(p) for p in params_cpu])
    a0_64 = torch.relu(feat64 @ d(params_cpu[0]).t() + d(params_cpu[1]))
    fg = f.cuda()
    params = [p.cuda() for p in params_cpu]
    cu = [x.cuda() for x in (xyz, scal, rot, flow, opac)]
    out = t._run_forward(fg, params, P, *cu, time, None, fused=(mode != "two-kernel"))
    res = {}
    for k, r in (("feat", feat64), ("a0", a0_64), ("pts", o64[0]), ("sc_d", o64[1]), ("rot_d", o64[2])):
        e = (out[k].double().cpu() - r).abs()
        scale = float(r.abs().max())
        res[k] = {"max_rel": float(e.max()) / scale, "rms_rel": float((e ** 2).mean().sqrt()) / scale}
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    allres = {}
    for mode, env in (("two-kernel", {}), ("one-kernel-f32", {"MOM_FIELD_MODE": "f32"}), ("one-kernel-bf16x3", {})):
        e = dict(os.environ, **env)
        p = subprocess.run([sys.executable, __file__, "--child", mode], env=e, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(p.stdout[-2000:], p.stderr[-2000:])
            sys.exit(1)
        allres[mode] = json.loads(line[0][7:])
        print(mode, json.dumps(allres[mode]))
    if len(sys.argv) > 1:
        json.dump({"what": "forward error against fp64, 50 k points, relative to each output's largest magnitude", "paths": allres},
                  open(sys.argv[1], "w"), indent=1)
