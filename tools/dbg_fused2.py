import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")
sa, ga, ta, op = bench.build_state(cfg, torch.device("cuda"), fused=False)
sf, gf, tf, op = bench.build_state(cfg, torch.device("cuda"), fused=True)
for it in range(3):
    ci = (3 * it + 1) % len(ta.cams)
    la = float(ta.step(5001 + it, cams=[ta.cams[ci]])); lf = float(tf.step(5001 + it, cams=[tf.cams[ci]]))
    torch.cuda.synchronize()
    print("it", it, "cam", ci, la, lf, "R", int(tf.fused.nr_host[0]), "cap", tf.fused.cap, "status", int(tf.fused.status_host[0]))
    na = dict(ga._deformation.named_parameters()); nf = dict(gf._deformation.named_parameters())
    worst = max(((float((na[k] - nf[k]).abs().max()), k) for k in na), key=lambda x: x[0])
    print("   deform worst", worst)
    for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
        print("   ", k, float((getattr(ga, k) - getattr(gf, k)).abs().max()))
    print("   lr", [round(g["lr"], 9) for g in ga.optimizer.param_groups], [round(g["lr"], 9) for g in gf.optimizer.param_groups])
