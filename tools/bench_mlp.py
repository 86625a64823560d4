"""Micro-benchmark of the fused deformation MLP kernels (library HIP-event slots), P = 200k."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
prof = importlib.import_module("iclr2025_3d-mom_amd.profiling")
P = 200_000
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) * 0.3).cuda()
for mode in ("ones", "zeros", "small"):
    params = [mk(64, 64), mk(64)]
    for nout in (3, 3, 4):
        params += [mk(64, 64), mk(64), mk(nout, 64), mk(nout)]
    params = [p.requires_grad_(True) for p in params]
    feat = (mk(P, 64) * 3).requires_grad_(True)
    xyz, scal, rot, flow = mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 3)
    prof.enable("mlp_fwd"); prof.enable("mlp_bwd")
    for it in range(6):
        o = ops.deform_mlp(feat, xyz, scal, rot, flow, 0.7, params)
        go = [torch.ones_like(t) if mode == "ones" else (torch.zeros_like(t) if mode == "zeros" else torch.randn_like(t) * 1e-4) for t in o]
        torch.autograd.backward(o, go)
        if it == 0:
            torch.cuda.synchronize(); prof.read("mlp_fwd"); prof.read("mlp_bwd")
    torch.cuda.synchronize()
    f, nf = prof.read("mlp_fwd"); b, nb = prof.read("mlp_bwd")
    print(f"{os.path.basename(os.environ.get('MOM4D_LIB','default'))} [{mode}]: fwd {f/nf*1e3:.0f} us  bwd {b/nb*1e3:.0f} us", flush=True)
