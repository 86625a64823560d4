"""Fixed cost of the MLP kernels: forward / backward time against the number of Gaussians (torch events, 50 launches each)."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
N = importlib.import_module("iclr2025_3d-mom_amd._native")
ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
lib, s = N.lib(), N.current_stream()
g = torch.Generator().manual_seed(0)
mk = lambda *sh: (torch.randn(*sh, generator=g) * 0.3).cuda()
params = [mk(64, 64), mk(64)]
for nout in (3, 3, 4):
    params += [mk(64, 64), mk(64), mk(nout, 64), mk(nout)]
grads = [torch.zeros_like(p) for p in params]
d = ops.DeformMLPFunction._desc(params, grads)
for P in [int(x) for x in os.environ.get("PS", "32,1024,5000,8192,50000,200000").split(",")]:
    feat, xyz, scal, rot, flow = mk(P, 64), mk(P, 3), mk(P, 3), mk(P, 4), mk(P, 3)
    pts, sc, rt, a0, dfeat = (torch.empty(P, k, device="cuda") for k in (3, 3, 4, 64, 64))
    scratch = torch.empty(lib.mom_deform_backward_scratch_bytes(P), dtype=torch.uint8, device="cuda")
    fwd = lambda: lib.mom_deform_forward(C.byref(d), P, feat.data_ptr(), xyz.data_ptr(), scal.data_ptr(), rot.data_ptr(), flow.data_ptr(), 0.7,
                                         pts.data_ptr(), sc.data_ptr(), rt.data_ptr(), a0.data_ptr(), s)
    bwd = lambda: lib.mom_deform_backward(C.byref(d), P, feat.data_ptr(), a0.data_ptr(), pts.data_ptr(), sc.data_ptr(), rt.data_ptr(),
                                          dfeat.data_ptr(), scratch.data_ptr(), s)
    out = []
    for f in (fwd, bwd):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"P {P:7d}  fwd {out[0]:7.1f} us   bwd (dx + dw) {out[1]:7.1f} us")
