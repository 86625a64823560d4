"""Phase stamps of the one-kernel MLP backward (csrc/deform_bwd_b3.hip built with -DB3F_STAMPS: MOM4D_LIB names that build):
s_memtime cycles per phase summed over a wave's tiles, median over the workgroups, for a head wave and the trunk wave."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
N = importlib.import_module("iclr2025_3d-mom_amd._native")
P = 200_000
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) * 0.3).cuda()
params = [mk(64, 64), mk(64)]
for nout in (3, 3, 4):
    params += [mk(64, 64), mk(64), mk(nout, 64), mk(nout)]
grads = [torch.zeros_like(p) for p in params]
d = ops.DeformMLPFunction._desc(params, grads)
feat, a0 = mk(P, 64) * 3, torch.relu(mk(P, 64))
dpts, dsc, drot = mk(P, 3), mk(P, 3), mk(P, 4)
dfeat = torch.empty(P, 64, device="cuda")
lib, s = N.lib(), N.current_stream()
nb = lib.mom_deform_backward_scratch_bytes(P)
scratch = torch.zeros(nb, dtype=torch.uint8, device="cuda")
for _ in range(3):
    N.check(lib.mom_deform_backward_split(C.byref(d), P, feat.data_ptr(), a0.data_ptr(), dpts.data_ptr(), dsc.data_ptr(), drot.data_ptr(),
                                          dfeat.data_ptr(), scratch.data_ptr(), s, s), "bwd")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    N.check(lib.mom_deform_backward_split(C.byref(d), P, feat.data_ptr(), a0.data_ptr(), dpts.data_ptr(), dsc.data_ptr(), drot.data_ptr(),
                                          dfeat.data_ptr(), scratch.data_ptr(), s, s), "bwd")
e1.record()
torch.cuda.synchronize()
print("kernel + reduce, alone: %.1f us per launch" % (e0.elapsed_time(e1) * 1e3 / 20))
if os.environ.get("MOM4D_LIB", "").endswith("gstamps.so"):
    # the eight-wave variant (-DB3G_STAMPS): per role wave, cycles alive and cycles spent polling a hand-over flag
    part_floats = 4 * (64 * 64 + 64) + 3 * (4 * 64 + 4)
    off = 256 * part_floats * 4
    dbg = scratch[off:off + 256 * 8 * 2 * 8].cpu().numpy().view(np.uint64).reshape(256, 8, 2).astype(np.float64)
    tiles = (P + 31) // 32 / 256
    for w, name in enumerate(["head0 A", "head1 A", "head2 A", "trunk A", "head0 B", "head1 B", "head2 B", "trunk B"]):
        tot, wait = np.median(dbg[:, w, 0]) / tiles, np.median(dbg[:, w, 1]) / tiles
        print(f"{name}: {tot:.0f} cycles per tile, of which {wait:.0f} polling ({100 * wait / tot:.0f} %)")
    sys.exit(0)
if not os.environ.get("MOM4D_LIB", "").endswith("stamps.so"):
    sys.exit(0)
part_floats = 4 * (64 * 64 + 64) + 3 * (4 * 64 + 4)
off = 256 * part_floats * 4
dbg = scratch[off:off + 256 * 4 * 8 * 8].cpu().numpy().view(np.uint64).reshape(256, 4, 8).astype(np.float64)
tiles = (P + 31) // 32 / 256
names_h = ["x_request", "a0 wait + split Ba0", "h1 MFMAs + relu", "stage, dW2, dH1, stage", "split Bd + dA0 MFMAs", "wait slot", "publish", "dW phase"]
names_t = ["x_request", "wait heads", "read slots, mask, release", "stage dH0", "split + dfeat MFMAs + store", "-", "-", "dW phase"]
for role, names, w in (("head 0", names_h, 0), ("head 2", names_h, 2), ("trunk", names_t, 3)):
    med = np.median(dbg[:, w, :], axis=0) / tiles
    print(role, "cycles per tile (s_memtime ticks):", " | ".join(f"{n}: {v:.0f}" for n, v in zip(names, med)), "| total", f"{med.sum():.0f}")
