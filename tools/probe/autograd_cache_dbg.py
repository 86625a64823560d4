import sys, os, importlib
sys.path.insert(0, os.getcwd())
import torch, bench
fa = importlib.import_module("iclr2025_3d-mom_amd.fused_autograd")
ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
cfg = bench.CONFIGS["c2"]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda", 0), fused=False)
orig = fa._field_grads
def wrapped(st, f, direct=True):
    field = st.field
    c = getattr(field, "_fa_grads", None)
    if c is not None:
        held = [p.grad if direct else None for p in st.planes]
        in_place = [gg is not None and ops._same_layout(gg, p) and ops._dense(gg) for gg, p in zip(held, st.planes)]
        key = (tuple(p.data_ptr() for p in st.planes), tuple(p.data_ptr() for p in st.mlp), tuple(field.aabb_host()),
               tuple(gg.data_ptr() if ip else 0 for gg, ip in zip(held, in_place)))
        import sys as _s
        print("key_same", c[0] == key, "k0", c[0][0] == key[0], "k1", c[0][1] == key[1], "k3", c[0][3] == key[3], "inplace", sum(in_place),
              "mlpgrad", st.mlp[0].grad is not None, "free3", ops._buffers_free(c[3]), "free2", ops._buffers_free(c[2], 2),
              "rc3", [_s.getrefcount(v) for v in c[3]][:3], "rc2", [_s.getrefcount(v) for v in c[2]][:3])
    return orig(st, f, direct)
fa._field_grads = wrapped
for i in range(6):
    trainer.step(5001 + i, cams=[trainer.cams[i]])
torch.cuda.synchronize()
c = ops.PlaneRegFunction._cache
print("reg cache", c is not None)
