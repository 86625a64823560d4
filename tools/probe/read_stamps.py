import ctypes as C, os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
N = importlib.import_module("iclr2025_3d-mom_amd._native")
scene, g, trainer, op = bench.build_state(bench.CONFIGS["c2"], torch.device("cuda"), fused=True)
for i in range(20): trainer.step(5001 + i, cams=[trainer.cams[i % 65]])
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 16))()
N.lib().mom_debug_read_stamps.argtypes = [C.c_void_p]
N.lib().mom_debug_read_stamps(buf)
import numpy as np
full = np.array(buf[:], dtype=np.uint64).reshape(64, 16).astype(np.int64)
d = np.diff(full[:, :5], axis=1)
print("phases (s_memtime ticks): head0 MFMA || store a0 | head1 MFMA || out0 | head2 MFMA || out1 | next trunk MFMA || out2")
print("median", np.median(d, axis=0)); print("wave 0", d[0]); print("wave 37", d[37])
