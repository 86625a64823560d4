"""Where a workgroup of render_fwd spends its life (csrc/raster_render.hip built with -DFWD_STAMPS; MOM4D_LIB names that build):
    tools/variants.sh raster_render.hip fwdstamps="-DFWD_STAMPS"
    MOM4D_LIB=iclr2025_3d-mom_amd/lib/var/fwdstamps.so python tools/probe/fwd_stamps.py
Per workgroup (thread 0): s_memtime at entry / range known / sorted / round 0's records in registers / loop left / pixel stores issued /
L1 epilogue done / stores acknowledged; s_memrealtime at entry and exit (100 MHz, one clock for the chip); HW_ID and XCC_ID.
Printed: the launch's extent on the common clock, the phases' share of the workgroups' summed lifetime, the same per CU (how
long the busiest / median CU is occupied and by what), and the start ramp."""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench

N = importlib.import_module("iclr2025_3d-mom_amd._native")
cfg = bench.CONFIGS[os.environ.get("KBENCH_CONFIG", "c2")]
scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
cams = trainer.cams
for i in range(40):
    trainer.step(5001 + i, cams=[cams[(17 * i) % len(cams)]])
trainer.drain()
torch.cuda.synchronize()
real = C.CDLL(N.LIB_PATH)
gx, gy = (cfg["W"] + 15) // 16, (cfg["H"] + 15) // 16
nt = gx * gy
K = 16
rows, rows_b = [], []
for rep in range(5):
    trainer.step(5050 + rep, cams=[cams[(17 * (40 + rep)) % len(cams)]])
    trainer.drain()
    torch.cuda.synchronize()
    buf = np.zeros(nt * K, dtype=np.uint64)
    rc = real.mom_debug_fwd_stamps(buf.ctypes.data_as(C.c_void_p), nt)
    assert rc == 0, rc
    rows.append(buf.reshape(nt, K).copy())
    bufb = np.zeros(nt * K, dtype=np.uint64)
    assert real.mom_debug_bwd_stamps(bufb.ctypes.data_as(C.c_void_p), nt) == 0
    rows_b.append(bufb.reshape(nt, K).copy())
out = os.environ.get("STAMPS_OUT")
if out:
    np.savez_compressed(out, fwd=np.stack(rows), bwd=np.stack(rows_b))

names = ["entry->range", "sort", "round-0 fetch", "loop (staging, lists, compositing)", "walked store + pixel stores", "L1 epilogue", "store ack"]
for st in rows[-2:]:
    t = st[:, :8].astype(np.float64)
    d = np.diff(t, axis=1)                                    # seven phases, shader cycles
    life = t[:, 7] - t[:, 0]
    r0, r1 = st[:, 8].astype(np.float64), st[:, 9].astype(np.float64)
    span = (r1.max() - r0.min()) / 100.0                       # us (100 MHz)
    hw = (st[:, 10] & 0xFFFFFFFF).astype(np.int64)
    xcc = (st[:, 10] >> 32).astype(np.int64) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    keys = (st[:, 11] & 0xFFFFFFFF).astype(np.int64)
    clk = np.median(life / np.maximum((r1 - r0) / 100.0, 1e-3)) / 1e3          # GHz
    print(f"launch: {span:.1f} us from the first entry to the last exit; shader clock {clk:.2f} GHz; {len(np.unique(cu_key))} CUs seen; "
          f"keys per tile mean {keys.mean():.0f} max {keys.max()}")
    print(f"  start ramp: entries spread over {(r0.max() - r0.min()) / 100.0:.1f} us (p50 {np.percentile(r0 - r0.min(), 50) / 100:.1f}, "
          f"p99 {np.percentile(r0 - r0.min(), 99) / 100:.1f}); exits p1 {np.percentile(r1 - r0.min(), 1) / 100:.1f} p50 "
          f"{np.percentile(r1 - r0.min(), 50) / 100:.1f} p99 {np.percentile(r1 - r0.min(), 99) / 100:.1f} us")
    bw = st[:, 12:16].astype(np.float64)                      # cycles each wave spent at the rounds' two barriers
    loop_c = t[:, 4] - t[:, 3]
    print(f"  rounds' barriers: the four waves wait {100 * bw.mean() / max(loop_c.mean(), 1):.1f} % of the loop's time on average "
          f"(the least waiting wave {100 * bw.min(axis=1).mean() / max(loop_c.mean(), 1):.1f} %, the most {100 * bw.max(axis=1).mean() / max(loop_c.mean(), 1):.1f} %); "
          f"median per workgroup {np.median(bw.mean(axis=1)) / clk / 1e3:.1f} us of a {np.median(loop_c) / clk / 1e3:.1f} us loop")
    tot = d.sum()
    print("  share of the workgroups' summed lifetime (wave 0's view): " + " | ".join(f"{n} {100 * d[:, k].sum() / tot:.1f} %" for k, n in enumerate(names)))
    print("  per workgroup, us (median / p90 / max): " + " | ".join(f"{n} {np.median(d[:, k]) / clk / 1e3:.1f}/{np.percentile(d[:, k], 90) / clk / 1e3:.1f}/{d[:, k].max() / clk / 1e3:.1f}"
                                                                     for k, n in enumerate(names)))
    print(f"  workgroup lifetime us: median {np.median(life) / clk / 1e3:.1f} p90 {np.percentile(life, 90) / clk / 1e3:.1f} max {life.max() / clk / 1e3:.1f}")
    # per CU: when its last workgroup leaves, and how many it ran
    ends, counts, keysum = {}, {}, {}
    for k, e, n in zip(cu_key, r1, keys):
        ends[k] = max(ends.get(k, 0), e)
        counts[k] = counts.get(k, 0) + 1
        keysum[k] = keysum.get(k, 0) + n
    e = (np.array(list(ends.values())) - r0.min()) / 100.0
    c = np.array(list(counts.values()))
    ks = np.array(list(keysum.values()))
    print(f"  per CU: last exit median {np.median(e):.1f} p10 {np.percentile(e, 10):.1f} max {e.max():.1f} us; workgroups per CU min {c.min()} median {np.median(c):.0f} max {c.max()}; "
          f"keys per CU min {ks.min()} median {np.median(ks):.0f} max {ks.max()}; corr(keys, last exit) {np.corrcoef(ks, e)[0, 1]:.2f}")
    # correlation of a workgroup's lifetime with its list length
    print(f"  lifetime ~ keys: corr {np.corrcoef(keys, life)[0, 1]:.2f}; lifetime of tiles with <64 keys: median {np.median(life[keys < 64]) / clk / 1e3 if (keys < 64).any() else float('nan'):.1f} us "
          f"({int((keys < 64).sum())} tiles)")
