"""One rank of tests/test_two_process_gpu.py: TWO real processes share GPU 0, rendezvous over gloo (which moves device tensors in
this torch build; RCCL refuses two ranks on one device) and run three fused training steps through parallel.DistContext --
start() / finish() with in-place asynchronous all-reduces of device buffers across a process boundary.
usage: two_process_rank.py <camera|camera-sharded|tile-row> <out prefix>
camera-sharded: the camera-batch shard with sharded Adam (reduce-scatter + 1/world of Adam + all-gather of the parameters)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

CFG = dict(P=20000, F=6, W=320, H=192, time_res=12, name="small")


def snapshot(g):
    dn = g._deformation.deformation_net
    planes = [p for lv in dn.grid.grids for p in lv]
    out = {"xyz": g._xyz, "opacity": g._opacity, "f_dc": g._features_dc, "scaling": g._scaling, "plane_xy": planes[0],
           "plane_xt": planes[2], "w0": dn._fused_params()[0], "accum": g.xyz_gradient_accum, "denom": g.denom,
           "maxr": g.max_radii2D, "f_rest": g._features_rest, "rotation": g._rotation}
    st = getattr(g, "optimizer", None)
    if st is not None and g._features_rest in st.state and "exp_avg" in st.state[g._features_rest]:
        out["m_f_rest"] = st.state[g._features_rest]["exp_avg"]
        out["v_opacity"] = st.state[g._opacity]["exp_avg_sq"]
    return out


def main():
    mode, prefix = sys.argv[1], sys.argv[2]
    import datetime
    import faulthandler
    import time
    rank_env = os.environ.get("RANK", "?")
    t_start = time.time()
    log = open(f"{prefix}_{rank_env}.log", "w")

    def stamp(what):
        """Progress marks of this rank (the parent test attaches both ranks' files to a failure: where each one stood, and when)."""
        log.write(f"{time.time() - t_start:8.2f} s  {what}\n")
        log.flush()

    stamp("process up, torch imported")
    # a hung rank says where: its Python stacks go to its log file (the parent's stderr is captured by pytest and easily lost)
    faulthandler.dump_traceback_later(float(os.environ.get("MOM_RANK_STACKS_AFTER", "100")), exit=False, file=log)
    torch.cuda.set_device(0)
    # a collective that cannot complete raises after the timeout instead of waiting for ever (gloo honours it per operation;
    # DistContext passes its own to every wait)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("MOM_PG_TIMEOUT_S", "90"))))
    rank, world = dist.get_rank(), dist.get_world_size()
    stamp("process group up")
    par = importlib.import_module("iclr2025_3d-mom_amd.parallel")
    scene, g, trainer, op = bench.build_state(CFG, torch.device("cuda"), fused=True, lambda_dssim=0.2)
    par.attach(trainer, rank, world, mode="camera" if mode == "camera-sharded" else mode, shard_adam=(mode == "camera-sharded"))
    stamp("state built")
    cams = trainer.cams
    for i in range(3):
        cam = cams[(i * world + rank) % len(cams)] if mode.startswith("camera") else cams[i % len(cams)]
        trainer.step(5001 + i, cams=[cam])
        stamp(f"step {i} enqueued (replayed so far: {trainer.replayed})")
    trainer.drain()
    trainer.fused.gather_moments()             # (sharded Adam: every rank's moments whole again, as before a checkpoint)
    stamp("drained")
    torch.cuda.synchronize()
    stamp("device idle")
    torch.save({k: v.detach().cpu() for k, v in snapshot(g).items()}, f"{prefix}_{rank}.pt")
    dist.barrier()
    stamp("barrier passed")
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    stamp("done")
    if rank == 0:
        print("done", flush=True)


if __name__ == "__main__":
    main()
