"""The autograd-free launch sequence (fused_step.py) must reproduce render() + loss.backward() + Adam."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(fused, steps=3, lambda_dssim=0.0):
    import bench
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=fused, lambda_dssim=lambda_dssim)
    assert (trainer.fused is not None) == fused
    losses = []
    for it in range(steps):
        cam = trainer.cams[(3 * it + 1) % len(trainer.cams)]
        losses.append(float(trainer.step(5001 + it, cams=[cam])))
    torch.cuda.synchronize()
    dn = g._deformation.deformation_net
    out = {"xyz": g._xyz, "f_dc": g._features_dc, "f_rest": g._features_rest, "scaling": g._scaling, "rotation": g._rotation,
           "opacity": g._opacity, "plane_xy": dn.grid.grids[1][0], "plane_zt": dn.grid.grids[0][5],
           "w0": dn.feature_out[0].weight, "b_rot": dn.rotations_deform[3].bias, "w_pos1": dn.pos_deform[1].weight,
           "accum": g.xyz_gradient_accum, "denom": g.denom, "maxr": g.max_radii2D}
    params = {k: v.detach().float().cpu().numpy().copy() for k, v in out.items()}
    # Adam's first moment after ONE step is (1 - beta1) * gradient exactly (it starts at zero), so it is the gradient the
    # step computed, read without Adam's normalisation in the way
    moments, lr_max = {}, max(grp["lr"] for grp in g.optimizer.param_groups)
    if steps == 1:
        for k in ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity", "plane_xy", "plane_zt", "w0", "b_rot", "w_pos1"):
            moments[k] = g.optimizer.state[out[k]]["exp_avg"].detach().float().cpu().numpy().copy()
    return losses, params, moments, lr_max


@pytest.mark.parametrize("lambda_dssim", [0.0, 0.2])
def test_fused_step_matches_autograd_path(lambda_dssim):
    """lambda_dssim 0 is the reference's default loss, 0.2 adds the SSIM term (train_4DGS.py:222-223).

    What must agree is the GRADIENT (both paths run the same kernels; only the order of their float atomics differs), so
    that is compared directly, through Adam's first moment after one step.  Parameters after three steps are compared too,
    but Adam divides by sqrt(v) + 1e-15: an element whose tiny gradient changes sign under a different summation order
    moves by a full learning rate (as it does between two runs of the reference itself), so a handful of elements may
    differ by up to 2 * lr per step while all the others agree tightly."""
    la, _, ma, _ = _run(False, steps=1, lambda_dssim=lambda_dssim)
    lf, _, mf, _ = _run(True, steps=1, lambda_dssim=lambda_dssim)
    np.testing.assert_allclose(lf, la, rtol=2e-5)
    for k in ma:
        scale = max(1e-30, float(np.abs(ma[k]).max()))
        err = float(np.abs(mf[k] - ma[k]).max())
        assert err <= 5e-5 * scale, ("gradient", k, err, scale)

    steps = 3
    la, pa, _, lr_max = _run(False, steps=steps, lambda_dssim=lambda_dssim)
    lf, pf, _, _ = _run(True, steps=steps, lambda_dssim=lambda_dssim)
    np.testing.assert_allclose(lf, la, rtol=2e-5)
    for k in pa:
        a, b = pf[k], pa[k]
        scale = max(1e-12, float(np.abs(b).max()))
        diff = np.abs(a - b)
        tight = 2e-4 * scale + 1e-6
        outliers = float((diff > tight).mean())
        assert outliers <= 1e-4, (k, "fraction of elements outside the tight tolerance", outliers)
        assert float(diff.max()) <= 2.0 * steps * lr_max * 1.01 + tight, (k, float(diff.max()), lr_max)
    np.testing.assert_array_equal(pf["denom"], pa["denom"])
    np.testing.assert_array_equal(pf["maxr"], pa["maxr"])


class _VirtualRank:
    """Stands in for parallel.DistContext on one GPU: captures each buffer at start() and then poisons it, which is what
    a concurrent in-place all-reduce does to it from the point of view of any later kernel that still reads it."""
    mode = "camera"

    def __init__(self, world):
        self.world, self.captured = world, []

    def start(self, tensor, op="sum"):
        assert tensor.is_contiguous()
        if tensor.dtype == torch.int32 and tensor.numel() == 1:      # the sticky overflow word (max over the ranks): leave it
            assert op == "max"
            return
        if tensor.dtype == torch.int32:      # camera-batch shard: [radii (P) | the overflow word], one max-all-reduce; the word is left alone
            assert op == "max"
            self.captured.append((op, tensor[:-1].clone()))
            tensor[:-1].fill_(-7)
            return
        self.captured.append((op, tensor.clone()))
        tensor.fill_(float("nan"))

    def finish(self):
        pass

    def wait_for(self, works):
        pass


def test_camera_batch_buckets_of_two_virtual_ranks_sum_to_the_batch_mean():
    """Multi-GPU exchange of the fused step, checked on one GPU: with world = 2 each rank's buckets carry 1/2, so the
    sum of what two ranks hand to start() must equal the mean of two plain single-GPU backward passes -- the
    reference's batch_size = 2 gradient (train_4DGS.py:189-229) -- and the radii their maximum.  Poisoning the buffers
    at start() proves no later kernel of the step reads a bucket that is being reduced."""
    import bench
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.2)
    fs, cams = trainer.fused, [trainer.cams[1], trainer.cams[4]]

    def snapshot():
        torch.cuda.synchronize()
        return {"radii": fs.radii.clone(), "early": fs.early_bucket.clone(), "late": fs._dg_flat.clone()}

    plain = []
    for cam in cams:                                   # no optimiser step in between: both see the same model
        fs.forward_backward(cam, 1)
        plain.append(snapshot())
    ranks = []
    for cam in cams:
        fs.dist = _VirtualRank(2)
        fs.forward_backward(cam, 1)
        torch.cuda.synchronize()
        ops_seen = [o for o, _ in fs.dist.captured]
        # three collectives per step: [radii | overflow word] (max), [appearance gradients 56 P | mean 2-D gradients 3 P] (sum), the
        # late bucket (sum) -- every torch.distributed call costs a rank's host 40-50 us
        assert ops_seen == ["max", "sum", "sum"], ops_seen
        ranks.append(dict(zip(("radii", "early", "late"), [t for _, t in fs.dist.captured])))
    fs.dist = None
    assert ranks[0]["early"].numel() == 59 * 6000 and ranks[0]["late"].numel() == fs._dg_flat.numel()
    assert fs.g2d.data_ptr() == fs.early_bucket[56 * 6000:].data_ptr() and int(fs.flags[0]) == 0
    torch.testing.assert_close(torch.maximum(ranks[0]["radii"], ranks[1]["radii"]),
                               torch.maximum(plain[0]["radii"], plain[1]["radii"]), rtol=0, atol=0)
    for k in ("early", "late"):
        got = ranks[0][k] + ranks[1][k]
        want = 0.5 * (plain[0][k] + plain[1][k])
        assert torch.isfinite(got).all(), k
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * scale + 1e-9, (k, float((got - want).abs().max()), scale)


from virtual_ranks import VirtualWorld  # noqa: E402

# the collectives of one tile-row step, in order: deformed state (gather of five arrays), compositing record (sum), position gradients
# (gather), [loss sums | deformation gradients] (one sum: the loss sums ride in front of the gradient bucket since round 6)
N_COLLECTIVES = 4


@pytest.mark.parametrize("world,lambda_dssim,split", [(2, 0.0, None), (3, 0.0, None), (2, 0.2, None), (3, 0.2, None), (6, 0.2, None),
                                                      (3, 0.2, [(0, 1), (1, 5), (5, 6)]), (3, 0.0, [(0, 4), (4, 6), (6, 6)])])
def test_tile_row_shard_of_the_fused_step_reproduces_the_unsharded_step(world, lambda_dssim, split):
    """BASELINE config 4 on one GPU: `world` virtual ranks take the same step -- each runs the deformation field on its slice of
    the Gaussians and renders its own tile rows of the same camera; they exchange the deformed state, the per-Gaussian record of
    the compositing backward, the position gradients, the deformation gradients and the loss sums -- and must all end with the
    unsharded step's gradients, statistics and loss.  With the SSIM term every rank also renders a one-tile-row halo and evaluates
    SSIM on that slab (no pixels are exchanged); world 6 gives every rank a single tile row, so every boundary has a halo.  The
    uneven splits are what a rebalance produces, one of them leaving the last rank without rows."""
    import bench
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")            # 6 tile rows
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=lambda_dssim)
    fs, cam = trainer.fused, trainer.cams[2]

    def run(dist):
        fs.dist = dist
        loss, radii, g2d = fs.forward_backward(cam, 1)
        torch.cuda.synchronize()
        return {"loss": float(loss), "radii": radii.clone(), "g2d": g2d.clone(), "early": fs.early.clone(),
                "late": fs._dg_flat[:fs._dg_n + 3 * cfg["P"]].clone(), "mse": float(fs.last["mse_sum"]), "rows": None if dist is None else dist.rows(6)}

    want = run(None)
    vw = VirtualWorld(world, split=split)
    results, ranks = vw.run(run)
    assert len(vw.resolved) == N_COLLECTIVES and [k[0][0] for k in vw.resolved] == ["gather", "reduce", "gather", "reduce"]
    rows = [r["rows"] for r in results]
    assert rows[0][0] == 0 and rows[-1][1] == 6 and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    for r, got in enumerate(results):
        assert abs(got["loss"] - want["loss"]) <= 1e-6 * max(1.0, abs(want["loss"])), (got["loss"], want["loss"])
        assert abs(got["mse"] - want["mse"]) <= 1e-5 * abs(want["mse"])
        torch.testing.assert_close(got["radii"], want["radii"], rtol=0, atol=0)
        for k in ("g2d", "early", "late"):
            scale = float(want[k].abs().max())
            err = float((got[k] - want[k]).abs().max())
            assert torch.isfinite(got[k]).all() and err <= 3e-5 * scale + 1e-9, (r, k, err, scale)
    # every rank ends with the SAME bits where the step promises replicas: what Adam consumes
    for got in results[1:]:
        for k in ("g2d", "early", "late"):
            assert torch.equal(got[k], results[0][k]), k
    fs.dist = None


def test_tile_row_rebalance_inside_the_step_keeps_the_gradients():
    """The fused step's own rebalance branch (every DistContext.REBALANCE_EVERY steps): it reads the per-row instance counts,
    agrees on a new split and must still finish THIS iteration on the buffers its forward filled -- the re-sized binning buffer
    is for the next iteration.  Both the rebalance iteration and the one after it must reproduce the unsharded gradients."""
    import bench
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")            # 6 tile rows
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.2)
    fs, cam = trainer.fused, trainer.cams[2]
    world, old, new = 3, [(0, 2), (2, 4), (4, 6)], [(0, 1), (1, 5), (5, 6)]

    want = None
    for splits, resplit in ((None, None), (old, new), (new, None)):       # unsharded; the rebalance iteration; the one after it (new rows)
        def run(dist):
            fs.dist = dist
            fs._resize_next = resplit is None                 # the iteration after a re-split sizes its buffer exactly
            loss, radii, g2d = fs.forward_backward(cam, 1)
            torch.cuda.synchronize()
            out = {"loss": float(loss), "g2d": g2d.clone(), "early": fs.early.clone(), "late": fs._dg_flat[:fs._dg_n + 3 * cfg["P"]].clone(),
                   "resize_next": fs._resize_next}
            return out
        if splits is None:
            want = run(None)
            continue
        results, ranks = VirtualWorld(world, split=list(splits), resplit=resplit).run(run)
        for d, got in zip(ranks, results):
            if resplit is not None:
                assert d.split == new and got["resize_next"]    # adopted, and the NEXT step will re-size
                own = d.row_counts
                assert float(own.sum()) > 0 and float(own[:splits[d.rank][0]].sum()) == 0 and float(own[splits[d.rank][1]:].sum()) == 0
            assert abs(got["loss"] - want["loss"]) <= 1e-6 * max(1.0, abs(want["loss"]))
            for k in ("g2d", "early", "late"):
                scale = float(want[k].abs().max())
                err = float((got[k] - want[k]).abs().max())
                assert torch.isfinite(got[k]).all() and err <= 3e-5 * scale + 1e-9, (resplit is not None, d.rank, k, err, scale)
    fs.dist = None


def _snapshot(g):
    dn = g._deformation.deformation_net
    t = {"xyz": g._xyz, "f_dc": g._features_dc, "scaling": g._scaling, "rotation": g._rotation, "opacity": g._opacity,
         "plane_xy": dn.grid.grids[1][0], "w0": dn.feature_out[0].weight, "accum": g.xyz_gradient_accum, "denom": g.denom,
         "maxr": g.max_radii2D}
    out = {k: v.detach().float().cpu().numpy().copy() for k, v in t.items()}
    out["adam_steps"] = sorted({float(st["step"]) for st in g.optimizer.state.values()})
    return out


def test_binning_overflow_skips_the_update_on_the_device_and_the_host_replays_it():
    """The host sizes a step's binning buffer from earlier frames and runs ahead of the GPU.  A frame that does not fit must
    not reach the model: Adam and the densification statistics of that step (and of the ones queued behind it) are no-ops on
    the device, the host finds the sticky flag FLAG_LAG steps later, and replays the skipped iterations with exactly sized
    buffers.  The result must equal a run that never overflowed: statistics counts exactly, Adam step counters exactly,
    parameters to float-atomic rounding."""
    import bench
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")
    seq = [(5001 + i, (3 * i + 1) % 9) for i in range(24)]

    def run(force):
        scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.2)
        fs = trainer.fused
        for i, (it, ci) in enumerate(seq):
            if force and i == 6:         # three frames whose buffer holds a quarter of their instances
                fs.HEADROOM, fs.MARGIN = 0.25, 0
                fs.cap, fs.binning = 1, None
            if force and i == 9:
                fs.HEADROOM, fs.MARGIN = type(fs).HEADROOM, type(fs).MARGIN
            trainer.step(it, cams=[trainer.cams[ci % len(trainer.cams)]])
        trainer.drain()
        torch.cuda.synchronize()
        assert int(fs.flags[0]) == 0
        return _snapshot(g), trainer.replayed

    clean, n0 = run(False)
    forced, n1 = run(True)
    assert n0 == 0 and n1 >= 3, (n0, n1)
    assert forced["adam_steps"] == clean["adam_steps"] == [float(len(seq))]
    np.testing.assert_array_equal(forced["denom"], clean["denom"])
    np.testing.assert_array_equal(forced["maxr"], clean["maxr"])
    for k in ("xyz", "f_dc", "scaling", "rotation", "opacity", "plane_xy", "w0", "accum"):
        a, b = forced[k], clean[k]
        scale = max(1e-12, float(np.abs(b).max()))
        frac = float((np.abs(a - b) > 1e-3 * scale + 1e-6).mean())
        assert frac <= 2e-3, (k, frac)


@pytest.mark.parametrize("mode", ["exact", "async"])
def test_nograd_render_fast_path_matches_the_regular_path(mode):
    """gaussian_renderer.render() under torch.no_grad() takes the forward-only launch sequence (fused_render.py); with
    gradients enabled it takes the autograd-capable path.  Same kernels for the deformation and the rasterizer, different
    activations (one kernel vs torch ops): images to 1e-6 on average, radii exact; the returned dictionary has the same keys, and the
    returned tensors are the caller's (a later frame must not overwrite them)."""
    import bench
    R = importlib.import_module("iclr2025_3d-mom_amd.gaussian_renderer")
    DGR = importlib.import_module("iclr2025_3d-mom_amd.diff_gaussian_rasterization")
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")
    scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True)
    cams = scene.getVideoCameras_side()[:6]
    DGR.set_sync_mode("exact")
    slow = [R.render(c, g, trainer.pipe, trainer.background, stage="fine", cam_type=scene.dataset_type, delta_scale=1) for c in cams]
    DGR.set_sync_mode(mode)
    try:
        with torch.no_grad():
            fast = [R.render(c, g, trainer.pipe, trainer.background, stage="fine", cam_type=scene.dataset_type, delta_scale=1)
                    for c in cams]
        torch.cuda.synchronize()
    finally:
        DGR.set_sync_mode("exact")
    assert getattr(g, "_fused_render", None) is not None                     # the fast path ran
    assert len({f["render"].data_ptr() for f in fast}) == len(fast)         # every frame has its own image
    for a, b in zip(fast, slow):
        assert set(a.keys()) == set(b.keys())
        # the activations differ in their last bits (one kernel with expf vs torch's exp / sigmoid / normalize), which moves a
        # few pixels by ~1e-5: mean far inside the 1e-4 parity bar, maximum bounded
        d_img = (a["render"] - b["render"].detach()).abs()
        assert float(d_img.mean()) <= 1e-6 and float(d_img.max()) <= 1e-4, (float(d_img.mean()), float(d_img.max()))
        d_dep = (a["depth"] - b["depth"].detach()).abs()
        assert float(d_dep.mean()) <= 1e-5 and float(d_dep.max()) <= 1e-3, (float(d_dep.mean()), float(d_dep.max()))
        assert torch.equal(a["radii"], b["radii"]) and torch.equal(a["visibility_filter"], b["visibility_filter"])
        assert a["viewspace_points"].shape == b["viewspace_points"].shape


def test_processing_orders_are_refreshed_beside_the_steps_and_stay_permutations():
    """HexPlaneField refreshes its processing orders (one Morton sort, six plane sorts) every REORDER_EVERY calls; the refresh is
    launched REFRESH_AHEAD calls early on a second stream and swapped in when due (scene/hexplane.py).  Whatever the positions were
    while it ran -- Adam writes them concurrently -- what is swapped in must be permutations with matching inverses; a prune round
    under a pending refresh must drop it; and training through several refreshes ends where it ends with the refresh on the step's
    own stream (the orders only decide the order of float atomics)."""
    import bench
    H = importlib.import_module("iclr2025_3d-mom_amd.scene.hexplane")
    cfg = dict(P=6000, F=4, W=160, H=96, time_res=10, name="tiny")

    def run(async_refresh):
        old = (H.HexPlaneField.ASYNC_REFRESH, H.HexPlaneField.REORDER_EVERY, H.HexPlaneField.REFRESH_AHEAD)
        H.HexPlaneField.ASYNC_REFRESH, H.HexPlaneField.REORDER_EVERY, H.HexPlaneField.REFRESH_AHEAD = async_refresh, 12, 4
        try:
            scene, g, trainer, op = bench.build_state(cfg, torch.device("cuda"), fused=True, lambda_dssim=0.0)
            field = g._deformation.deformation_net.grid
            seen_pending, swaps = 0, 0
            for it in range(40):
                before = field._order
                trainer.step(5001 + it, cams=[trainer.cams[it % len(trainer.cams)]])
                # churn the caller's stream's allocator: whatever the refresh on the second stream still uses must be kept alive by
                # the field, or these fills land in its scratch (a dropped reference here once gave garbage orders and a GPU fault)
                junk = [torch.full((n,), -1, dtype=torch.int32, device="cuda") for n in (6000, 24000, 6000 * 3 * 2, 40000, 100000)]
                del junk
                seen_pending += getattr(field, "_pending", None) is not None
                swaps += field._order is not before
                if it == 20:
                    # restructure the model -- in the async run under a pending refresh: the next call must not swap the stale orders in
                    if async_refresh:
                        if getattr(field, "_pending", None) is None:
                            field._prefetch_orders(g._xyz.detach(), trainer.fused.side.cuda_stream)
                        assert field._pending is not None
                    keep = torch.ones(g._xyz.shape[0], dtype=torch.bool, device="cuda")
                    keep[::7] = False
                    trainer.drain()
                    g.prune_points(~keep)
            trainer.drain()
            torch.cuda.synchronize()
            P = g._xyz.shape[0]
            order = field._order.long() & 0xFFFFFFFF
            assert order.shape[0] == P and torch.equal(torch.sort(order).values, torch.arange(P, device="cuda"))
            po, inv = field._porders
            for k in range(po.shape[0]):
                for lv in range(po.shape[1]):
                    o = po[k, lv].long()
                    assert torch.equal(torch.sort(o).values, torch.arange(P, device="cuda"))
                    assert torch.equal(inv[k, lv].long()[o], torch.arange(P, device="cuda"))
            return {"xyz": g._xyz.detach().clone(), "plane": field.grids[0][0].detach().clone()}, seen_pending, swaps
        finally:
            H.HexPlaneField.ASYNC_REFRESH, H.HexPlaneField.REORDER_EVERY, H.HexPlaneField.REFRESH_AHEAD = old

    a, pend_a, swaps_a = run(True)
    assert pend_a >= 6 and swaps_a >= 3            # refreshes were prefetched (several calls each) and swapped in
    b, pend_b, swaps_b = run(False)
    assert pend_b == 0 and swaps_b >= 3
    assert a["xyz"].shape == b["xyz"].shape and a["xyz"].shape[0] < 6000
    for k in a:
        scale = max(1e-12, float(b[k].abs().max()))
        assert float(((a[k] - b[k]).abs() > 1e-3 * scale + 1e-6).float().mean()) <= 2e-3, k


@pytest.mark.parametrize("P,world", [(20_003, 2), (20_003, 8), (4_000_000, 8)])
def test_sharded_adam_slices_of_every_rank_together_are_one_full_step_to_the_bit(P, world):
    """Sharded Adam (parallel.py; SURVEY 8e's "reduce-scatter + sharded Adam + all-gather of updated params"): with the gradients held
    fixed, the `world` ranks' FusedAdam.step_partial(ranges=DistContext.shard_ranges(...)) launches -- each on its own chunk of the
    flat appearance bucket [f_dc 3P | f_rest 45P | scaling 3P | rotation 4P | opacity P] -- leave parameters and both moments
    exactly where ONE replicated step leaves them: same bits, two steps in a row (the second with non-zero moments), chunks that
    cut tensors at unaligned offsets (P odd) and BASELINE configs[4]'s per-GPU model (4 M Gaussians, 8 ranks)."""
    ops = importlib.import_module("iclr2025_3d-mom_amd.ops")
    par = importlib.import_module("iclr2025_3d-mom_amd.parallel")
    dev = torch.device("cuda")
    gen = torch.Generator(device="cuda").manual_seed(5)
    shapes = ((1, 3), (15, 3), (3,), (4,), (1,))
    lrs = (2.5e-3, 1.25e-4, 5e-3, 1e-3, 5e-2)
    cuts = [0]
    for shp in shapes:
        cuts.append(cuts[-1] + P * int(np.prod(shp)))

    def make():
        g0 = torch.Generator(device="cuda").manual_seed(11)
        ps = [torch.nn.Parameter(torch.randn((P,) + shp, device=dev, generator=g0)) for shp in shapes]
        return ps, ops.FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)], eps=1e-15)

    pa, oa = make()
    pb, ob = make()
    ranks = [par.DistContext(r, world, shard_adam=True) for r in range(world)]
    chunk = ranks[0].chunk(cuts[-1])
    assert chunk % 4 == 0 and world * chunk >= cuts[-1]
    covered = 0
    for step in range(2):
        grads = [torch.randn((P,) + shp, device=dev, generator=gen) * (10.0 ** -step) for shp in shapes]
        for p, q, g in zip(pa, pb, grads):
            p.grad, q.grad = g, g.clone()
        oa.step_partial(pa)                                        # the replicated step
        oa.step()
        for r, dc in enumerate(ranks):                             # every rank's slice, one after the other, on the second copy
            rg = dc.shard_ranges(cuts, chunk)
            if step == 0:
                covered += sum(n for _, n in rg)
            ob.step_partial(pb, ranges={id(p): x for p, x in zip(pb, rg)})
            ob._early = None
            if r + 1 < world:
                ob.rewind(1)                                       # (one optimizer stands in for `world` of them: one step, not eight)
        torch.cuda.synchronize()
        for k, (p, q) in enumerate(zip(pa, pb)):
            assert torch.equal(p.data, q.data), (step, k)
            for key in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(oa.state[p][key], ob.state[q][key]), (step, k, key)
            assert float(oa.state[p]["step"]) == float(ob.state[q]["step"]) == step + 1
    assert covered == cuts[-1]                                     # the chunks cover every element exactly once
