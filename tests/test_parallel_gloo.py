"""N > 1 path on the CPU: world_size-2 gloo run of the camera-batch shard (oracle backend standing in for libmom4d)
must reproduce the single-process batch_size=2 step of the reference semantics (train_4DGS.py:172-229)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(P=1500, F=4, W=80, H=48, time_res=10, name="tiny")


def _run_steps(rank, world, port, out):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from oracle import cpu_backend, raster_oracle as ro
    ro.set_threads(1)
    import bench
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    with cpu_backend.installed():
        scene, g, trainer, op = bench.build_state(CFG, "cpu")
        cams = trainer.cams
        if world > 1:
            importlib.import_module("iclr2025_3d-mom_amd.parallel").attach(trainer, rank, world)
        else:
            op.batch_size = 2
        for it in range(3):
            pair = [cams[(2 * it) % len(cams)], cams[(2 * it + 1) % len(cams)]]
            trainer.step(5001 + it, cams=[pair[rank]] if world > 1 else pair)
        res = {"xyz": g._xyz.detach().numpy().copy(), "opacity": g._opacity.detach().numpy().copy(),
               "f_rest": g._features_rest.detach().numpy().copy(),
               "grid0": g._deformation.deformation_net.grid.grids[0][2].detach().contiguous().numpy().copy(),
               "w": g._deformation.deformation_net.pos_deform[3].weight.detach().numpy().copy(),
               "accum": g.xyz_gradient_accum.numpy().copy(), "maxr": g.max_radii2D.numpy().copy()}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    np.savez(out, **res)


@pytest.mark.timeout(600)
def test_camera_batch_shard_matches_single_process_batch(tmp_path):
    port = 29500 + os.getpid() % 2000
    outs = [str(tmp_path / f"r{r}.npz") for r in range(2)]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run_steps, args=(r, 2, port, outs[r])) for r in range(2)]
    for p in procs:
        p.start()
    single = str(tmp_path / "single.npz")
    _run_steps(0, 1, 0, single)
    for p in procs:
        p.join(timeout=500)
        assert p.exitcode == 0
    a, b, s = np.load(outs[0]), np.load(outs[1]), np.load(single)
    for k in s.files:
        np.testing.assert_array_equal(a[k], b[k])                      # replicas stay bit-identical
        np.testing.assert_allclose(a[k], s[k], rtol=2e-5, atol=1e-7)   # and equal the batch_size=2 reference semantics


def _run_async_buckets(rank, world, port, out):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = importlib.import_module("iclr2025_3d-mom_amd.parallel")
    dc = P.DistContext(rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    radii = torch.randint(0, 50, (257,), generator=g, dtype=torch.int32)
    early, late = torch.randn(56 * 257, generator=g), torch.randn(4099, generator=g)
    mine = {"radii": radii.numpy().copy(), "early": early.numpy().copy(), "late": late.numpy().copy()}
    # the order fused_step.py issues them in -- the integer bucket is a VIEW of a larger store ([radii (P) | overflow word] of a
    # capacity-sized buffer), as the step hands it over; start() returns the work handles and wait_for() waits for exactly those
    # (what the step's second stream does before the early Adam launch) while they stay pending for finish()
    store = torch.full((300,), -5, dtype=torch.int32)
    store[:257] = radii
    store[257] = rank                                   # the "overflow word": the max over the ranks must come back
    w0 = dc.start(store[:258], "max")
    w1 = dc.start(early, "sum")
    dc.wait_for([w0, w1])
    assert int(store[257]) == world - 1 and int(store[258]) == -5      # reduced, and nothing beyond the view was touched
    radii.copy_(store[:257])
    early_view_ok = True
    try:
        dc.start(early.view(257, 56).t(), "sum")      # not a whole contiguous buffer: refused, nothing enqueued
        early_view_ok = False
    except ValueError:
        pass
    dc.start(late, "sum")
    dc.finish()
    dc.finish()                     # idempotent: nothing pending
    dist.barrier()
    dist.destroy_process_group()
    np.savez(out, radii=radii.numpy(), early=early.numpy(), late=late.numpy(), refused=np.array(early_view_ok),
             **{"mine_" + k: v for k, v in mine.items()})


@pytest.mark.timeout(300)
def test_async_bucket_all_reduce_world2(tmp_path):
    """DistContext.start()/finish(): several in-place all-reduces in flight at once (max for the radii, sum for the two
    gradient buckets), finished together -- the protocol the fused step uses to overlap its exchange with the backward."""
    port = 31500 + os.getpid() % 2000
    outs = [str(tmp_path / f"a{r}.npz") for r in range(2)]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run_async_buckets, args=(r, 2, port, outs[r])) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=250)
        assert p.exitcode == 0
    a, b = np.load(outs[0]), np.load(outs[1])
    assert bool(a["refused"]) and bool(b["refused"])
    np.testing.assert_array_equal(a["radii"], np.maximum(a["mine_radii"], b["mine_radii"]))
    for k in ("early", "late"):
        np.testing.assert_array_equal(a[k], b[k])                                   # every rank holds the same sum
        np.testing.assert_allclose(a[k], a["mine_" + k] + b["mine_" + k], rtol=0, atol=0)
    np.testing.assert_array_equal(a["radii"], b["radii"])


def _run_gather(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("iclr2025_3d-mom_amd.parallel")
    dc = par.DistContext(rank, world, mode="tile-row")
    P = 1000                                            # 3 ranks: 352 rows each (a multiple of 32), the last rank owns 296
    S = dc.slice_rows(P)
    a = torch.full((world * S, 3), -1.0)
    b = torch.full((world * S, 4), -1.0)
    g0 = rank * S
    a[g0:g0 + S] = torch.arange(S * 3, dtype=torch.float32).view(S, 3) + 10000 * rank
    b[g0:g0 + S] = float(rank + 1)
    red = torch.full((7,), float(rank + 1))
    dc.start_gather([a, b], S)
    dc.start(red, "sum")                                # a reduction in flight beside the gathers
    dc.finish()
    np.savez(out, S=S, a=a.numpy(), b=b.numpy(), red=red.numpy())
    dist.destroy_process_group()


def test_slice_gather_world3(tmp_path):
    """DistContext.start_gather(): the in-place all-gather of row slices a tile-row shard uses for the deformed state and the
    position gradients -- world 3, uneven last slice, beside an all-reduce, every rank ends with every rank's rows."""
    port = 33500 + os.getpid() % 2000
    outs = [str(tmp_path / f"g{r}.npz") for r in range(3)]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run_gather, args=(r, 3, port, outs[r])) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=250)
        assert p.exitcode == 0
    res = [np.load(o) for o in outs]
    S = int(res[0]["S"])
    assert S == 352
    for r in res[1:]:
        np.testing.assert_array_equal(r["a"], res[0]["a"])
        np.testing.assert_array_equal(r["b"], res[0]["b"])
    for k in range(3):
        np.testing.assert_array_equal(res[0]["a"][k * S:(k + 1) * S], np.arange(S * 3, dtype=np.float32).reshape(S, 3) + 10000 * k)
        assert (res[0]["b"][k * S:(k + 1) * S] == k + 1).all()
        np.testing.assert_array_equal(res[k]["red"], np.full(7, 6.0, np.float32))


def test_split_rows_partitions():
    """parallel.split_rows: contiguous, ordered, complete; balanced by weight; never (0, 0) for an empty range."""
    split = importlib.import_module("iclr2025_3d-mom_amd.parallel").split_rows
    assert split(34, 8) == [(0, 4), (4, 8), (8, 12), (12, 17), (17, 21), (21, 25), (25, 29), (29, 34)]   # 960x540, 8 GPUs
    assert split(34, 2) == [(0, 17), (17, 34)]
    assert split(7, 1) == [(0, 7)]
    assert split(5, 8) == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]              # more ranks than rows
    assert split(1, 3) == [(0, 1), (1, 1), (1, 1)]
    assert split(6, 3, [10, 0, 0, 0, 0, 10]) == [(0, 1), (1, 5), (5, 6)]                               # by work, not by count
    assert split(6, 3, [1, 1, 1, 1, 1, 100]) == [(0, 4), (4, 5), (5, 6)]                               # no rank starved
    assert split(4, 4, [0, 0, 0, 0]) == [(0, 1), (1, 2), (2, 3), (3, 4)]
    rng = np.random.default_rng(0)
    for _ in range(200):
        n, w = int(rng.integers(1, 70)), int(rng.integers(1, 10))
        weights = None if rng.random() < 0.3 else rng.integers(0, 1000, n).tolist()
        parts = split(n, w, weights)
        assert len(parts) == w and [r for a, b in parts for r in range(a, b)] == list(range(n))
        assert all(a <= b for a, b in parts) and all(p != (0, 0) for p in parts)
        if n >= w:
            assert all(b > a for a, b in parts)
    for bad in ((0, 3), (3, 0)):
        with pytest.raises(ValueError):
            split(*bad)
    with pytest.raises(ValueError):
        split(3, 2, [1, 2])


def _run_rebalance(rank, world, port, out):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = importlib.import_module("iclr2025_3d-mom_amd.parallel")
    dc = P.DistContext(rank, world, mode="tile-row")
    n_rows = 9
    before = dc.rows(n_rows)
    counts = torch.tensor([5000., 4000., 3000., 200., 100., 50., 20., 10., 5.])      # the work sits in the top rows
    own = torch.zeros(n_rows)
    own[before[0]:before[1]] = counts[before[0]:before[1]]                             # a rank only knows its own rows
    due = [dc.rebalance_due() for _ in range(P.DistContext.REBALANCE_EVERY)]
    changed = dc.rebalance_rows(own)
    after = dc.rows(n_rows)
    dist.barrier()
    dist.destroy_process_group()
    np.savez(out, before=np.array(before), after=np.array(after), weights=np.array(dc.row_weights), changed=np.array(changed),
             due=np.array(due), other=np.array(dc.rows(7)))


@pytest.mark.timeout(300)
def test_tile_row_rebalance_agrees_across_ranks(tmp_path):
    """DistContext.rebalance_rows: every rank contributes the instance counts of its own tile rows, all ranks end with the same
    weights and therefore the same split, which moves rows towards the ranks with less work; an image with another number of
    tile rows keeps the plain split."""
    world, port = 3, 32500 + os.getpid() % 2000
    outs = [str(tmp_path / f"b{r}.npz") for r in range(world)]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run_rebalance, args=(r, world, port, outs[r])) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=250)
        assert p.exitcode == 0
    res = [np.load(o) for o in outs]
    for r in res[1:]:
        np.testing.assert_array_equal(r["weights"], res[0]["weights"])
    np.testing.assert_allclose(res[0]["weights"], np.array([5000., 4000., 3000., 200., 100., 50., 20., 10., 5.]) + 1.0)
    assert [tuple(r["before"]) for r in res] == [(0, 3), (3, 6), (6, 9)]
    after = [tuple(r["after"]) for r in res]
    assert after[0][0] == 0 and after[-1][1] == 9 and all(a[1] == b[0] for a, b in zip(after, after[1:]))
    assert after == [(0, 1), (1, 2), (2, 9)], after                       # 5001 | 4001 | the rest
    assert all(bool(r["changed"]) for r in res)
    for k, r in enumerate(res):
        assert list(r["due"]) == [False] * 15 + [True]
        assert tuple(r["other"]) == ((0, 2), (2, 4), (4, 7))[k]          # 7 rows: no weights for that height, plain split


def _adam_ref(p, g, m, v, step, lr=1e-2, b1=0.9, b2=0.999, eps=1e-15):
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.sub_(lr / (1 - b1 ** step) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps))


def _run_sharded(rank, world, port, out):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = importlib.import_module("iclr2025_3d-mom_amd.parallel")
    dc = P.DistContext(rank, world, shard_adam=True)
    n_g = 1001                                                 # "Gaussians": five tensors of 3, 45, 3, 4, 1 floats each, as the bucket
    widths = (3, 45, 3, 4, 1)
    cuts = [0]
    for w in widths:
        cuts.append(cuts[-1] + w * n_g)
    chunk = dc.chunk(cuts[-1])
    assert chunk % 4 == 0 and world * chunk >= cuts[-1] > (world - 1) * chunk
    gen = torch.Generator().manual_seed(7)
    params0 = torch.randn(world * chunk, generator=gen)
    res = {}
    for path in ("all_reduce", "sharded"):
        pflat = params0.clone()
        m, v = torch.zeros_like(pflat), torch.zeros_like(pflat)
        for step in (1, 2, 3):
            g = torch.Generator().manual_seed(1000 * step + rank)
            grad = torch.zeros(world * chunk)
            grad[:cuts[-1]] = torch.randn(cuts[-1], generator=g) / world
            if path == "all_reduce":
                dc.start(grad, "sum")
                dc.finish()
                _adam_ref(pflat[:cuts[-1]], grad[:cuts[-1]], m[:cuts[-1]], v[:cuts[-1]], step)
            else:
                dc.start_reduce_scatter(grad, chunk, "sum")
                dc.finish()
                for a, (off, n) in zip(cuts, dc.shard_ranges(cuts, chunk)):       # this rank's slice of each of the five tensors
                    if n:
                        sl = slice(a + off, a + off + n)
                        _adam_ref(pflat[sl], grad[sl], m[sl], v[sl], step)
                dc.start_gather_flat(pflat, chunk)
                dc.finish()
        if path == "sharded":
            # the moments of the other ranks' slices are stale until gathered (as before a densify / prune round or a checkpoint)
            class _Opt:
                state = {}
            tensors = [pflat[a:b].clone() for a, b in zip(cuts[:-1], cuts[1:])]
            opt = _Opt()
            opt.state = {t: {"exp_avg": m[a:b].clone(), "exp_avg_sq": v[a:b].clone()} for t, a, b in zip(tensors, cuts[:-1], cuts[1:])}
            dc.gather_moments(opt, tensors, cuts, chunk)
            m = torch.cat([opt.state[t]["exp_avg"] for t in tensors])
            v = torch.cat([opt.state[t]["exp_avg_sq"] for t in tensors])
        res[path] = (pflat[:cuts[-1]].clone(), m[:cuts[-1]].clone(), v[:cuts[-1]].clone())
    ranges = dc.shard_ranges(cuts, chunk)
    # first-step replica check: passes on identical buffers, raises on every rank when one rank's copy differs
    dc.verified = False
    dc.verify_replicas([res["sharded"][0], torch.arange(5, dtype=torch.int32)])
    dc.verified = False
    bad = res["sharded"][0].clone()
    if rank == world - 1:
        bad[17] += 1e-3
    raised = False
    try:
        dc.verify_replicas([torch.arange(5, dtype=torch.int32), bad])
    except RuntimeError as e:
        raised = "buffers [1]" in str(e)
    dist.barrier()
    dist.destroy_process_group()
    np.savez(out, p_ar=res["all_reduce"][0].numpy(), m_ar=res["all_reduce"][1].numpy(), v_ar=res["all_reduce"][2].numpy(),
             p_sh=res["sharded"][0].numpy(), m_sh=res["sharded"][1].numpy(), v_sh=res["sharded"][2].numpy(),
             covered=np.array(sum(n for _, n in ranges)), raised=np.array(raised))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_adam_protocol_equals_all_reduce_plus_replicated_adam(world, tmp_path):
    """DistContext.start_reduce_scatter / shard_ranges / start_gather_flat / gather_moments ("reduce-scatter + sharded Adam +
    all-gather of updated params"): three Adam steps on a five-tensor flat bucket, each rank updating only its chunk, end with the
    same parameters AND moments, to the bit, as the all-reduce + replicated Adam on every rank; the chunks cover every element
    exactly once; verify_replicas passes on agreeing replicas and raises on every rank when one differs."""
    port = 34500 + os.getpid() % 2000 + world
    outs = [str(tmp_path / f"s{r}.npz") for r in range(world)]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run_sharded, args=(r, world, port, outs[r])) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=250)
        assert p.exitcode == 0
    res = [np.load(o) for o in outs]
    assert sum(int(r["covered"]) for r in res) == 56 * 1001
    for r in res:
        assert bool(r["raised"])
        for k in ("p", "m", "v"):
            np.testing.assert_array_equal(r[k + "_sh"], res[0][k + "_sh"])          # replicas agree
            np.testing.assert_array_equal(r[k + "_sh"], r[k + "_ar"])               # and equal the all-reduce path, bit for bit
    assert np.abs(res[0]["p_sh"]).max() > 0 and np.abs(res[0]["m_sh"]).max() > 0
