"""Multi-GPU execution of the training step: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X; "gloo" in the CPU tests).

The reference is single-GPU; its only batch axis is `batch_size` cameras per iteration rendered one after the other
(train_4DGS.py:172-229) with the loss averaged over the batch.  CAMERA-BATCH SHARD reproduces exactly that with
one camera per rank: every rank holds the full model (replica), renders and back-propagates its own camera, then

  * parameter gradients: ONE sum-all-reduce over a single flat fp32 bucket (59 floats per Gaussian + the
    deformation field; 59 MB at 200k Gaussians), scaled by 1/world -- identical to the reference's mean over the
    batch.  On the xGMI full mesh RCCL moves S/8 per link per phase, so one large bucket is the right shape;
  * densification statistics: max-all-reduce of the radii, sum-all-reduce of the screen-space gradients
    (train_4DGS.py:203-204,227-229).

The fused training step (fused_step.py) keeps its gradients in two flat buckets from the start, scales the loss
gradient by 1/world at its source and calls start() as soon as a bucket is final: the early bucket (SH, scaling,
rotation, opacity: 56 floats per Gaussian, with the 3 floats of the screen-space gradient behind them) is all-reduced
on RCCL's stream while the deformation backward still runs -- and the appearance parameters' Adam launch follows it on
the step's second stream (DistContext.wait_for) --; only the bucket with xyz and the deformation field (3 floats per
Gaussian + 2.9 M) is exposed.  The radii and the sticky overflow word travel as ONE integer bucket (max).  Three
collectives per step: every torch.distributed call costs the rank's host 40-50 us, and the host paces a rank.
finish() makes the compute stream wait for all of them before Adam.  No packing, no rescaling pass.

Everything downstream (densify / prune / Adam) then runs replicated and stays bit-identical across ranks.  The one
random draw in densify_and_split (gaussian_model.py:525) is made identical by seeding every rank's generator with
(seed, iteration) before the call.

SHARDED ADAM (shard_adam=True, camera mode; SURVEY 8e's second design): the appearance bucket is REDUCE-SCATTERED instead of
all-reduced, every rank runs Adam on its 1/world slice of the five appearance parameters (FusedAdam.step_partial(ranges=...):
Adam is element-wise) and the ranks all-gather the UPDATED PARAMETERS -- the same bytes on the wire as the all-reduce, 1/world
of the Adam pass (1.6 of 11.4 ms per step at 4 M Gaussians).  The parameters live in one flat buffer laid out like the gradient
bucket (FusedStep re-homes them), so both collectives are in place.  The moments of the slices a rank does not own go stale;
before any iteration that reads or restructures the optimizer state (densify / prune / reset / checkpoint) the ranks gather
them (gather_moments), and that iteration itself takes the replicated path.

TRANSPORT.  On RCCL the step's collectives go through libmom4d's own entry points (mom_comm_* in include/mom4d.h: ncclXxx on a
raw stream handle, ordered with stream marks): ~2 us of host time per collective where a torch.distributed call costs 40-50, and
the host paces a rank.  torch.distributed stays the rendezvous (the 128-byte communicator id travels through its store), the
CPU tests' transport (gloo) and the fallback (MOM_COMM=torch).
"""
import os

import torch
import torch.distributed as dist


def split_rows(n_rows, world, weights=None):
    """Contiguous split of `n_rows` (>= 1) tile rows over `world` ranks: [(row0, row1), ...], one entry per rank,
    covering [0, n_rows) in order.  `weights[r]` (e.g. the instance count of tile row r) balances the split by work
    instead of by row count: rank k ends at the first row where the running weight reaches (k+1)/world of the total.
    Every rank gets at least one row when n_rows >= world (a rank never takes rows the ranks after it need); with fewer
    rows than ranks the last ranks get the empty range (n_rows, n_rows) -- never (0, 0), which MomRasterArgs reads as
    "every row"."""
    if n_rows < 1 or world < 1:
        raise ValueError("split_rows: need n_rows >= 1 and world >= 1")
    w = [1.0] * n_rows if weights is None else [max(0.0, float(x)) for x in weights]
    if len(w) != n_rows:
        raise ValueError("split_rows: one weight per row")
    if sum(w) <= 0:
        w = [1.0] * n_rows
    total, out, start, run = sum(w), [], 0, 0.0
    for k in range(world):
        if start >= n_rows:
            out.append((n_rows, n_rows))
            continue
        if k == world - 1:
            end = n_rows
        else:
            target = total * (k + 1) / world
            last = max(start + 1, n_rows - (world - k - 1))        # leave one row for each rank still to come
            end = start
            while end < last and (end == start or run + w[end] <= target + 1e-9 * total):
                run += w[end]
                end += 1
            run = sum(w[:end])
        out.append((start, end))
        start = end
    return out


class DirectComm:
    """The step's collectives on RCCL through libmom4d (csrc/comm.hip), on a communication stream of this object's own.
    A started collective is identified by the mark slot recorded behind it (ops.next_ring_mark): waiting for it is one
    hipStreamWaitEvent on whichever stream needs the result.  Every rank issues the same sequence of calls."""

    def __init__(self, rank, world, device):
        import ctypes as C
        from . import _native as N
        from . import ops
        self.N, self.ops, self.C = N, ops, C
        lib = self.lib = N.lib()
        if not lib.mom_comm_available():
            raise N.MomError("mom_comm: " + lib.mom_comm_last_error().decode())
        idb = (C.c_char * 128)()
        if rank == 0:
            N.check(lib.mom_comm_unique_id(idb), "mom_comm_unique_id")
        box = [bytes(idb.raw)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)            # the rendezvous torch.distributed already has
        idb.raw = box[0]
        self.comm = C.c_void_p()
        with torch.cuda.device(device):
            rc = lib.mom_comm_create(C.byref(self.comm), idb, world, rank)
        if rc != 0:
            raise N.MomError(f"mom_comm_create failed: {lib.mom_comm_last_error().decode()}")
        self.rank, self.world = rank, world
        self.stream_obj = torch.cuda.Stream(device=device)
        self.stream = self.stream_obj.cuda_stream
        self._grouped = False

    _DT = {torch.float32: 0, torch.int32: 1}
    _OP = {"sum": 0, "max": 1}

    def _begin(self):
        """The communication stream starts behind the caller's stream's current tail (the kernels that produced the buffer)."""
        if not self._grouped:
            self.ops.stream_wait_stream(self.stream, self.N.current_stream())

    def _end(self):
        return None if self._grouped else self.ops.next_ring_mark(self.stream)

    def group(self):
        """with comm.group() as g: several collectives submitted as ONE launch (ncclGroupStart / End); g.work is their handle."""
        comm = self

        class _G:
            work = None

            def __enter__(self_g):
                comm.ops.stream_wait_stream(comm.stream, comm.N.current_stream())
                comm.N.check(comm.lib.mom_comm_group_start(), "mom_comm_group_start")
                comm._grouped = True
                return self_g

            def __exit__(self_g, *exc):
                comm._grouped = False
                comm.N.check(comm.lib.mom_comm_group_end(), "mom_comm_group_end")
                self_g.work = comm.ops.next_ring_mark(comm.stream)
        return _G()

    def all_reduce(self, t, op):
        self._begin()
        self.N.check(self.lib.mom_comm_all_reduce(self.comm, t.data_ptr(), t.numel(), self._DT[t.dtype], self._OP[op], self.stream),
                     "mom_comm_all_reduce")
        return self._end()

    def all_gather(self, flat, count_per_rank):
        self._begin()
        self.N.check(self.lib.mom_comm_all_gather(self.comm, flat.data_ptr(), count_per_rank, self._DT[flat.dtype], self.stream),
                     "mom_comm_all_gather")
        return self._end()

    def reduce_scatter(self, flat, count_per_rank, op):
        self._begin()
        self.N.check(self.lib.mom_comm_reduce_scatter(self.comm, flat.data_ptr(), count_per_rank, self._DT[flat.dtype], self._OP[op],
                                                      self.stream), "mom_comm_reduce_scatter")
        return self._end()

    def wait(self, work, stream=None):
        if work is not None:
            self.ops.stream_wait_mark(self.N.current_stream() if stream is None else stream, work)

    def close(self):
        if self.comm:
            self.lib.mom_comm_destroy(self.comm)
            self.comm = None


class DistContext:
    """mode "camera": every rank renders its own camera of the step (the reference's batch axis).
    mode "tile-row": every rank renders the SAME camera, restricted to its tile rows, and runs the deformation field (HexPlane +
    MLP, forward and backward: half of a step) on its own SLICE of the Gaussians.  Exchanges per step: all-gather of the
    deformed state (15 floats per Gaussian), sum of the compositing backward's per-Gaussian record (12 floats), all-gather of
    the position gradients (3 floats), sum of the deformation field's gradients (2.9 M floats); projection, its backward, the
    activations and Adam run replicated on identical inputs and stay bit-identical (fused step only).

    World sizes: the loss gradient (camera mode) and the regularisers (tile-row mode) carry the factor 1/world at their source
    and the all-reduce sums `world` such terms: for a power-of-two world (1, 2, 4, 8: one node) that restores the unsharded
    value EXACTLY; for any other world the replicas still agree with each other bit for bit (every rank receives the same
    reduced buffer) but differ from a single-GPU run by the rounding of x/world.  `exact_scaling` says which case this is.
    The RCCL branch of start_gather() (in-place all_gather_into_tensor, input a view of the output) has only ever run with
    world == 1 (MOM_FORCE_DIST=1): no multi-GPU box is available to this repository's tests; the gloo list form is what
    tests/test_two_process_gpu.py and tests/virtual_ranks.py exercise."""

    def __init__(self, rank, world, seed=6666, mode="camera", shard_adam=False):
        if mode not in ("camera", "tile-row"):
            raise ValueError(f"unknown shard mode {mode!r}")
        self.rank, self.world, self.seed, self.mode = rank, world, seed, mode
        self.exact_scaling = world >= 1 and (world & (world - 1)) == 0
        self.shard_adam = bool(shard_adam) and mode == "camera"
        self.direct = None               # DirectComm on RCCL (connect()); None: torch.distributed carries the collectives
        # gloo's wait() blocks the HOST until the collective is done (RCCL's only orders streams): the fused step then queues its
        # own GPU work first and waits afterwards
        self.host_blocking = dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo"
        self.verified = world == 1       # first-step replica check (verify_replicas) still to run
        self._flat = None
        self._pending = []
        self.row_weights = None          # per-tile-row work estimate agreed by all ranks (rebalance_rows)
        self.steps_since_rebalance = 0

    REBALANCE_EVERY = 16

    def rows(self, n_rows):
        """This rank's tile rows of an image with n_rows rows of tiles: split by the agreed per-row weights when there are
        some for this image height, else by row count."""
        w = self.row_weights if (self.row_weights is not None and len(self.row_weights) == n_rows) else None
        return split_rows(n_rows, self.world, w)[self.rank]

    def rebalance_due(self):
        self.steps_since_rebalance += 1
        return self.steps_since_rebalance >= self.REBALANCE_EVERY

    def rebalance_rows(self, own_row_counts):
        """own_row_counts[r] = instances this rank binned in tile row r, zero for the rows it does not own (float tensor, one
        entry per tile row).  Sums them over the ranks and adopts the result as the weights of the next splits.  Every rank
        gets the same vector, so every rank derives the same split.  Returns True when this rank's rows changed."""
        n_rows = own_row_counts.shape[0]
        before = self.rows(n_rows)
        total = own_row_counts.clone()
        if self.direct is not None:           # one transport per step: two communicators in flight at once is how ranks deadlock
            self.direct.wait(self.direct.all_reduce(total, "sum"))
        else:
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
        self.row_weights = [float(x) + 1.0 for x in total.tolist()]     # +1: an empty row still costs a launch slot
        self.steps_since_rebalance = 0
        return self.rows(n_rows) != before

    def slice_rows(self, P):
        """Rows per rank of the Gaussian-sharded part of a tile-row step: ceil(P / world) rounded up to a whole tile of 32
        (the deformation kernels work on tiles of 32 Gaussians).  Rank r owns rows [r S, min((r + 1) S, P)); the per-Gaussian
        buffers that are gathered hold world x S rows."""
        return ((P + self.world - 1) // self.world + 31) // 32 * 32

    def connect(self, device):
        """Pick the transport of the step's collectives: libmom4d's RCCL entry points when the process group is on RCCL and the
        library can resolve librccl (MOM_COMM=torch keeps torch.distributed).  A failure to set the direct path up is loud but
        not fatal: the torch path computes the same thing."""
        if self.direct is not None or os.environ.get("MOM_COMM", "direct") == "torch":
            return self
        if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"):
            return self
        # two phases, each agreed by ALL ranks before anyone goes on (a rank that took the direct path while another fell back to
        # torch would wait in a collective the other never enters): first "can this rank resolve librccl at all", then "did
        # ncclCommInitRank succeed here" -- a communicator is only created once every rank has said yes to the first
        from . import _native as N
        err = None
        try:
            ok = bool(N.lib().mom_comm_available())
            if not ok:
                err = N.lib().mom_comm_last_error().decode()
        except Exception as e:                                   # noqa: BLE001
            ok, err = False, str(e)
        if not self._all_ranks(ok, device):
            self._warn_fallback(err or "another rank cannot resolve librccl")
            return self
        comm = None
        try:
            comm = DirectComm(self.rank, self.world, device)
        except Exception as e:                                   # noqa: BLE001 -- whatever it was, say so and carry on through torch
            err = str(e)
        if self._all_ranks(comm is not None, device):
            self.direct = comm
        else:
            if comm is not None:
                comm.close()
            self._warn_fallback(err or "another rank could not create its communicator")
        return self

    def _all_ranks(self, flag, device):
        """True iff `flag` is true on every rank (one tiny MIN all-reduce through torch.distributed, at set-up time only)."""
        if self.world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))

    def _warn_fallback(self, why):
        import warnings
        warnings.warn(f"parallel: direct RCCL path unavailable ({why}); the step's collectives go through torch.distributed")

    def group(self):
        """Context manager: the collectives started inside are one launch on the direct path (and nothing special elsewhere)."""
        if self.direct is not None:
            return self.direct.group()
        import contextlib
        return contextlib.nullcontext(type("_NoGroup", (), {"work": None})())

    def start_gather(self, tensors, S):
        """Begin the in-place all-gather of every tensor in `tensors` ([world * S, k] rows, contiguous): this rank's rows
        [rank S, (rank + 1) S) are valid going in, all rows coming out of finish()."""
        for t in tensors:
            if not t.is_contiguous() or t.shape[0] != self.world * S:
                raise ValueError("start_gather(): needs contiguous [world * S, k] buffers")
            own = t[self.rank * S:(self.rank + 1) * S]
            if self.direct is not None:
                self._pending.append(self.direct.all_gather(t, own.numel()))
            elif dist.get_backend() == "nccl":
                self._pending.append(dist.all_gather_into_tensor(t, own, async_op=True))
            else:                                   # gloo (tests): list form, no aliasing of input and output
                self._pending.append(dist.all_gather([t[r * S:(r + 1) * S] for r in range(self.world)], own.clone(), async_op=True))

    _OPS = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}

    def start(self, tensor, op="sum"):
        """Begin an in-place all-reduce of `tensor` (a whole, contiguous buffer) without waiting for it."""
        if not tensor.is_contiguous():
            raise ValueError("start() all-reduces in place and needs a contiguous buffer")
        if self.direct is not None:
            w = self.direct.all_reduce(tensor, op)
        else:
            w = dist.all_reduce(tensor, op=self._OPS[op], async_op=True)
        self._pending.append(w)
        return w

    # ---- sharded Adam: reduce-scatter of a flat bucket, all-gather of a flat parameter buffer (both in place)
    def chunk(self, n):
        """Elements per rank of a flat buffer of n elements cut into `world` equal chunks: ceil(n / world) rounded up to 4
        (16 bytes: the Adam kernel's vector path); the buffer must hold world x chunk elements."""
        return ((n + self.world - 1) // self.world + 3) // 4 * 4

    def start_reduce_scatter(self, flat, chunk, op="sum"):
        """flat: world x chunk elements, contiguous.  Afterwards this rank's chunk [rank chunk, (rank + 1) chunk) holds the
        reduction over the ranks; the other chunks are unspecified.  (gloo has no reduce-scatter, and the torch path is the
        fallback: both all-reduce the whole buffer -- the same values in the own chunk, bit for bit, as every rank sums in
        the same order.)"""
        if not flat.is_contiguous() or flat.numel() != self.world * chunk:
            raise ValueError("start_reduce_scatter(): needs a contiguous buffer of world x chunk elements")
        if self.direct is not None:
            w = self.direct.reduce_scatter(flat, chunk, op)
        else:
            w = dist.all_reduce(flat, op=self._OPS[op], async_op=True)
        self._pending.append(w)
        return w

    def start_gather_flat(self, flat, chunk):
        """flat: world x chunk elements, contiguous, this rank's chunk valid going in, every chunk coming out."""
        if not flat.is_contiguous() or flat.numel() != self.world * chunk:
            raise ValueError("start_gather_flat(): needs a contiguous buffer of world x chunk elements")
        if self.direct is not None:
            w = self.direct.all_gather(flat, chunk)
        elif dist.get_backend() == "nccl":
            w = dist.all_gather_into_tensor(flat, flat[self.rank * chunk:(self.rank + 1) * chunk], async_op=True)
        else:
            w = dist.all_gather([flat[r * chunk:(r + 1) * chunk] for r in range(self.world)],
                                flat[self.rank * chunk:(self.rank + 1) * chunk].clone(), async_op=True)
        self._pending.append(w)
        return w

    def shard_ranges(self, cuts, chunk):
        """cuts: the element offsets [c0, c1, ..., cn] of n tensors laid end to end in a flat buffer.  Returns, per tensor, the
        (first element, count) of its intersection with this rank's chunk, relative to the tensor's own start."""
        lo, hi = self.rank * chunk, (self.rank + 1) * chunk
        out = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            x, y = max(a, lo), min(b, hi)
            out.append((x - a, y - x) if y > x else (0, 0))
        return out

    def gather_moments(self, optimizer, params, cuts, chunk):
        """Sharded Adam leaves every rank with up-to-date moments for its own chunk only.  Before an iteration that reads or
        restructures the optimizer state every rank gets all of them: per moment, own slices -> a flat scratch buffer,
        all-gather in place, flat -> the tensors.  (Two collectives of the parameters' size, once per densification interval.)"""
        if self.world == 1:
            return
        ranges = self.shard_ranges(cuts, chunk)
        dev = params[0].device
        flat = torch.empty(self.world * chunk, dtype=torch.float32, device=dev)
        for key in ("exp_avg", "exp_avg_sq"):
            tensors = [optimizer.state[p][key] for p in params if key in optimizer.state[p]]
            if len(tensors) != len(params):
                return                                     # no state yet: nothing has been sharded
            for t, a, (off, n) in zip(tensors, cuts, ranges):
                if n:
                    flat[a + off:a + off + n].copy_(t.reshape(-1)[off:off + n])
            self.start_gather_flat(flat, chunk)
            self.finish()
            for t, a, b in zip(tensors, cuts[:-1], cuts[1:]):
                t.reshape(-1).copy_(flat[a:b])

    def verify_replicas(self, tensors):
        """Once, on the first step of a world > 1: every rank's copy of what the collectives are supposed to have made identical
        (the reduced buckets, the gathered slab, the gathered parameters) is checksummed and compared across the ranks; a
        mismatch raises on every rank.  The in-place all-gather / reduce-scatter forms have never met a second GPU in this
        repository's tests -- if their first real node gets them wrong, the job must stop, not train on."""
        if self.verified:
            return
        self.verified = True
        sums = []
        for t in tensors:
            v = t.detach().contiguous().reshape(-1).view(torch.int32).to(torch.int64)
            w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) if v.numel() <= (1 << 22) else None
            sums.append(v.sum() if w is None else (v * (w % 8191 + 1)).sum())
        c = torch.stack(sums)
        both = torch.cat((c, -c))
        dist.all_reduce(both, op=dist.ReduceOp.MAX)               # max(c) and max(-c) = -min(c): equal on every rank iff all agree
        n = c.numel()
        bad = (both[:n] != -both[n:]).nonzero().flatten().tolist()
        if bad:
            raise RuntimeError(f"parallel: replicas differ after the first step's collectives (buffers {bad} of {n}; rank {self.rank} of "
                               f"{self.world}, transport {'direct RCCL' if self.direct is not None else dist.get_backend()})")

    # A collective that cannot complete (a peer died, a rank took another branch) must end the job, not hang it: gloo's wait()
    # honours a timeout and raises; RCCL's wait() only orders streams -- there the process group's own timeout (init_process_group)
    # and its watchdog abort the communicator.  MOM_COLLECTIVE_TIMEOUT_S overrides (seconds).
    TIMEOUT_S = float(__import__("os").environ.get("MOM_COLLECTIVE_TIMEOUT_S", "120"))

    def _wait(self, w, stream=None):
        if self.direct is not None:
            self.direct.wait(w, stream)
        elif dist.get_backend() == "gloo":
            import datetime
            w.wait(datetime.timedelta(seconds=self.TIMEOUT_S))
        else:
            w.wait()

    def wait_for(self, works, stream=None):
        """The CURRENT stream (or the raw handle `stream`: direct path) waits for these collectives (handles returned by
        start()); they stay pending for finish(), which waits for them again on its own stream (waiting twice is harmless)."""
        if stream is not None and self.direct is None:
            raise ValueError("wait_for(stream=...) needs the direct path; torch.distributed orders the current torch stream")
        for w in works:
            self._wait(w, stream)

    def finish(self):
        """Every all-reduce begun with start() is complete for work issued after this returns (on a GPU the current
        stream waits for RCCL's; the host does not block)."""
        for w in self._pending:
            self._wait(w)
        self._pending.clear()

    def sync_param_grads(self, optimizer):
        """Average the gradients of every optimised parameter across ranks through one flat bucket."""
        ps = [p for g in optimizer.param_groups for p in g["params"]]
        if not ps:
            return
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        if self._flat is None or self._flat.numel() != n or self._flat.device != dev:
            self._flat = torch.empty(n, dtype=torch.float32, device=dev)
        flat, off = self._flat, 0
        for p in ps:
            k = p.numel()
            seg = flat[off:off + k]
            if p.grad is None:
                seg.zero_()          # ranks must agree on the bucket layout even if a rank has no grad for p
            else:
                # storage-order copy keeps channel-last planes cheap
                seg.view(p.grad.shape if p.grad.is_contiguous() else (-1,)).copy_(
                    p.grad if p.grad.is_contiguous() else p.grad.permute(*_storage_order(p.grad)).reshape(-1))
            off += k
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / self.world)
        off = 0
        for p in ps:
            k = p.numel()
            if p.grad is not None:
                seg = flat[off:off + k]
                if p.grad.is_contiguous():
                    p.grad.copy_(seg.view(p.grad.shape))
                else:
                    order = _storage_order(p.grad)
                    p.grad.permute(*order).copy_(seg.view([p.grad.shape[i] for i in order]))
            off += k

    def sync_stats(self, radii, vsp_grad):
        dist.all_reduce(radii, op=dist.ReduceOp.MAX)
        dist.all_reduce(vsp_grad, op=dist.ReduceOp.SUM)
        vsp_grad.mul_(1.0 / self.world)
        return radii, radii > 0, vsp_grad

    def seed_for(self, iteration):
        """Same random stream on every rank for the densification draw of this iteration."""
        torch.manual_seed(self.seed * 1_000_003 + iteration)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(self.seed * 1_000_003 + iteration)


def _storage_order(t):
    """Dimension order that walks t's memory contiguously (largest stride first)."""
    return sorted(range(t.dim()), key=lambda i: (-t.stride(i), i))


def attach(trainer, rank, world, seed=6666, mode="camera", shard_adam=None):
    """shard_adam (camera mode, fused step): reduce-scatter + 1/world of Adam + all-gather of the parameters instead of an
    all-reduce and a replicated Adam (module docstring); None reads MOM_SHARD_ADAM (default off)."""
    if shard_adam is None:
        shard_adam = os.environ.get("MOM_SHARD_ADAM", "0") == "1"
    fused = getattr(trainer, "fused", None)
    if shard_adam and (fused is None or mode != "camera"):
        raise ValueError("shard_adam needs the fused step and the camera-batch shard")
    trainer.dist = DistContext(rank, world, seed, mode, shard_adam=shard_adam)
    dev = trainer.g._xyz.device
    if dev.type == "cuda":
        trainer.dist.connect(dev)
    if fused is not None:
        fused.dist = trainer.dist
    elif mode == "tile-row":
        raise ValueError("tile-row sharding is implemented by the fused step (Trainer(..., fused=True), fine stage, batch_size 1)")
    return trainer.dist
