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
"""
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

    def __init__(self, rank, world, seed=6666, mode="camera"):
        if mode not in ("camera", "tile-row"):
            raise ValueError(f"unknown shard mode {mode!r}")
        self.rank, self.world, self.seed, self.mode = rank, world, seed, mode
        self.exact_scaling = world >= 1 and (world & (world - 1)) == 0
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
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        self.row_weights = [float(x) + 1.0 for x in total.tolist()]     # +1: an empty row still costs a launch slot
        self.steps_since_rebalance = 0
        return self.rows(n_rows) != before

    def slice_rows(self, P):
        """Rows per rank of the Gaussian-sharded part of a tile-row step: ceil(P / world) rounded up to a whole tile of 32
        (the deformation kernels work on tiles of 32 Gaussians).  Rank r owns rows [r S, min((r + 1) S, P)); the per-Gaussian
        buffers that are gathered hold world x S rows."""
        return ((P + self.world - 1) // self.world + 31) // 32 * 32

    def start_gather(self, tensors, S):
        """Begin the in-place all-gather of every tensor in `tensors` ([world * S, k] rows, contiguous): this rank's rows
        [rank S, (rank + 1) S) are valid going in, all rows coming out of finish()."""
        for t in tensors:
            if not t.is_contiguous() or t.shape[0] != self.world * S:
                raise ValueError("start_gather(): needs contiguous [world * S, k] buffers")
            own = t[self.rank * S:(self.rank + 1) * S]
            if dist.get_backend() == "nccl":
                self._pending.append(dist.all_gather_into_tensor(t, own, async_op=True))
            else:                                   # gloo (tests): list form, no aliasing of input and output
                self._pending.append(dist.all_gather([t[r * S:(r + 1) * S] for r in range(self.world)], own.clone(), async_op=True))

    _OPS = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}

    def start(self, tensor, op="sum"):
        """Begin an in-place all-reduce of `tensor` (a whole, contiguous buffer) without waiting for it."""
        if not tensor.is_contiguous():
            raise ValueError("start() all-reduces in place and needs a contiguous buffer")
        w = dist.all_reduce(tensor, op=self._OPS[op], async_op=True)
        self._pending.append(w)
        return w

    # A collective that cannot complete (a peer died, a rank took another branch) must end the job, not hang it: gloo's wait()
    # honours a timeout and raises; RCCL's wait() only orders streams -- there the process group's own timeout (init_process_group)
    # and its watchdog abort the communicator.  MOM_COLLECTIVE_TIMEOUT_S overrides (seconds).
    TIMEOUT_S = float(__import__("os").environ.get("MOM_COLLECTIVE_TIMEOUT_S", "120"))

    def _wait(self, w):
        if dist.get_backend() == "gloo":
            import datetime
            w.wait(datetime.timedelta(seconds=self.TIMEOUT_S))
        else:
            w.wait()

    def wait_for(self, works):
        """The CURRENT stream waits for these collectives (handles returned by start()); they stay pending for finish(), which
        waits for them again on its own stream (waiting twice is harmless)."""
        for w in works:
            self._wait(w)

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


def attach(trainer, rank, world, seed=6666, mode="camera"):
    trainer.dist = DistContext(rank, world, seed, mode)
    if getattr(trainer, "fused", None) is not None:
        trainer.fused.dist = trainer.dist
    elif mode == "tile-row":
        raise ValueError("tile-row sharding is implemented by the fused step (Trainer(..., fused=True), fine stage, batch_size 1)")
    return trainer.dist
