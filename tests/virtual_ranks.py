"""Virtual ranks of a tile-row shard on ONE GPU: stand-ins for parallel.DistContext that let `world` ranks take the same training
step one after the other on one model, with their collectives resolved between rounds.

A step issues its collectives in a fixed order (all-gather of the deformed state, sum of the compositing record, all-gather of
the position gradients, sum of the deformation gradients, loss sums).  Round k runs every rank with the results of the first k
collectives fed back (as the in-place collective would leave them) and captures what each rank contributes to collective k; the
world then forms that collective's result -- elementwise sum / max, or the concatenation of the ranks' row slices -- and the
next round starts.  After the last collective one more round gives every rank's final state."""
import importlib

import torch


class VRank:
    mode = "tile-row"

    def __init__(self, world_obj, rank):
        self.w, self.rank, self.world = world_obj, rank, world_obj.world
        self.n = 0
        self.split = None if world_obj.split is None else list(world_obj.split)
        self.resplit = world_obj.resplit
        self.row_counts = None

    # ---- what fused_step.py asks of a DistContext in tile-row mode
    def rows(self, n_rows):
        if self.split is not None:
            return self.split[self.rank]
        return importlib.import_module("iclr2025_3d-mom_amd.parallel").split_rows(n_rows, self.world)[self.rank]

    def slice_rows(self, P):
        return ((P + self.world - 1) // self.world + 31) // 32 * 32

    def rebalance_due(self):
        return self.resplit is not None

    def rebalance_rows(self, own_row_counts):
        self.row_counts = own_row_counts.clone()
        before = self.rows(own_row_counts.shape[0])
        self.split, self.resplit = self.resplit, None
        return self.rows(own_row_counts.shape[0]) != before

    def start(self, tensor, op="sum"):
        if tensor.dtype == torch.int32 and tensor.numel() == 1:      # the sticky overflow word
            assert op == "max"
            return
        assert tensor.is_contiguous()
        self._collective(("reduce", op), [tensor])

    def start_gather(self, tensors, S):
        for t in tensors:
            assert t.is_contiguous() and t.shape[0] == self.world * S
        self._collective(("gather", S), list(tensors))

    def finish(self):
        pass

    def wait_for(self, works):
        pass

    def _collective(self, kind, tensors):
        i, self.n = self.n, self.n + 1
        done = self.w.resolved
        if i < len(done):
            assert done[i][0] == kind
            for t, r in zip(tensors, done[i][1]):
                t.copy_(r)
        elif i == len(done):
            self.w.pending[self.rank] = (kind, [t.clone() for t in tensors])
        # collectives behind the first unresolved one work on garbage in this round: ignored


class VirtualWorld:
    def __init__(self, world, split=None, resplit=None):
        self.world, self.split, self.resplit = world, split, resplit
        self.resolved, self.pending, self.history = [], {}, []

    def run(self, step):
        """step(dist) -> anything: one whole step of one rank.  Returns ([result of rank r in the final round], [its VRank])."""
        while True:
            self.pending = {}
            ranks = [VRank(self, r) for r in range(self.world)]
            results = [step(d) for d in ranks]
            if not self.pending:
                return results, ranks
            assert len(self.pending) == self.world, "every rank must take part in every collective"
            kind = self.pending[0][0]
            caps = [self.pending[r][1] for r in range(self.world)]
            assert all(self.pending[r][0] == kind for r in range(self.world))
            if kind[0] == "reduce":
                stack = torch.stack([c[0] for c in caps])
                out = [stack.sum(0) if kind[1] == "sum" else stack.max(0).values]
            else:
                S = kind[1]
                out = []
                for k in range(len(caps[0])):
                    full = torch.empty_like(caps[0][k])
                    for r in range(self.world):
                        full[r * S:(r + 1) * S] = caps[r][k][r * S:(r + 1) * S]
                    out.append(full)
            self.history.append((kind, caps))
            self.resolved.append((kind, out))
