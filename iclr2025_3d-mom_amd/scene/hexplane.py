"""HexPlaneField with the reference's constructor, parameter names and numerics
(reference scene/hexplane.py:109-183), whose forward/backward run as two fused HIP kernels
(csrc/hexplane.hip) instead of 12 grid_sample launches per direction.

Planes keep the reference's logical shape [1, C, H, W] (regularisers, state_dict and the optimizer see
exactly what they see in the reference) but live in memory channel-last, so one texel is one 128-byte line."""
import itertools
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .. import ops


def normalize_aabb(pts, aabb):
    # hexplane.py:19-20, incl. the reference's flipped aabb rows (row 0 = xyz_max)
    return (pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0


def init_grid_param(grid_nd: int, in_dim: int, out_dim: int, reso: Sequence[int], a: float = 0.1, b: float = 0.5):
    """hexplane.py:48-70: one plane per coordinate pair; space planes U(a,b), space-time planes 1.
    Random numbers are drawn into a contiguous NCHW tensor (same RNG stream as the reference) and copied into the
    channel-last storage."""
    assert in_dim == len(reso) and grid_nd <= in_dim
    planes = nn.ParameterList()
    for comb in itertools.combinations(range(in_dim), grid_nd):
        shape = [1, out_dim] + [reso[c] for c in comb[::-1]]
        init = torch.empty(shape)
        if in_dim == 4 and 3 in comb:
            nn.init.ones_(init)
        else:
            nn.init.uniform_(init, a=a, b=b)
        if grid_nd == 2:
            p = ops.make_plane(out_dim, shape[2], shape[3])
            p.copy_(init)
        else:
            p = init
        planes.append(nn.Parameter(p))
    return planes


class HexPlaneField(nn.Module):
    def __init__(self, bounds, planeconfig, multires) -> None:
        super().__init__()
        aabb = torch.tensor([[bounds, bounds, bounds], [-bounds, -bounds, -bounds]])
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = multires
        self.concat_features = True
        self.grids = nn.ModuleList()
        self.feat_dim = 0
        self._order, self._order_age = None, 0   # cached spatial processing order (speed only, never a result)
        for res in self.multiscale_res_multipliers:
            config = self.grid_config[0].copy()
            # multi-resolution on the three space axes only (hexplane.py:131-134)
            config["resolution"] = [r * res for r in config["resolution"][:3]] + config["resolution"][3:]
            gp = init_grid_param(grid_nd=config["grid_dimensions"], in_dim=config["input_coordinate_dim"],
                                 out_dim=config["output_coordinate_dim"], reso=config["resolution"])
            self.feat_dim = self.feat_dim + gp[-1].shape[1] if self.concat_features else gp[-1].shape[1]
            self.grids.append(gp)
        print("feature_dim:", self.feat_dim)

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def aabb_host(self):
        """The six aabb floats as a python list, read back from the device only when the parameter was replaced
        (set_aabb) or written in place (load_state_dict) since the last call -- not once per iteration."""
        a = self.aabb
        key = (id(a), a._version, a.data_ptr())
        if getattr(self, "_aabb_host_key", None) != key:
            self._aabb_host_val = a.detach().float().cpu().reshape(-1).tolist()
            self._aabb_host_key = key
        return self._aabb_host_val

    def set_aabb(self, xyz_max, xyz_min):
        import numpy as np
        aabb = torch.tensor(np.asarray([np.asarray(xyz_max, np.float32), np.asarray(xyz_min, np.float32)]), dtype=torch.float32)
        self.aabb = nn.Parameter(aabb.to(self.aabb.device), requires_grad=False)
        print("Voxel Plane: set aabb=", self.aabb)

    def get_density(self, pts: torch.Tensor, timestamps=None):
        """[N,3] points (+ [N,1] timestamps, or one python float for all points) -> [N, feat_dim]."""
        pts = pts.reshape(-1, pts.shape[-1])
        levels = [list(g) for g in self.grids]
        order = self._processing_order(pts)
        kw = {}
        if hasattr(ops.BACKEND, "hexplane_orders"):
            kw["plane_orders"] = self._plane_orders(pts)
        return ops.BACKEND.hexplane_features(pts, timestamps, self.aabb, levels, order=order,
                                             aabb_host=self.aabb_host() if self.aabb.is_cuda else None, **kw)

    REORDER_EVERY = int(__import__("os").environ.get("MOM_REORDER_EVERY", "64"))

    # The refresh -- one Morton sort and six plane sorts, 0.8 ms of GPU time at 200 k points -- is PREFETCHED when the caller has a
    # second stream to give it: REFRESH_AHEAD calls before it is due, _processing_order() notes that it is due soon, and the caller
    # (the fused training step, right after it has queued its gradient-bucket clearing and the plane regularisers on its second
    # stream) calls prefetch_if_due(): the sorts run there into fresh buffers, beside the step's forward and backward (they read the
    # positions while Adam may be writing them: an order is a permutation whatever the keys were, and only speed depends on it), and
    # when the refresh is due the new orders are swapped in behind a mark.  On the step's own stream it was an 0.8 ms stall every 64
    # iterations, 1.4 % of the step (tools/probe/per_camera.py: the 1.6-1.7 ms steps).  The step's OWN second stream, not a third
    # one: the device has four hardware queues, and a fifth stream in the process collided with the two of the forward-only render
    # pool (5150 -> 3800 frames/s).  Callers without a second stream (render(), the autograd path) get the refresh on their own
    # stream, as before.  MOM_ASYNC_ORDERS=0 switches the prefetch off.
    REFRESH_AHEAD = 8
    ASYNC_REFRESH = __import__("os").environ.get("MOM_ASYNC_ORDERS", "1") != "0"

    def _order_key(self, pts):
        return (pts.data_ptr(), pts.shape[0], pts.device, tuple(self.aabb_host()) if pts.is_cuda else None)

    def prefetch_if_due(self, pts, stream):
        """Launch the refresh on `stream` (a raw handle of the caller's second stream) if _processing_order() found it due soon."""
        if getattr(self, "_prefetch_due", False):
            self._prefetch_due = False
            if self.ASYNC_REFRESH and getattr(self, "_pending", None) is None and getattr(self, "_porders", None) is not None:
                self._prefetch_orders(pts, stream)

    def _prefetch_orders(self, pts, side):
        """The refresh on the stream `side`; the result waits in self._pending = (key, order, plane orders, kept buffers)."""
        cur = ops.N.current_stream()
        # the outputs and scratch come from the caller's stream's pool (their memory may have been in use there a moment ago), so
        # the second stream starts behind the caller's stream's current tail
        ops.stream_wait_stream(side, cur)
        keep = []
        order = ops.BACKEND.morton_order(pts, stream=side, keepalive=keep)
        porders = ops.BACKEND.hexplane_orders(pts, [list(g) for g in self.grids], self.aabb, aabb_host=self.aabb_host(), stream=side,
                                              keepalive=keep)
        # no process-wide mark slot: several fields (or fused steps) of one device prefetch on different second streams, and a
        # shared slot re-recorded by another field would make this one wait for the wrong stream -- the stream handle itself is kept,
        # and whoever takes or drops the result waits for that stream's tail (once per REORDER_EVERY calls)
        self._pending = (self._order_key(pts), order, porders, keep, side)

    def _drop_pending(self):
        """Forget a prefetched refresh that no longer fits (the model was restructured under it).  Its buffers go back to the
        allocator, which may hand them out again for work on the caller's stream: that stream is first put behind the second
        stream's kernels, which may still be writing them."""
        if getattr(self, "_pending", None) is not None:
            ops.stream_wait_stream(ops.N.current_stream(), self._pending[4])
            self._pending = None

    def _processing_order(self, pts):
        """Morton order of the points, refreshed when their number changes (densify / prune) and every
        REORDER_EVERY calls (positions drift slowly).  Only the speed of the fused kernels depends on it."""
        if not hasattr(ops.BACKEND, "morton_order") or pts.shape[0] == 0:
            return None
        self._porders_swapped = False
        pending = getattr(self, "_pending", None)
        if self._order is None or self._order.shape[0] != pts.shape[0] or self._order.device != pts.device:
            self._drop_pending()
            self._order = ops.BACKEND.morton_order(pts)
            self._order_age = 0
        elif self._order_age >= self.REORDER_EVERY:
            if pending is not None and pending[0] == self._order_key(pts):
                # the prefetched orders: this stream waits for the tail of the stream that sorted them (long past, normally)
                ops.stream_wait_stream(ops.N.current_stream(), pending[4])
                self._order, self._porders, self._porders_key = pending[1], pending[2], tuple(self.aabb_host())
                self._porders_swapped = True
                self._pending = None
            else:
                self._drop_pending()
                self._order = ops.BACKEND.morton_order(pts)
            self._order_age = 0
        elif pts.is_cuda and pending is None and ops.BACKEND.name == "hip" and self._order_age == self.REORDER_EVERY - self.REFRESH_AHEAD:
            self._prefetch_due = True        # (a caller with a second stream picks it up: prefetch_if_due)
        self._order_age += 1
        return self._order

    def _plane_orders(self, pts):
        """Per-space-plane orders of the two-pass backward (ops.hexplane_orders), rebuilt together with the Morton order:
        call right after _processing_order()."""
        if not hasattr(ops.BACKEND, "hexplane_orders") or pts.shape[0] == 0 or not pts.is_cuda:
            return None
        po = getattr(self, "_porders", None)
        if po is None or po[0].shape[-1] != pts.shape[0] or po[0].device != pts.device \
                or (self._order_age == 1 and not getattr(self, "_porders_swapped", False)) \
                or getattr(self, "_porders_key", None) != tuple(self.aabb_host()):
            self._porders = ops.BACKEND.hexplane_orders(pts, [list(g) for g in self.grids], self.aabb, aabb_host=self.aabb_host())
            self._porders_key = tuple(self.aabb_host())
        return self._porders

    def _slice_order(self, pts, g0, g1, bump=True):
        """Morton order of the points [g0, g1) (positions relative to g0): what a rank of a tile-row shard, which runs the
        field on its slice of the Gaussians only, hands to the kernels.  Cached like the whole-cloud order: it ages once per
        step -- the forward's call counts, the backward's (bump=False) takes the forward's order as it is, so a rebuild never
        lands between the two halves of one step -- and the key carries the point tensor's storage and length, which densify /
        prune replace even when the slice bounds happen to stay."""
        if not hasattr(ops.BACKEND, "morton_order") or g1 <= g0:
            return None
        c = getattr(self, "_sl_order", None)
        key = (g0, g1, pts.device, pts.data_ptr(), pts.shape[0])
        if c is None or c[0] != key or (bump and c[2] >= self.REORDER_EVERY):
            c = self._sl_order = [key, ops.BACKEND.morton_order(pts[g0:g1]), 0]
            self._sl_porders = None
        if bump:
            c[2] += 1
        return c[1]

    def _slice_plane_orders(self, pts, g0, g1):
        """Per-space-plane orders of the slice [g0, g1) for the two-pass backward; rebuilt with _slice_order (call it first)."""
        if not hasattr(ops.BACKEND, "hexplane_orders") or g1 <= g0 or not pts.is_cuda:
            return None
        key = (g0, g1, pts.data_ptr(), pts.shape[0], tuple(self.aabb_host()))
        po = getattr(self, "_sl_porders", None)
        if po is None or po[0] != key:
            po = self._sl_porders = (key, ops.BACKEND.hexplane_orders(pts[g0:g1], [list(g) for g in self.grids], self.aabb,
                                                                     aabb_host=self.aabb_host()))
        return po[1]

    def forward(self, pts: torch.Tensor, timestamps=None):
        return self.get_density(pts, timestamps)
