"""One fine-stage training iteration as an explicit sequence of ~25 libmom4d launches -- no autograd graph, no
per-op tensor allocation, no host synchronisation.

It computes exactly what `train.Trainer.step` computes through `render()` + `loss.backward()` (the reference's
train_4DGS.py:149-297 with batch_size 1, stage "fine", L1 (+ lambda_dssim SSIM) loss, the shipped deformation config) and leaves the
same `.grad` tensors behind, so the optimizer step, the densification statistics and everything downstream are shared
with the autograd path.  `tests/test_fused_step_gpu.py` checks the two paths against each other.

Order of launches: hexplane_fwd -> deform_fwd -> activations_fwd -> preprocess / binning / sort / render_fwd -> l1 ->
render_bwd / preprocess_bwd -> activations_bwd -> deform_bwd (dx, dw) -> hexplane_bwd -> plane_reg -> adam.
"""
import ctypes as C
import math
import os

import torch

from . import _native as N
from . import ops
from .diff_gaussian_rasterization import _C as RC

# The L1 epilogue of the compositing forward leaves one pair of sums per tile (MomRasterArgs.l1_partials) instead of adding 2040
# workgroups' pairs into one line; MOM_L1_PARTIALS=0 is the A/B switch (render_fwd 135 -> 127 us with HIP events around it).
L1_PARTIALS = os.environ.get("MOM_L1_PARTIALS", "1") != "0"


class FusedStep:
    def __init__(self, gaussians, opt, hyper, background):
        self.g, self.opt, self.hyper, self.bg = gaussians, opt, hyper, background
        dn = gaussians._deformation.deformation_net
        if not dn._fusable():
            raise N.MomError("FusedStep needs the shipped deformation configuration (W=64, D=0, no_do, no_dshs)")
        self.P = -1
        self.lib = N.lib()
        self.last = {}
        self.dist = None      # parallel.DistContext (camera-batch shard): set by parallel.attach()

    # ------------------------------------------------------------------ buffers (re-made when P changes)
    def _rows(self, P):
        """Rows of the per-Gaussian buffers a tile-row shard gathers: world x S (DistContext.slice_rows), else P."""
        dc = self.dist
        return dc.world * dc.slice_rows(P) if (dc is not None and dc.mode == "tile-row") else P

    def _ensure(self, P, W, H, dev):
        """The step's buffers.  Per-Gaussian storage is CAPACITY based: a densify / prune round changes P by a few percent every
        hundred iterations (train_4DGS.py:264-290), and re-making fifty buffers each time sent the step after every round back to
        the driver for fresh memory (tools/probe/c5_leg.py: the first round of a process cost 148 ms instead of 13 and the steps
        behind it ran at 59 instead of 94 per second until the new pages had been touched).  Storage is re-made only when the
        model outgrows it (then with a quarter of headroom) or shrinks below half of it; otherwise the attributes below are
        re-sliced views of the same memory."""
        pad = self._rows(P)
        key = (P, W, H, pad, dev)
        if key == getattr(self, "_key", None):
            return
        self._key = key
        self.P, self._wh, self._pad = P, (W, H), pad
        f = dict(dtype=torch.float32, device=dev)
        e = lambda *s: torch.empty(*s, **f)
        cap = getattr(self, "_rows_cap", 0)
        same_frame = getattr(self, "_store_for", None) == (W, H, dev)
        if not same_frame:
            self._store_for = (W, H, dev)
            self.color, self.depth, self.dimg = e(3, H, W), e(1, H, W), e(3, H, W)
            self.img = torch.empty(self.lib.mom_raster_image_bytes(W, H), dtype=torch.uint8, device=dev)
            self.nr_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self.nr_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            # Sticky overflow word (mom_raster_forward_render only ever sets bits in it).  The binning buffer is sized from
            # earlier frames without waiting for this frame's count; when a frame does not fit, its image and gradients are
            # truncated, the word becomes nonzero and stays so, and Adam / the densification statistics of that step and of every
            # later one are no-ops ON THE DEVICE (mom_adam_step / mom_densify_stats skip_if_nonzero) until the host -- which
            # runs several steps ahead and reads the word through flag_ring a few steps later -- clears it and replays the
            # skipped iterations with an exactly sized buffer (train.Trainer._recover).  Nothing truncated ever reaches the model.
            # (the word itself lives behind the radii, below: one integer bucket for a camera-batch shard's max-all-reduce)
            self.flag_ring = torch.zeros(self.RING, dtype=torch.int32).pin_memory()
            self.next_tag = 1
            # loss accumulators live in spare words of the image scratch's header, which the rasterizer forward clears at the
            # start of every step together with its tile counters: no memset of their own (mom_l1_loss_acc / mom_plane_regulation_acc)
            hdr_f = self.img[(-self.img.data_ptr()) % 256:][:256].view(torch.float32)
            self.sums, self.regval = hdr_f[8:10], hdr_f[10:11]
            # the compositing forward's L1 epilogue leaves one pair of sums per TILE here (MomRasterArgs.l1_partials) instead of 2040
            # workgroups adding into self.sums: added up only when somebody reads the loss (LazyLoss / _TileSums)
            self.l1_part = torch.empty(((W + 15) // 16) * ((H + 15) // 16), 2, dtype=torch.float32, device=dev)
            self.ssim_dm = None                  # SSIM term: made on first use (lambda_dssim may be switched on later)
            self.binning = None
        if not same_frame or pad > cap or 2 * pad < cap:
            grew = same_frame and cap and pad > cap
            cap = self._rows_cap = pad + pad // 4 if grew else pad
            st = self._store = {name: e(cap, cols) for name, cols, _ in self._ROW_BUFFERS}
            st["radii"] = torch.zeros(cap + 1, dtype=torch.int32, device=dev)      # + the sticky overflow word behind the P radii
            st["early"] = e(59 * cap + 4 * self._world())                          # + the screen-space gradients behind the 56 P (sharded Adam: behind world x chunk)
            st["pflat"] = None                                                     # sharded Adam: the appearance parameters' flat home (_rehome)
            st["loc"] = (e(cap, 3), e(cap, 4))
            self.geom = torch.empty(self.lib.mom_raster_geom_bytes(cap), dtype=torch.uint8, device=dev)
            self.dh_scratch = torch.empty(self.lib.mom_deform_backward_scratch_bytes(cap), dtype=torch.uint8, device=dev)
        # the step after a change of P sizes its binning buffer from its own count (one sync); the buffer itself is kept if it fits
        self.cap = 0
        self._resize_next = False
        st = self._store
        # (the gathered buffers of a tile-row shard carry world x S >= P rows; every kernel reads the first P)
        for name, _, padded in self._ROW_BUFFERS:
            setattr(self, name, st[name][:pad if padded else P])
        # [radii (P) | overflow word]: contiguous, so that a camera-batch shard agrees on both with ONE max-all-reduce.  The word is
        # sticky across steps: its value moves with it when P (and so its position) changes
        old_flags = getattr(self, "flags", None)
        self.ibucket = st["radii"][:P + 1]
        self.radii, self.flags = self.ibucket[:P], self.ibucket[P:]
        if old_flags is not None and old_flags.device == self.flags.device and old_flags.data_ptr() != self.flags.data_ptr():
            self.flags.copy_(old_flags)
        elif old_flags is None or old_flags.device != self.flags.device:
            self.flags.zero_()
        # parameter gradients (persist across steps; .grad points at them).  They live in two flat buckets so that a
        # multi-GPU run all-reduces them in place, without packing: `early` (final once the activation backward has run:
        # SH, scaling, rotation, opacity = 56 floats per Gaussian) and `late` (xyz + the deformation field, final only
        # after the HexPlane backward; made in _deform_grads).
        # (behind the 56 P: the screen-space gradients, 3 P -- the densification statistics' input, summed over a camera-batch
        # shard's ranks in the same all-reduce)
        # sharded Adam (DistContext.shard_adam): the 56 P appearance gradients are cut into world chunks of `_chunk` elements for the
        # reduce-scatter, so the bucket's appearance part is padded to world x chunk and the screen-space gradients sit behind THAT
        dc = self.dist
        self._chunk = dc.chunk(56 * P) if (dc is not None and getattr(dc, "shard_adam", False)) else 0
        app = self._world() * self._chunk if self._chunk else 56 * P
        self.early_bucket = st["early"][:app + 3 * P]
        self.app_flat = self.early_bucket[:app]
        self.early = self.early_bucket[:56 * P]
        self.g2d_flat = self.early_bucket[app:]
        self.g2d = self.g2d_flat.view(P, 3)
        self._cut = cut = [0, 3 * P, 48 * P, 51 * P, 55 * P, 56 * P]
        seg = lambda i: self.early[cut[i]:cut[i + 1]]
        self.gdc, self.grest = seg(0).view(P, 1, 3), seg(1).view(P, 15, 3)
        self.gsc, self.grot, self.gop = seg(2).view(P, 3), seg(3).view(P, 4), seg(4).view(P, 1)
        self._loc = (st["loc"][0][:P], st["loc"][1][:P])

    # per-Gaussian float buffers: (attribute, floats per row, sized for a tile-row shard's padded row count)
    _ROW_BUFFERS = (("feat", 64, False), ("a0", 64, False), ("dfeat", 64, False), ("pts", 3, True), ("sc_d", 3, False),
                    ("rot_d", 4, True), ("sc", 3, True), ("rot", 4, True), ("op", 1, True), ("gcol", 3, False),
                    ("gcov", 6, False))

    def _world(self):
        return self.dist.world if self.dist is not None else 1

    _APP_SHAPES = ((1, 3), (15, 3), (3,), (4,), (1,))

    def _app_params(self):
        g = self.g
        return [g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity]

    def _rehome(self):
        """Sharded Adam: the five appearance parameters live in ONE flat buffer laid out like their gradient bucket
        ([f_dc 3P | f_rest 45P | scaling 3P | rotation 4P | opacity P], padded to world x chunk), so that the all-gather of the
        updated parameters is in place.  A parameter that is not there (first step, after a densify / prune round replaced it) is
        copied in and its .data pointed at its slice; the Parameter objects -- and with them the optimizer's state -- stay."""
        P, st = self.P, self._store
        need = self._world() * self._chunk
        flat = st.get("pflat")
        if flat is None or flat.numel() < need or flat.device != self.early.device:
            flat = st["pflat"] = torch.empty(56 * self._rows_cap + 4 * self._world(), dtype=torch.float32, device=self.early.device)
        self.pflat = flat[:need]
        for p, a, b, shp in zip(self._app_params(), self._cut[:-1], self._cut[1:], self._APP_SHAPES):
            v = self.pflat[a:b].view((P,) + shp)
            if p.data_ptr() != v.data_ptr():
                v.copy_(p.data)
                p.data = v

    def gather_moments(self):
        """Before an iteration that reads or restructures the optimizer state: every rank gets the moments of every slice (sharded
        Adam keeps only its own up to date)."""
        dc = self.dist
        if dc is None or not getattr(dc, "shard_adam", False) or self.P <= 0 or not self._chunk:
            return
        if self.g._xyz.shape[0] != self.P:
            return                       # the model was restructured since the last step: its state is whole (the round gathered first)
        dc.gather_moments(self.g.optimizer, self._app_params(), self._cut, self._chunk)

    RING = 64
    OVERLAP_DW = os.environ.get("MOM_OVERLAP_DW", "1") != "0"     # the MLP weight-gradient kernel on a second stream, beside the HexPlane backward
    EARLY_ADAM = os.environ.get("MOM_EARLY_ADAM", "1") != "0"     # the appearance parameters' Adam on that stream, beside the MLP backward
    side = None
    keep_all_tiles = False            # True: bin whole rectangles like the reference (MomRasterArgs.keep_all_tiles; measurement only)
    HEADROOM, MARGIN = 1.5, 65536     # binning capacity = HEADROOM x an earlier frame's instance count + MARGIN

    def exact_next(self):
        """Size the binning buffer of the next step from that step's own instance count (one host sync): it cannot overflow."""
        self._resize_next = True

    def post_flag(self, slot):
        """Copy the overflow word to ring slot `slot` behind everything enqueued so far; the returned event tells when the
        slot is valid.  The word holds 0 or the tag of the FIRST step that overflowed since it was cleared."""
        self.flag_ring[slot:slot + 1].copy_(self.flags, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def _hex_scratch_for(self, hp, rows, dev):
        """Scratch of the two-pass HexPlane backward for `rows` points (common-factor rows + the time lines): kept while it is
        large enough and not more than twice too large."""
        need = self.lib.mom_hexplane_backward_scratch_bytes(C.byref(hp), rows)
        t = getattr(self, "_hex_scratch", None)
        if t is None or t.device != dev or t.numel() < need or t.numel() > 2 * need + (1 << 20):
            grow = t is not None and t.device == dev and t.numel() < need
            self._hex_scratch = torch.empty(need + need // 4 if grow else need, dtype=torch.uint8, device=dev)

    def _gacc_view(self, P, W, H):
        """The per-Gaussian record of the compositing backward inside the geometry scratch, as a flat fp32 tensor."""
        lay = N.MomRasterLayout()
        self.lib.mom_raster_layout(P, W, H, 0, C.byref(lay))
        base = self.geom[(-self.geom.data_ptr()) % 256:]
        return base[lay.geom_gacc:lay.geom_gacc + P * 4 * N.GACC_FLOATS].view(torch.float32)

    def _deform_grads(self):
        """Flat zero-able gradient storage for the deformation field (planes channel-last + live MLP tensors)."""
        dn = self.g._deformation.deformation_net
        planes = [p for lv in dn.grid.grids for p in lv]
        mlp = dn._fused_params()
        key = tuple(p.data_ptr() for p in planes + mlp) + (self.P, self._pad, self._rows_cap)
        if getattr(self, "_dg_key", None) != key:
            n = self._dg_n = sum(p.numel() for p in planes + mlp)
            store = getattr(self, "_dg_store", None)
            if store is None or store.numel() != 4 + n + 3 * self._rows_cap or store.device != planes[0].device:
                store = self._dg_store = torch.zeros(4 + n + 3 * self._rows_cap, dtype=torch.float32, device=planes[0].device)
            # [loss sums (4) | field gradients (n) | xyz gradients (3 pad)]: the four floats in front carry a tile-row shard's loss
            # sums (sum |d|, sum d^2, sum of the SSIM map over the rank's own rows) into the SAME all-reduce as the field
            # gradients -- a collective of their own cost the step ~20 us of stream hand-over and four small launches
            self._dg_all = store[:4 + n + 3 * self._pad]
            self._tr_sums = self._dg_all[:4]
            self._dg_flat = self._dg_all[4:]                                                                # the `late` bucket
            self.gxyz_rows = self._dg_flat[n:].view(self._pad, 3)
            self.gxyz = self.gxyz_rows[:self.P]
            off, self._dg_planes, self._dg_mlp = 0, [], []
            for p in planes:       # same channel-last strides as the parameter
                st = ops.plane_storage(p)
                v = self._dg_flat[off:off + p.numel()].view(st.shape).permute(2, 0, 1).unsqueeze(0)
                self._dg_planes.append(v)
                off += p.numel()
            for p in mlp:
                self._dg_mlp.append(self._dg_flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self._dg_key = key
        return planes, mlp

    # ------------------------------------------------------------------ one iteration (forward + backward)
    def forward_backward(self, cam, delta_scale=1, early_adam=None):
        """early_adam: callable(list of parameters) or None -- see EARLY_ADAM below; the caller promises that its optimizer step
        for this iteration follows with nothing in between that reads or replaces those parameters."""
        g, lib, s = self.g, self.lib, N.current_stream()
        dev = g._xyz.device
        P = g._xyz.shape[0]
        W, H = int(cam.image_width), int(cam.image_height)
        self._ensure(P, W, H, dev)
        sharded = bool(self._chunk) and early_adam is not None and self.EARLY_ADAM     # this iteration takes the sharded-Adam path
        if self._chunk:
            self._rehome()
        if self._resize_next:                   # after a tile-row re-split, an overflow, or on request (exact_next)
            self.cap, self._resize_next = 0, False
        view, proj, campos, gt = cam.device_tensors(dev)
        dn = g._deformation.deformation_net
        field = dn.grid
        planes, mlp = self._deform_grads()
        xyz, scal, rot, opac = g._xyz.detach(), g._scaling.detach(), g._rotation.detach(), g._opacity.detach()
        flow = g._scene_flow if g._scene_flow.is_contiguous() else g._scene_flow.contiguous()
        for t in (xyz, scal, rot, opac, g._features_dc, g._features_rest):
            assert t.is_contiguous()
        time = float(cam.time)
        order = field._processing_order(xyz)
        optr = None if order is None else order.data_ptr()
        porders = field._plane_orders(xyz)           # per-space-plane orders of the two-pass HexPlane backward

        # ---- deformation field
        # the descriptors only hold pointers and shapes: rebuilt when a parameter or the aabb moves, not every iteration
        dkey = (self._dg_key, tuple(field.aabb_host()))
        if getattr(self, "_desc_key", None) != dkey:
            levels = [list(lv) for lv in field.grids]
            gl, k = [], 0
            for lv in levels:
                gl.append(self._dg_planes[k:k + 6])
                k += 6
            hp, keep = ops._hexplane_desc([[p.detach() for p in lv] for lv in levels], field.aabb, gl, aabb_host=field.aabb_host())
            md = ops.DeformMLPFunction._desc([p.detach() for p in mlp], self._dg_mlp)
            self._desc = (hp, keep, md)
            self._desc_key, self._reg_arr = dkey, None
        hp, keep, md = self._desc
        coef = float(delta_scale * cam.frame_num)
        # ---- second stream, first job: clear the deformation field's gradient bucket and run the plane regularisers (value +
        # gradient; they depend on the planes only, so they need not wait for the backward) while this stream renders.  The
        # backward's kernels then ADD into what the regulariser left; this stream waits for the event before its first write
        # to the bucket (the projection backward's d xyz).
        dc = self.dist
        reg_scale = 1.0 / dc.world if dc is not None else 1.0
        # each on its own check: `side` is a public attribute a caller may have set
        if self.side is None:
            self.side = torch.cuda.Stream(device=dev)
        if getattr(self, "regacc", None) is None:
            self.regacc = torch.zeros(1, dtype=torch.float32, device=dev)
        hy = self.hyper
        reg = None
        # (everything on the second stream goes there through its raw handle -- libmom4d's launches, mom_zero_async, the ordering
        # calls of csrc/stream_order.hip: entering a torch stream context, a wait_stream and an event cost ~10 us of host time each,
        # and at config 1 the host paces the step)
        side = self.side.cuda_stream
        ops.stream_wait_stream(side, s)
        ops.zero_async(self._dg_all, side)
        # the compositing backward's accumulator record too (MomRasterArgs.accum_cleared): its last reader, the previous
        # step's projection backward, is behind this stream's wait above
        gk = (P, W, H, self.geom.data_ptr())
        if getattr(self, "_gacc_cache", (None,))[0] != gk:
            self._gacc_cache = (gk, self._gacc_view(P, W, H))
        ops.zero_async(self._gacc_cache[1], side)
        if hy.time_smoothness_weight != 0:
            rkey = (hy.time_smoothness_weight, hy.plane_tv_weight, hy.l1_time_planes, reg_scale)
            if self._reg_arr is None or self._reg_arr[0] != rkey:
                arr = (N.MomRegPlane * len(planes))()
                for i, p in enumerate(planes):
                    st, gs = ops.plane_storage(p.detach()), ops.plane_storage(self._dg_planes[i])
                    arr[i].plane, arr[i].grad = st.data_ptr(), gs.data_ptr()
                    arr[i].H, arr[i].W = st.shape[0], st.shape[1]
                    tplane = (i % 6) in (2, 4, 5)
                    arr[i].w_smooth = hy.time_smoothness_weight if tplane else hy.plane_tv_weight
                    arr[i].w_l1 = hy.l1_time_planes if tplane else 0.0
                    arr[i].grad_scale = reg_scale      # identical on every rank: the sum over ranks restores it
                self._reg_arr = (rkey, arr)
            arr = self._reg_arr[1]
            ops.zero_async(self.regacc, side)
            N.check(lib.mom_plane_regulation_acc(arr, len(planes), self.regacc.data_ptr(), side), "plane_reg")
            reg = self.regacc
        ops.stream_mark(ops.MARK_BUCKET, side)
        # the field's processing orders, if their refresh is due within a few steps: sorted on the second stream from here on, beside
        # this step's forward and backward (scene/hexplane.py: prefetch_if_due)
        if dc is None or dc.mode == "camera":
            field.prefetch_if_due(xyz, side)
        # HexPlane lookup + MLP + the activations (exp / normalize / sigmoid) in one kernel (csrc/deform_field.hip)
        dc = self.dist
        sl = None
        if dc is not None and dc.mode == "tile-row":
            # the deformation field is half of a step and does not care which rank computes which Gaussian: every rank takes a
            # contiguous slice [g0, g1) of them, forward and backward, and the ranks all-gather the deformed state (what the
            # replicated projection needs of ALL Gaussians: 15 floats each) while nothing else is in flight
            S = dc.slice_rows(P)
            g0 = min(P, dc.rank * S)
            g1 = min(P, g0 + S)
            sl = (S, g0, g1)
            # The five tensors of the deformed state are gathered IN PLACE, each in its own [world * S, k] array (the slice's
            # field kernel writes this rank's rows directly), as ONE grouped submission on the direct RCCL transport
            # (ncclGroupStart / End: one launch).  Round 5 packed them into one slab per rank for a single torch.distributed call
            # (40-50 us of host each) and paid five strided copies behind the gather at every world size; with the collectives
            # behind the C ABI a call costs 0.6 us and the copies are gone.  (torch.distributed, the fallback and the CPU tests'
            # transport, issues the five gathers one after the other.)
            if g1 > g0:
                v = lambda t: t[g0:g1]
                so = field._slice_order(xyz, g0, g1)
                lines_kept = ops.field_forward(hp, md, g1 - g0, v(xyz), time, so, v(scal), v(rot), v(flow), coef, v(self.pts), v(self.sc_d),
                                  v(self.rot_d), v(self.feat), v(self.a0), v(opac), v(self.sc), v(self.rot), v(self.op), s)
            grp = None
            if hasattr(dc, "group"):
                with dc.group() as grp:
                    dc.start_gather([self.pts, self.sc, self.rot, self.op, self.rot_d], S)
            else:
                dc.start_gather([self.pts, self.sc, self.rot, self.op, self.rot_d], S)
            if grp is not None and getattr(grp, "work", None) is not None:
                dc._pending.append(grp.work)
            dc.finish()
        else:
            lines_kept = ops.field_forward(hp, md, P, xyz, time, order, scal, rot, flow, coef, self.pts, self.sc_d, self.rot_d, self.feat, self.a0,
                              opac, self.sc, self.rot, self.op, s)
        # ---- rasterizer forward (async: capacity from the previous iterations, checked below)
        a = N.MomRasterArgs()
        a.P, a.D, a.M, a.W, a.H = P, g.active_sh_degree, 16, W, H
        a.accum_cleared = 1                     # on the second stream, above
        a.background, a.means3D = self.bg.data_ptr(), self.pts.data_ptr()
        a.shs, a.shs_rest = g._features_dc.data_ptr(), g._features_rest.data_ptr()
        a.colors_precomp, a.opacities = None, self.op.data_ptr()
        a.scales, a.rotations, a.cov3D_precomp = self.sc.data_ptr(), self.rot.data_ptr(), None
        a.viewmatrix, a.projmatrix, a.campos = view.data_ptr(), proj.data_ptr(), campos.data_ptr()
        a.scale_modifier = 1.0
        a.tan_fovx, a.tan_fovy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        a.prefiltered, a.debug = 0, 0
        a.keep_all_tiles = int(self.keep_all_tiles)
        # L1 (its gradient image and its sums) in the compositing kernel's epilogue.  A tile-row shard too, unless its forward renders a
        # halo (SSIM term): the kernel runs this rank's tiles only, leaves one pair of sums per tile, and the rank adds up ITS rows
        tr_rows = dc is not None and dc.mode == "tile-row"
        l1_scaled = False
        fuse_l1 = not tr_rows or (L1_PARTIALS and float(self.opt.lambda_dssim) == 0)
        if fuse_l1:
            a.l1_target, a.l1_grad = gt.data_ptr(), self.dimg.data_ptr()
            # camera-batch shard: the batch loss is the mean over the ranks' cameras, every gradient carries 1 / world (below)
            # -- in the epilogue itself when the loss is L1 alone (with the SSIM term the whole gradient image is scaled once, below)
            l1_scaled = dc is not None and dc.mode == "camera" and float(self.opt.lambda_dssim) == 0
            a.l1_grad_scale = 1.0 / dc.world if l1_scaled else 0.0
            if L1_PARTIALS or tr_rows:
                a.l1_partials = self.l1_part.data_ptr()
            else:
                a.l1_sums = self.sums.data_ptr()
        a.overflow_tag = self.next_tag          # what this step leaves in the sticky word if its binning overflows (Trainer numbers the steps)
        rows = fwd_rows = None
        if dc is not None and dc.mode == "tile-row":
            # every rank renders this same camera, restricted to its rows of 16-pixel tiles.  With the SSIM term the forward
            # also renders one tile row of halo on each side: the 11x11 window makes an owned pixel depend on map pixels up to
            # 5 rows outside, and those on image pixels up to 10 rows outside -- all inside the 16-row halo, so nothing is
            # exchanged (the neighbour composites the same rows to the same bits).  The backward covers the own rows only.
            gy = (H + 15) // 16
            rows = dc.rows(gy)
            halo = 1 if (float(self.opt.lambda_dssim) != 0 and rows[1] > rows[0]) else 0
            fwd_rows = (max(rows[0] - halo, 0), min(rows[1] + halo, gy))
            a.tile_row0, a.tile_row1 = fwd_rows
        # an earlier iteration's instance count (whichever copy landed last: a sizing hint, read without blocking)
        prev_R = int(self.nr_host[0])
        N.check(lib.mom_raster_forward_geometry(C.byref(a), self.geom.data_ptr(), self.img.data_ptr(), self.radii.data_ptr(),
                                                self.nr_dev.data_ptr(), self.nr_host.data_ptr(), s), "raster_geometry")
        if self.cap == 0:                       # first call: size exactly (one sync)
            torch.cuda.current_stream().synchronize()
            prev_R = int(self.nr_host[0])
        want = max(prev_R, int(prev_R * self.HEADROOM) + self.MARGIN)
        if self.binning is None or want > self._bin_cap or want < self._bin_cap // 4:
            self._bin_cap = want
            self.binning = torch.empty(lib.mom_raster_binning_bytes(P, W, H, want), dtype=torch.uint8, device=dev)
        self.cap = self._bin_cap          # the capacity every launch of this step is told (cap == 0 above: "size exactly")
        N.check(lib.mom_raster_forward_render(C.byref(a), self.geom.data_ptr(), self.binning.data_ptr(), self.cap,
                                              self.img.data_ptr(), self.color.data_ptr(), self.depth.data_ptr(),
                                              self.flags.data_ptr(), s), "raster_render")
        early_works = []                        # camera-batch shard: what the early Adam launch must see reduced (below)
        if dc is not None and dc.mode != "camera":
            early_works.append(dc.start(self.flags, "max"))         # every rank skips (and later replays) the same steps
            # (a camera-batch shard agrees on the word together with the radii, behind its own backward: below)
        # ---- loss: L1 (+ its gradient image) ; regulariser value and gradient
        n = self.color.numel()
        if not fuse_l1:
            N.check(lib.mom_l1_loss_acc(n, self.color.data_ptr(), gt.data_ptr(), self.dimg.data_ptr(), self.sums.data_ptr(), s), "l1")
        # camera-batch shard: the batch loss is the mean over the ranks' cameras (train_4DGS.py:189-229), so every
        # gradient carries 1/world and the all-reduces below are plain sums (1/2, 1/4, 1/8 are exact in fp32)
        inv_world = 1.0 / dc.world if (dc is not None and dc.mode == "camera") else 1.0
        # tile-row shard: the ranks' deformation gradients are SUMMED (each holds its slice's share); the regulariser's gradient,
        # which every rank computes in full, must enter that sum once
        reg_scale = 1.0 / dc.world if dc is not None else 1.0
        lam = float(self.opt.lambda_dssim)
        if lam != 0:
            # loss += lambda_dssim * (1 - ssim(image, gt))  (train_4DGS.py:222-223): its gradient is added into dimg
            win = ops._ssim_window()
            if self.ssim_dm is None:         # derivative maps and the map sum (mom_ssim_forward)
                self.ssim_dm = torch.empty((3, 3, H, W), dtype=torch.float32, device=dev)
                self.ssim_sum = torch.empty(N.SSIM_SUM_SLOTS, dtype=torch.float64, device=dev)   # [0] = the sum
            if rows is None:
                N.check(lib.mom_ssim_forward(3, H, W, win, self.color.data_ptr(), gt.data_ptr(), self.ssim_dm.data_ptr(),
                                             self.ssim_sum.data_ptr(), s), "ssim_fwd")
                N.check(lib.mom_ssim_backward(3, H, W, win, self.color.data_ptr(), gt.data_ptr(), self.ssim_dm.data_ptr(),
                                              -lam / n, None, self.dimg.data_ptr(), s), "ssim_bwd")
            elif rows[1] > rows[0]:
                # the slab this rank rendered (own rows + halo), as a pitched view of the full buffers: map rows of the own
                # tile rows count toward the sum; derivative maps are kept 5 rows beyond them (up to the image's own edges,
                # where zero padding is the reference's behaviour) and are zero elsewhere
                ys0, ys1 = fwd_rows[0] * 16, min(H, fwd_rows[1] * 16)
                y0, y1 = rows[0] * 16, min(H, rows[1] * 16)
                off = ys0 * W * 4
                N.check(lib.mom_ssim_forward_slab(3, ys1 - ys0, W, H * W, y0 - ys0, y1 - ys0, max(0, y0 - 5) - ys0,
                                                  min(H, y1 + 5) - ys0, win, self.color.data_ptr() + off, gt.data_ptr() + off,
                                                  self.ssim_dm.data_ptr() + off, self.ssim_sum.data_ptr(), s), "ssim_fwd_slab")
                N.check(lib.mom_ssim_backward_slab(3, ys1 - ys0, W, H * W, win, self.color.data_ptr() + off, gt.data_ptr() + off,
                                                   self.ssim_dm.data_ptr() + off, -lam / n, None, self.dimg.data_ptr() + off, s),
                        "ssim_bwd_slab")
            else:
                self.ssim_sum.zero_()
        if dc is not None and dc.mode == "camera" and not (fuse_l1 and l1_scaled):
            self.dimg.mul_(inv_world)            # (L1 alone: the forward's epilogue wrote the gradient image with the factor in it)
        # ---- rasterizer backward
        ops.stream_wait_mark(s, ops.MARK_BUCKET)      # the gradient bucket is cleared and holds the regulariser's share
        gr = N.MomRasterGrads()
        # scale / rotation / opacity gradients leave the projection backward already through exp / normalize / sigmoid
        # (MomRasterGrads.act_rotations_raw): no activation-backward launch behind it
        gr.dL_dmeans2D, gr.dL_dcolors, gr.dL_dopacity = self.g2d.data_ptr(), self.gcol.data_ptr(), self.gop.data_ptr()
        gr.dL_dmeans3D, gr.dL_dcov3D = self.gxyz.data_ptr(), self.gcov.data_ptr()
        gr.dL_dsh, gr.dL_dsh_rest = self.gdc.data_ptr(), self.grest.data_ptr()
        gr.dL_dscales, gr.dL_drotations = self.gsc.data_ptr(), self.grot.data_ptr()
        gr.act_rotations_raw = self.rot_d.data_ptr()
        if dc is not None and dc.mode == "camera":
            # the deformation backward still reads this rank's own d_sc / d_rot while the bucket that holds them is being reduced in
            # place: its private copies (7 floats per Gaussian) come out of the same kernel
            gr.dL_dscales_copy, gr.dL_drotations_copy = self._loc[0].data_ptr(), self._loc[1].data_ptr()
        if rows is None:
            N.check(lib.mom_raster_backward(C.byref(a), self.radii.data_ptr(), self.geom.data_ptr(), self.binning.data_ptr(),
                                            self.cap, self.img.data_ptr(), self.dimg.data_ptr(), None, C.byref(gr), s), "raster_bwd")
        else:
            # tile-row shard: the compositing backward covers this rank's rows only (dimg outside them is never read); the
            # per-Gaussian record it leaves is summed over the ranks, after which the projection backward -- linear in that
            # record -- and everything downstream give the same gradients on every rank, with nothing left to exchange
            if hasattr(dc, "rebalance_due") and dc.rebalance_due():
                # instance counts per tile row of the rows this rank owns -> agreed weights for the next splits; after a
                # re-split the local instance count can jump, so the next step sizes its binning buffer exactly again
                gx_ = (W + 15) // 16
                lay = N.MomRasterLayout()
                lib.mom_raster_layout(P, W, H, 0, C.byref(lay))
                base = self.img[(-self.img.data_ptr()) % 256:]
                per_row = base[lay.img_tile_counts:lay.img_tile_counts + gx_ * gy * 4].view(torch.int32).view(gy, gx_).sum(1).float()
                own = torch.zeros_like(per_row)
                own[rows[0]:rows[1]] = per_row[rows[0]:rows[1]]
                if dc.rebalance_rows(own):
                    # the local instance count can jump with the new rows: size the NEXT step's buffer exactly.  This step's
                    # backward below still runs on the buffer (and capacity) its forward filled.
                    self._resize_next = True
            a.tile_row0, a.tile_row1 = rows          # own rows only (the forward may have covered a halo)
            N.check(lib.mom_raster_backward_render(C.byref(a), self.geom.data_ptr(), self.binning.data_ptr(), self.cap,
                                                   self.img.data_ptr(), self.dimg.data_ptr(), None, s), "raster_bwd_render")
            dc.start(self._gacc_view(P, W, H), "sum")
            dc.finish()
            N.check(lib.mom_raster_backward_geometry(C.byref(a), self.radii.data_ptr(), self.geom.data_ptr(), C.byref(gr), s),
                    "raster_bwd_geometry")
        if dc is not None and dc.mode == "camera":
            # densification statistics (train_4DGS.py:203-204,227-229): largest radius, mean 2-D gradient
            # (this rank's backward has read its OWN radii by now; the overflow word rides in the same integer bucket.  Every
            # torch.distributed call costs the host 40-50 us and the host paces a rank: three collectives per step, not five)
            early_works.append(dc.start(self.ibucket, "max"))
        d_sc, d_rot = self.gsc, self.grot       # also the gradients w.r.t. the MLP's scale / rotation outputs
        if dc is not None and dc.mode == "camera":      # 56 of the 59 floats per Gaussian travel underneath the deformation backward
            d_sc, d_rot = self._loc            # private copies, written by the projection backward (above)
            if sharded:
                # reduce-scatter of the appearance gradients (this rank keeps the sum of ITS chunk) + the small all-reduce of the
                # screen-space gradients every rank's statistics need in full: one launch on the direct path
                with dc.group() as grp:
                    w_rs = dc.start_reduce_scatter(self.app_flat, self._chunk, "sum")
                    w_g2 = dc.start(self.g2d_flat, "sum")
                if grp.work is not None:
                    dc._pending.append(grp.work)
                    early_works.append(grp.work)
                else:
                    early_works += [w_rs, w_g2]
            else:
                early_works.append(dc.start(self.early_bucket, "sum"))     # appearance gradients + mean 2-D gradients
        # ---- deformation backward: pts = xyz + dx(...) so d xyz starts as d pts (already in gxyz); the HexPlane adds its share
        # the MLP's weight-gradient kernel (matrix pipe) runs on a second stream beside the HexPlane backward (vector issue,
        # memory latency); joined below, before anything reads the weight gradients
        side = self.side.cuda_stream if self.OVERLAP_DW else s
        early_cam = None
        if early_adam is not None and self.EARLY_ADAM:
            # The appearance parameters' gradients (SH, scaling, rotation, opacity: 56 of a Gaussian's 59 floats) are final here.
            # Their Adam update -- a pure HBM stream, 335 of Adam's 412 MB -- goes to the second stream now and runs underneath the
            # MLP backward (matrix pipe, 2 TB/s); nothing on this stream reads those parameters again before the join below.
            for p, gbuf in ((g._features_dc, self.gdc), (g._features_rest, self.grest), (g._scaling, self.gsc),
                            (g._rotation, self.grot), (g._opacity, self.gop)):
                p.grad = gbuf
            if dc is None or dc.mode == "tile-row":
                # (a tile-row shard's gradients are already the full sums here: the ranks summed the compositing backward's record,
                # and projection / activation backward ran replicated on it)
                ops.stream_wait_stream(self.side.cuda_stream, s)
                early_adam([g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity], stream=self.side.cuda_stream)
            else:
                # camera-batch shard: the launch needs the REDUCED bucket (and, for the densification statistics it carries, the
                # reduced radii / screen-space gradients and the agreed overflow word).  The second stream waits for exactly those
                # collectives -- each was begun behind the kernels that produced its buffer, so it orders the stream behind them
                # too -- and not for this stream, which goes on with the deformation backward.  Enqueued BELOW, after that
                # backward: on RCCL wait() only makes the waiting stream wait, but gloo (the tests) blocks the host in it, and the
                # host should block with the GPU's work already queued.
                early_cam = list(early_works)

        def launch_early_cam():
            """Camera-batch shard: the second stream waits for the early collectives, then takes the appearance parameters' Adam
            launch (sharded: this rank's 1/world of it, then the in-place all-gather of the UPDATED parameters behind it -- begun from
            inside the second stream's context, so it is ordered behind that stream's Adam launch; the caller's DistContext.finish()
            makes the main stream wait for it before anything reads a parameter)."""
            app = [g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity]
            if getattr(dc, "direct", None) is not None and not sharded:
                # direct RCCL path: a wait is one hipStreamWaitEvent on the second stream's raw handle (no torch stream context: ~10 us)
                dc.wait_for(early_cam, stream=self.side.cuda_stream)
                early_adam(app, stream=self.side.cuda_stream)
                return
            with torch.cuda.stream(self.side):          # (torch.distributed's wait() orders the CURRENT torch stream)
                dc.wait_for(early_cam)
                if sharded:
                    ranges = {id(p_): r for p_, r in zip(app, dc.shard_ranges(self._cut, self._chunk))}
                    early_adam(app, stream=self.side.cuda_stream, ranges=ranges)
                    dc.start_gather_flat(self.pflat, self._chunk)
            if not sharded:
                early_adam(app, stream=self.side.cuda_stream)

        if early_cam is not None and not getattr(dc, "host_blocking", False):
            # On RCCL (either transport) a wait only orders streams: the launch goes to the second stream NOW, ahead of the MLP
            # backward's reduction kernel that is about to be queued there, and runs underneath that backward as in the unsharded
            # step.  (Queued behind the reduction it ran beside the HexPlane gather instead, which it slows from 83 to 116 us.)
            launch_early_cam()
            early_cam = None
        if sl is None:
            N.check(lib.mom_deform_backward_split(C.byref(md), P, self.feat.data_ptr(), self.a0.data_ptr(), self.gxyz.data_ptr(),
                                                  d_sc.data_ptr(), d_rot.data_ptr(), self.dfeat.data_ptr(),
                                                  self.dh_scratch.data_ptr(), s, side), "deform_bwd")
            if porders is not None:
                self._hex_scratch_for(hp, P, dev)
            if porders is not None and lines_kept:      # the forward's time lines are still in the field scratch
                N.check(lib.mom_hexplane_backward_lines(C.byref(hp), P, xyz.data_ptr(), time, optr, self.dfeat.data_ptr(),
                                                        self.gxyz.data_ptr(), porders[0].data_ptr(), porders[1].data_ptr(),
                                                        self._hex_scratch.data_ptr(), ops.field_scratch(hp, dev).data_ptr(), s),
                        "hexplane_bwd")
            else:
                N.check(lib.mom_hexplane_backward(C.byref(hp), P, xyz.data_ptr(), None, time, optr, self.dfeat.data_ptr(),
                                                  self.gxyz.data_ptr(), None if porders is None else porders[0].data_ptr(),
                                                  None if porders is None else porders[1].data_ptr(),
                                                  None if porders is None else self._hex_scratch.data_ptr(), s), "hexplane_bwd")
        else:
            # tile-row shard: the deformation backward of this rank's slice only.  Its weight / plane gradients are partial sums
            # (summed over the ranks below); its position gradients complete gxyz for the slice's rows, which the ranks then
            # all-gather -- the other rows hold d pts only, replicated by the projection backward.
            S, g0, g1 = sl
            ns = g1 - g0
            if ns > 0:
                v = lambda t: t[g0:g1]
                N.check(lib.mom_deform_backward_split(C.byref(md), ns, v(self.feat).data_ptr(), v(self.a0).data_ptr(),
                                                      v(self.gxyz).data_ptr(), v(d_sc).data_ptr(), v(d_rot).data_ptr(),
                                                      v(self.dfeat).data_ptr(), self.dh_scratch.data_ptr(), s, side), "deform_bwd")
                so = field._slice_order(xyz, g0, g1, bump=False)      # the forward's order of this step
                spo = field._slice_plane_orders(xyz, g0, g1)
                if spo is not None:
                    self._hex_scratch_for(hp, ns, dev)
                N.check(lib.mom_hexplane_backward(C.byref(hp), ns, v(xyz).data_ptr(), None, time, None if so is None else so.data_ptr(),
                                                  v(self.dfeat).data_ptr(), v(self.gxyz).data_ptr(),
                                                  None if spo is None else spo[0].data_ptr(), None if spo is None else spo[1].data_ptr(),
                                                  None if spo is None else self._hex_scratch.data_ptr(), s), "hexplane_bwd")
            dc.start_gather([self.gxyz_rows], S)
        if early_cam is not None:
            launch_early_cam()          # (gloo: its wait() blocks the host, so only now, with the deformation backward queued)
        if self.side is not None:
            ops.stream_wait_stream(s, self.side.cuda_stream)
        if dc is not None and dc.mode == "camera":
            dc.start(self._dg_flat, "sum")     # xyz + deformation field; the caller waits (DistContext.finish) before Adam
        elif sl is not None:
            # this rank's share of the logged loss value -- [sum |d|, sum d^2, sum of the SSIM map] over its OWN rows (the gradient
            # image is already normalised by the whole image's element count) -- rides in front of the field gradients
            if rows[1] > rows[0]:
                if fuse_l1:     # the compositing epilogue left one pair of sums per tile of this rank's rows
                    gx_ = (W + 15) // 16
                    torch.sum(self.l1_part[rows[0] * gx_:rows[1] * gx_], dim=0, out=self._tr_sums[:2])
                else:
                    y0, y1 = rows[0] * 16, min(H, rows[1] * 16)
                    for c in range(3):
                        N.check(lib.mom_l1_loss_acc((y1 - y0) * W, self.color[c, y0:y1].data_ptr(), gt[c, y0:y1].data_ptr(), None,
                                                    self._tr_sums.data_ptr(), s), "l1_slab")
                if lam != 0:
                    self._tr_sums[2:3].copy_(self.ssim_sum[:1])
            # the deformation field's gradients: the slices' shares (and the regulariser's, once) + the loss sums, one collective
            dc.start(self._dg_all[:4 + self._dg_n], "sum")
            dc.finish()
        # ---- hand the gradients to the parameters
        for p, gbuf in ((g._xyz, self.gxyz), (g._features_dc, self.gdc), (g._features_rest, self.grest), (g._scaling, self.gsc),
                        (g._rotation, self.grot), (g._opacity, self.gop)):
            p.grad = gbuf
        for p, gbuf in zip(planes, self._dg_planes):
            p.grad = gbuf
        for p, gbuf in zip(mlp, self._dg_mlp):
            p.grad = gbuf
        if rows is None:
            l1 = None                        # formed lazily from self.sums (LazyLoss): no kernels for a value nobody may read
        else:
            # the image holds this rank's rows only: the logged values are the sums over the ranks, reduced above
            self.sums.copy_(self._tr_sums[:2])
            if lam != 0:
                self.ssim_sum[:1].copy_(self._tr_sums[2:3])
            l1 = self._tr_sums[0] / n
        sums = _TileSums(self.l1_part) if (fuse_l1 and L1_PARTIALS and rows is None) else self.sums
        loss = LazyLoss(sums if l1 is None else None, l1, reg, self.ssim_sum if lam != 0 else None, lam, n)
        self.last = {"loss": loss, "mse_sum": _Lazy(sums, 1), "n": n}       # float(last["mse_sum"]): formed when read
        return loss, self.radii, self.g2d


class _TileSums:
    """[sum |image - target|, sum (image - target)^2] of the last step from the per-tile pairs the compositing forward left: sums[k] is a
    device scalar formed on demand (one torch reduction, in a fixed order: the same bits for the same image).  Like LazyLoss it must be
    read before the next step overwrites the pairs."""

    def __init__(self, part):
        self._part, self._t = part, None

    def __getitem__(self, k):
        if self._t is None:
            self._t = self._part.sum(0)
        return self._t[k]


class _Lazy:
    """float(x) / x.tensor() = sums[k], looked up when asked for."""

    def __init__(self, sums, k):
        self._sums, self._k = sums, k

    def tensor(self):
        return self._sums[self._k]

    def __float__(self):
        return float(self.tensor())


class LazyLoss:
    """The step's loss value, formed on demand from the accumulators the kernels left on the device (the training loop only
    needs it for logging): float(loss), loss.item(), loss.tensor().  The accumulators are overwritten by the next step, so a
    value wanted later must be taken (tensor()) before that step is enqueued; tensor() is stream-ordered like any torch op."""

    def __init__(self, sums, l1, reg, ssim_sum, lam, n):
        self._sums, self._l1, self._reg, self._ssim, self._lam, self._n, self._t = sums, l1, reg, ssim_sum, lam, n, None

    @property
    def l1(self):
        return self._l1 if self._l1 is not None else self._sums[0] / self._n

    def tensor(self):
        if self._t is None:
            v = self.l1
            if self._reg is not None:
                v = v + self._reg[0]
            if self._ssim is not None:
                v = v + self._lam * (1.0 - (self._ssim[0] / self._n).float())
            self._t = v
        return self._t

    def detach(self):
        return self.tensor().detach()

    def item(self):
        return self.tensor().item()

    def __float__(self):
        return float(self.tensor())
