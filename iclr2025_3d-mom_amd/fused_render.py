"""Forward-only rendering of one camera as a fixed launch sequence: HexPlane -> deformation MLP -> activations ->
projection / binning / sort / compositing.  gaussian_renderer.render() takes this path when gradients are disabled and the
model has the shipped configuration (the case of render_4DGS.py and of every evaluation render): the same kernels as the
fused training step, no SH concatenation (the rasterizer reads the DC and the rest coefficients through two pointers), no
gradient holder, no per-op allocation of intermediates.  Only the returned image, depth and radii are fresh tensors --
callers keep them.
"""
import ctypes as C
import math
from collections import deque

import torch

from . import _native as N
from . import ops
from .diff_gaussian_rasterization import _C as RC


class FusedRender:
    def __init__(self, gaussians):
        self.g = gaussians
        self.lib = N.lib()
        self.key = None
        self.cap = 0
        self.binning = None
        self._desc_key = None
        self.serial = 0
        self.pending = deque()
        self.collect = False         # True: frames found overflowed FLAG_LAG frames later go to self.bad instead of raising
        self.bad = []

    def _ensure(self, P, W, H, dev):
        if self.key == (P, W, H, dev):
            return
        self.key = (P, W, H, dev)
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.feat = e(P, 64)
        self.pts, self.sc_d, self.rot_d = e(P, 3), e(P, 3), e(P, 4)
        self.sc, self.rot, self.op = e(P, 3), e(P, 4), e(P, 1)
        self.geom = torch.empty(self.lib.mom_raster_geom_bytes(P), dtype=torch.uint8, device=dev)
        self.img = torch.empty(self.lib.mom_raster_image_bytes(W, H), dtype=torch.uint8, device=dev)
        self.nr_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.nr_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        # per-frame status word of the image scratch (header word 1: bit 0 = this frame's binning overflowed): the compositing kernel
        # posts it, with the frame's serial number, into a slot of this pinned ring (MomRasterArgs.status_post) -- no copy command and
        # no event behind an async-mode frame (a blit kernel and a marker were 12 us of the stream per frame)
        self.hdr = self.img[(-self.img.data_ptr()) % 256:][:8].view(torch.int32)
        self.flag_ring = torch.zeros(self.RING, dtype=torch.int64).pin_memory()
        self._ring_np = self.flag_ring.numpy()      # (a view of the same pinned words: reading one costs 0.1 us, indexing the tensor 2)
        self.pending = deque()               # (frame serial, ring slot), oldest first
        self.cap, self.binning = 0, None

    RING, FLAG_LAG = 64, 8
    HEADROOM, MARGIN = 1.5, 65536       # async binning capacity = HEADROOM x an earlier frame's instance count + MARGIN

    def overflowed(self, lag=0):
        """Serial numbers (FusedRender.serial after the render() that produced them) of async-mode frames older than `lag`
        frames whose binning buffer overflowed -- their images are incomplete and must be rendered again.  lag=0 waits for
        everything rendered so far.  The capacity is raised so that a repeat fits."""
        bad = []
        while len(self.pending) > lag:
            serial, slot = self.pending.popleft()
            if self._posted(serial, slot) & 1:
                bad.append(serial)
        if bad:
            self.cap_floor = max(getattr(self, "cap_floor", 0), 2 * self.cap)
        return bad

    def _posted(self, serial, slot, timeout_s=60.0):
        """The status bits frame `serial` left in ring slot `slot`: polled until the slot's upper half carries that serial (the frames
        that are asked about are FLAG_LAG frames old: it is there)."""
        import time
        want = serial & 0xFFFFFFFF
        ring = self._ring_np
        v = int(ring[slot])
        if (v >> 32) & 0xFFFFFFFF != want:
            t0 = time.perf_counter()
            while True:
                v = int(ring[slot])
                if (v >> 32) & 0xFFFFFFFF == want:
                    break
                if time.perf_counter() - t0 > timeout_s:
                    raise N.MomError(f"async render(): frame {serial} never posted its status (slot {slot} holds {v:#x})")
        return v & 0xFFFFFFFF

    def render(self, cam, bg, delta_scale, scaling_modifier=1.0, debug=False, order=False):
        """order: the field's processing order if the caller already has it (FusedRenderPool takes it on the caller's stream)."""
        g, lib, s = self.g, self.lib, N.current_stream()
        dev = g._xyz.device
        P = g._xyz.shape[0]
        W, H = int(cam.image_width), int(cam.image_height)
        self._ensure(P, W, H, dev)
        view, proj, campos, _ = cam.device_tensors(dev)
        dn = g._deformation.deformation_net
        field = dn.grid
        xyz, scal, rot, opac = g._xyz.detach(), g._scaling.detach(), g._rotation.detach(), g._opacity.detach()
        flow = g._scene_flow if g._scene_flow.is_contiguous() else g._scene_flow.contiguous()
        if order is False:
            order = field._processing_order(xyz)
        planes = [p for lv in field.grids for p in lv]
        mlp = dn._fused_params()
        dkey = (tuple(p.data_ptr() for p in planes + mlp), tuple(field.aabb_host()))
        if self._desc_key != dkey:
            hp, keep = ops._hexplane_desc([[p.detach() for p in lv] for lv in field.grids], field.aabb, None,
                                          aabb_host=field.aabb_host())
            md = ops.DeformMLPFunction._desc([p.detach() for p in mlp], None)
            self._desc, self._desc_key = (hp, keep, md), dkey
        hp, keep, md = self._desc
        # this renderer's OWN field scratch (time-line table + feature buffer): a FusedRenderPool keeps one frame per slot in flight
        # on unsynchronised streams, and the per-device scratch of ops.field_scratch would be rewritten by frame k+1's line kernel
        # while frame k's field kernel still reads it
        need = lib.mom_deform_field_scratch_bytes(C.byref(hp), P)
        if getattr(self, "_fscratch", None) is None or self._fscratch.numel() < need or self._fscratch.device != dev:
            self._fscratch = torch.empty(need, dtype=torch.uint8, device=dev)
        ops.field_forward(hp, md, P, xyz, float(cam.time), order, scal, rot, flow, float(delta_scale * cam.frame_num), self.pts,
                          self.sc_d, self.rot_d, None, None, opac, self.sc, self.rot, self.op, s, scratch_feat=self.feat,
                          scratch=self._fscratch)
        a = N.MomRasterArgs()
        a.P, a.D, a.M, a.W, a.H = P, g.active_sh_degree, 16, W, H
        a.background, a.means3D = bg.data_ptr(), self.pts.data_ptr()
        a.shs, a.shs_rest = g._features_dc.data_ptr(), g._features_rest.data_ptr()
        a.colors_precomp, a.opacities = None, self.op.data_ptr()
        a.scales, a.rotations, a.cov3D_precomp = self.sc.data_ptr(), self.rot.data_ptr(), None
        a.viewmatrix, a.projmatrix, a.campos = view.data_ptr(), proj.data_ptr(), campos.data_ptr()
        a.scale_modifier = float(scaling_modifier)
        a.tan_fovx, a.tan_fovy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        a.prefiltered, a.debug = 0, int(bool(debug))
        a.keep_all_tiles = int(RC._state["keep_all_tiles"])
        a.forward_only = 1               # no backward follows: cov3D / clamped / final_T / n_contrib are not written
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        # Binning capacity, as diff_gaussian_rasterization._C does it.  "exact" (the reference's own synchronisation point,
        # rasterizer_impl.cu:282): wait for this frame's instance count.  "async": size from the previous frames' counts with
        # headroom and do not wait; a frame that does not fit is flagged per frame (overflowed()): callers that render a
        # whole trajectory collect the flagged frames at the end and render them again (render.render_set); a caller that
        # never asks gets a MomError FLAG_LAG frames later.
        prev_R = int(self.nr_host[0])
        late = self.overflowed(self.FLAG_LAG) if len(self.pending) > self.FLAG_LAG else []
        if late and self.collect:
            self.bad += late
        elif late:
            raise N.MomError(f"async render(): frames {late} overflowed the binning capacity and are incomplete; render them "
                             "again (the capacity has been raised), collect such frames with FusedRender.overflowed(), or use "
                             "set_sync_mode('exact')")
        wait_for_count = RC._state["mode"] == "exact" or self.cap == 0
        if wait_for_count:
            if getattr(self, "_unwaited", False):
                # frames whose count nobody waited for (async mode) may still be queued: their geometry stage would overwrite the
                # sentinel below with THEIR count, and this frame's binning would be sized from it.  Once, on the switch to a
                # waiting frame: let the stream drain.
                torch.cuda.current_stream(dev).synchronize()
                self._unwaited = False
            self.nr_host[0] = RC.COUNT_PENDING       # (after prev_R was read: the geometry stage overwrites it with this frame's count)
        else:
            self._unwaited = True
        N.check(lib.mom_raster_forward_geometry(C.byref(a), self.geom.data_ptr(), self.img.data_ptr(), radii.data_ptr(),
                                                self.nr_dev.data_ptr(), self.nr_host.data_ptr(), s), "raster_geometry")
        if wait_for_count:
            # polled, not waited for with hipStreamSynchronize (RC.wait_count: that wait parks the thread and is woken up to a
            # millisecond late -- four frames' worth at config 2)
            count = RC.wait_count(self.nr_host)
            want = count + (0 if RC._state["mode"] == "exact" else count // 2 + 65536)
        else:
            want = max(self.cap, int(prev_R * self.HEADROOM) + self.MARGIN, getattr(self, "cap_floor", 0))
        if self.binning is None or want > self.cap or want < self.cap // 4:
            self.cap = want
            self.binning = torch.empty(lib.mom_raster_binning_bytes(P, W, H, self.cap), dtype=torch.uint8, device=dev)
        self.serial += 1
        post = RC._state["mode"] != "exact"
        if post:
            # (the slot is reused RING frames later, long after overflowed() has looked at it FLAG_LAG frames behind; serial 0 is never
            # posted, so a fresh ring matches nothing)
            slot = self.serial % self.RING
            a.status_post, a.status_serial = self.flag_ring.data_ptr() + 8 * slot, self.serial & 0xFFFFFFFF
        N.check(lib.mom_raster_forward_render(C.byref(a), self.geom.data_ptr(), self.binning.data_ptr(), self.cap,
                                              self.img.data_ptr(), color.data_ptr(), depth.data_ptr(),
                                              None, s), "raster_render")
        if post:
            self.pending.append((self.serial, slot))
        return color, depth, radii


class FusedRenderPool:
    """Consecutive frames on n alternating streams (gaussian_renderer.set_render_streams(n), n > 1).

    A frame's deformation field, projection and binning are short kernels bound by latency (three waves per SIMD for 20 us at a
    time); its compositing kernel is bound by vector issue and fills the chip.  One stream runs them strictly one after the other:
    276 us per frame at config 2.  Frames are independent, so the next frame's front half can run while this frame composites, if
    it is on another stream with its own scratch: 3715 -> 5290 frames/s with two streams, 5720 with three (tools/probe/
    fps_two_streams.py).  Every slot is a FusedRender of its own (geometry / image / binning state, capacity bookkeeping, overflow
    ring); the model's tensors are shared and only read.

    What the caller must know: a frame is produced on ITS SLOT'S stream and render() does not make the caller's stream wait for it
    (that wait would chain every frame behind the one before).  render() returns the stream and an event (`"stream"`, `"ready"`
    in gaussian_renderer.render()'s dict): consume the frame inside `with torch.cuda.stream(out["stream"])`, or after
    `torch.cuda.current_stream().wait_event(out["ready"])` (+ `tensor.record_stream(...)` if it is kept past its next use on the
    slot's stream), or after a device synchronisation.  That is why the mode is opt-in: the reference's own loop reads the image
    with `.cpu()` right away (render_4DGS.py:64), which only waits for the current stream."""

    def __init__(self, gaussians, n):
        self.g = gaussians
        self.n = int(n)
        self.slots = [FusedRender(gaussians) for _ in range(self.n)]
        self.streams = [torch.cuda.Stream(device=gaussians._xyz.device) for _ in range(self.n)]
        self.count = 0               # frames rendered so far (the next frame's global index)
        self._where = {}             # (slot, the slot's serial after the frame) -> global frame index
        self.bad = []

    @property
    def collect(self):
        return self.slots[0].collect

    @collect.setter
    def collect(self, v):
        for sl in self.slots:
            sl.collect = v
            if v:
                sl.bad = []

    def render(self, cam, bg, delta_scale, scaling_modifier=1.0, debug=False):
        g = self.g
        k = self.count % self.n
        sl, st = self.slots[k], self.streams[k]
        dev = g._xyz.device
        # whatever is created lazily and cached for later frames is created on the CALLER's stream, which the slot's stream then
        # waits for: the camera's device copies, the field's processing order (rebuilt every 64 calls)
        cam.device_tensors(dev)
        order = g._deformation.deformation_net.grid._processing_order(g._xyz.detach())
        # ... but only if that stream has anything pending: a marker on it per frame puts traffic on its hardware queue, which one of
        # the slots' streams may share (four hardware queues per device: after a training run -- two streams -- a three-slot pool
        # fell from 5700 to 4400 frames/s through such a collision)
        cur = torch.cuda.current_stream(dev)
        if not cur.query():
            st.wait_stream(cur)
        with torch.cuda.stream(st):
            color, depth, radii = sl.render(cam, bg, delta_scale, scaling_modifier, debug, order=order)
            visible = radii > 0
            ev = torch.cuda.Event()
            ev.record(st)
        self._where[(k, sl.serial)] = self.count
        if len(self._where) > 4096:
            for key in list(self._where)[:2048]:
                del self._where[key]
        self.count += 1
        return color, depth, radii, visible, st, ev

    def zero_points(self):
        """The `viewspace_points` of a no-grad frame: zeros of the model's shape (gaussian_renderer/__init__.py:36 of the
        reference makes them per call; nobody writes them without a backward).  One tensor per model size, made once: a fill kernel
        per frame on the caller's stream is the kind of traffic render() above avoids."""
        z = getattr(self, "_zeros", None)
        if z is None or z.shape != self.g._xyz.shape or z.device != self.g._xyz.device:
            z = self._zeros = torch.zeros_like(self.g._xyz)
        return z

    def overflowed(self, lag=0):
        """Global indices (FusedRenderPool.count at the time of the render() that produced them) of the async-mode frames whose
        binning buffer overflowed: FusedRender.overflowed() of every slot, plus what the slots collected meanwhile."""
        out = []
        for k, sl in enumerate(self.slots):
            serials = list(sl.bad) + sl.overflowed(lag)
            sl.bad = []
            out += [self._where[(k, s)] for s in serials if (k, s) in self._where]
        return sorted(out)
