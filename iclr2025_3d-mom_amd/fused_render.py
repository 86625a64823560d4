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
        # per-frame status word of the image scratch (header word 1: bit 0 = this frame's binning overflowed), copied to a
        # pinned ring behind an event after every async-mode frame
        self.hdr = self.img[(-self.img.data_ptr()) % 256:][:8].view(torch.int32)
        self.flag_ring = torch.zeros(self.RING, dtype=torch.int32).pin_memory()
        self.pending = deque()               # (frame serial, ring slot, event), oldest first
        self.cap, self.binning = 0, None

    RING, FLAG_LAG = 64, 8
    HEADROOM, MARGIN = 1.5, 65536       # async binning capacity = HEADROOM x an earlier frame's instance count + MARGIN

    def overflowed(self, lag=0):
        """Serial numbers (FusedRender.serial after the render() that produced them) of async-mode frames older than `lag`
        frames whose binning buffer overflowed -- their images are incomplete and must be rendered again.  lag=0 waits for
        everything rendered so far.  The capacity is raised so that a repeat fits."""
        bad = []
        while len(self.pending) > lag:
            serial, slot, ev = self.pending.popleft()
            ev.synchronize()
            if int(self.flag_ring[slot]) & 1:
                bad.append(serial)
        if bad:
            self.cap_floor = max(getattr(self, "cap_floor", 0), 2 * self.cap)
        return bad

    def render(self, cam, bg, delta_scale, scaling_modifier=1.0, debug=False):
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
        ops.field_forward(hp, md, P, xyz, float(cam.time), order, scal, rot, flow, float(delta_scale * cam.frame_num), self.pts,
                          self.sc_d, self.rot_d, None, None, opac, self.sc, self.rot, self.op, s, scratch_feat=self.feat)
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
        N.check(lib.mom_raster_forward_geometry(C.byref(a), self.geom.data_ptr(), self.img.data_ptr(), radii.data_ptr(),
                                                self.nr_dev.data_ptr(), self.nr_host.data_ptr(), s), "raster_geometry")
        if RC._state["mode"] == "exact" or self.cap == 0:
            torch.cuda.current_stream().synchronize()
            want = int(self.nr_host[0]) + (0 if RC._state["mode"] == "exact" else int(self.nr_host[0]) // 2 + 65536)
        else:
            want = max(self.cap, int(prev_R * self.HEADROOM) + self.MARGIN, getattr(self, "cap_floor", 0))
        if self.binning is None or want > self.cap or want < self.cap // 4:
            self.cap = want
            self.binning = torch.empty(lib.mom_raster_binning_bytes(P, W, H, self.cap), dtype=torch.uint8, device=dev)
        N.check(lib.mom_raster_forward_render(C.byref(a), self.geom.data_ptr(), self.binning.data_ptr(), self.cap,
                                              self.img.data_ptr(), color.data_ptr(), depth.data_ptr(),
                                              None, s), "raster_render")
        self.serial += 1
        if RC._state["mode"] != "exact":
            slot = self.serial % self.RING
            self.flag_ring[slot:slot + 1].copy_(self.hdr[1:2], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending.append((self.serial, slot, ev))
        return color, depth, radii
