"""gaussian_renderer.render() under autograd as ONE autograd node.

The reference's own training loop (train_4DGS.py:189-297) calls render(), forms the loss in torch, calls loss.backward() and
optimizer.step().  Through the per-op mirror (HexPlane op, MLP op, activations, rasterizer op ...) that is ~40 small torch
launches and as many autograd nodes per iteration and the host becomes the bottleneck (3.8 ms per iteration at 200 k Gaussians
where the kernels need 1.3 ms).  Here the whole fine-stage forward of one camera is one Function whose forward is the launch
sequence of fused_render.py (with the state a backward needs kept) and whose backward is the second half of fused_step.py:
rasterizer backward -> activations -> MLP -> HexPlane, every parameter gradient returned at once.  The loss, the
regularisers, the statistics and Adam stay where the reference has them.

Every call owns its buffers (they come from torch's caching allocator), so several cameras can be rendered before the
backward -- the reference's batch loop -- and images may be kept.  Binning capacity follows the rasterizer module's sync mode
(diff_gaussian_rasterization._C.set_sync_mode): "exact" waits for the frame's instance count like the reference's cudaMemcpy;
"async" sizes from earlier frames and reports an overflow through the module's sticky flag.
"""
import ctypes as C
import math

import torch

from . import _native as N
from . import ops
from .diff_gaussian_rasterization import _C as RC


DIRECT_GRADS = True      # False: always return the parameter gradients through the autograd graph


class grads_through_graph:
    """Context manager: render() and the regulariser return their parameter gradients to the autograd engine instead of writing
    p.grad themselves.  Needed around torch.autograd.grad() (which must not touch .grad and expects the gradients back); hooks,
    frozen parameters and backward(inputs=...) are detected and need nothing (ops.direct_grads_ok)."""

    def __enter__(self):
        global DIRECT_GRADS
        self._old = (DIRECT_GRADS, ops.PlaneRegFunction.DIRECT_GRADS)
        DIRECT_GRADS = ops.PlaneRegFunction.DIRECT_GRADS = False

    def __exit__(self, *a):
        global DIRECT_GRADS
        DIRECT_GRADS, ops.PlaneRegFunction.DIRECT_GRADS = self._old


# Internal buffers of a render() call (what no caller ever sees: the field's outputs, the rasterizer's geometry and image state, the
# backward's intermediate gradients and scratch) are handed from one call to the next through a free list per (P, W, H, stream)
# instead of going back to torch's allocator and being asked for again: twenty torch.empty calls and as many frees per iteration,
# 40 us of the host time that paces this path.  A call takes a set in its forward and gives it back at the end of its backward; a
# call that is never back-propagated simply keeps its set (the garbage collector frees it), and several cameras rendered before one
# backward each hold a set of their own.  The tensors a caller CAN hold -- image, depth, radii, every gradient -- are never pooled.
_POOL = {}
_POOL_MAX = 4


def _pool_take(key):
    free = _POOL.get(key)
    return free.pop() if free else None


def _pool_give(key, bufs):
    free = _POOL.setdefault(key, [])
    if len(free) < _POOL_MAX:
        free.append(bufs)
    if len(_POOL) > 8:                    # the model changed size a few times (densify / prune): forget the old sizes
        for k in list(_POOL)[:-4]:
            del _POOL[k]


class _State:
    __slots__ = ("pool_key", "bufs", "a", "keep", "P", "W", "H", "cam_time", "order", "porders", "feat", "a0", "pts", "sc_d", "rot_d", "sc", "rot", "op",
                 "color", "depth", "radii", "geom", "img", "binning", "cap", "xyz", "scal", "rotq", "opac", "flow", "coef", "planes",
                 "mlp", "field", "f_dc", "f_rest", "ready", "side")


def applies(cam, pc, pipe, stage, override_color, cam_type):
    """Same conditions as the no-grad fast path: fine stage, the shipped deformation configuration, SH colours and covariances
    computed by the rasterizer, an ordinary camera, everything on the GPU -- and gradients wanted."""
    if not torch.is_grad_enabled() or stage != "fine" or override_color is not None or cam_type == "PanopticSports":
        return False
    if pipe.compute_cov3D_python or pipe.convert_SHs_python or not hasattr(cam, "device_tensors"):
        return False
    if not pc.get_xyz.is_cuda or pc.get_xyz.shape[0] == 0 or ops.BACKEND.name != "hip":
        return False
    if getattr(pipe, "per_op_autograd", False):
        return False
    dn = getattr(pc._deformation, "deformation_net", None)
    return dn is not None and hasattr(dn, "_fusable") and dn._fusable() and pc._features_rest.shape[1] == 15


def field_params(pc):
    """(planes, MLP tensors) of the deformation field, cached on the model until a parameter object is replaced (walking the
    nn.Module tree costs 70 us per call)."""
    dn = pc._deformation.deformation_net
    c = getattr(dn, "_fa_params", None)
    if c is None or c[0] is not dn.grid.grids[0][0] or c[1] is not dn.feature_out[0].weight:
        planes, mlp = [p for lv in dn.grid.grids for p in lv], dn._fused_params()
        c = dn._fa_params = (planes[0], mlp[0], planes, mlp)
    return c[2], c[3]


def _forward_desc(pc, field, planes, mlp):
    """HexPlane / MLP descriptors of the forward (pointers and shapes only): rebuilt when a parameter or the aabb moves."""
    key = (tuple(p.data_ptr() for p in planes), tuple(p.data_ptr() for p in mlp), tuple(field.aabb_host()))
    c = getattr(pc, "_fa_desc", None)
    if c is None or c[0] != key:
        hp, keep = ops._hexplane_desc([[p.detach() for p in lv] for lv in field.grids], field.aabb, None, aabb_host=field.aabb_host())
        md = ops.DeformMLPFunction._desc([p.detach() for p in mlp], None)
        c = pc._fa_desc = (key, hp, keep, md)
    return c[1], c[2], c[3]


def _forward(pc, cam, bg, delta_scale, scaling_modifier, debug):
    lib, s = N.lib(), N.current_stream()
    st = _State()
    dev = pc._xyz.device
    P = st.P = pc._xyz.shape[0]
    W, H = st.W, st.H = int(cam.image_width), int(cam.image_height)
    view, proj, campos, _ = cam.device_tensors(dev)
    dn = pc._deformation.deformation_net
    field = st.field = dn.grid
    st.planes, st.mlp = field_params(pc)
    xyz, scal, rotq, opac = pc._xyz.detach(), pc._scaling.detach(), pc._rotation.detach(), pc._opacity.detach()
    st.f_dc, st.f_rest = pc._features_dc.detach(), pc._features_rest.detach()
    for t in (xyz, scal, rotq, opac, st.f_dc, st.f_rest):
        if not t.is_contiguous():
            raise N.MomError("fused render(): the Gaussian parameters must be contiguous")
    flow = pc._scene_flow if pc._scene_flow.is_contiguous() else pc._scene_flow.contiguous()
    st.xyz, st.scal, st.rotq, st.opac, st.flow = xyz, scal, rotq, opac, flow
    st.cam_time = float(cam.time)
    st.coef = float(delta_scale * cam.frame_num)
    st.order = field._processing_order(xyz)
    st.porders = field._plane_orders(xyz)
    f = dict(dtype=torch.float32, device=dev)
    e = lambda *sh: torch.empty(*sh, **f)
    st.pool_key = (P, W, H, dev, s)
    b = st.bufs = _pool_take(st.pool_key)
    if b is None:
        b = st.bufs = {"feat": e(P, 64), "a0": e(P, 64), "pts": e(P, 3), "sc_d": e(P, 3), "rot_d": e(P, 4), "sc": e(P, 3),
                       "rot": e(P, 4), "op": e(P, 1),
                       "geom": torch.empty(lib.mom_raster_geom_bytes(P), dtype=torch.uint8, device=dev),
                       "img": torch.empty(lib.mom_raster_image_bytes(W, H), dtype=torch.uint8, device=dev),
                       "nr_dev": torch.empty(1, dtype=torch.int32, device=dev)}
    st.feat, st.a0 = b["feat"], b["a0"]
    st.pts, st.sc_d, st.rot_d = b["pts"], b["sc_d"], b["rot_d"]
    st.sc, st.rot, st.op = b["sc"], b["rot"], b["op"]
    st.color, st.depth = e(3, H, W), e(1, H, W)
    st.radii = torch.empty(P, dtype=torch.int32, device=dev)
    st.geom, st.img = b["geom"], b["img"]
    hp, keep, md = _forward_desc(pc, field, st.planes, st.mlp)
    ops.field_forward(hp, md, P, xyz, st.cam_time, st.order, scal, rotq, flow, st.coef, st.pts, st.sc_d, st.rot_d, st.feat, st.a0,
                      opac, st.sc, st.rot, st.op, s)
    a = st.a = N.MomRasterArgs()
    a.P, a.D, a.M, a.W, a.H = P, pc.active_sh_degree, 16, W, H
    a.background, a.means3D = bg.data_ptr(), st.pts.data_ptr()
    a.shs, a.shs_rest = st.f_dc.data_ptr(), st.f_rest.data_ptr()
    a.colors_precomp, a.opacities = None, st.op.data_ptr()
    a.scales, a.rotations, a.cov3D_precomp = st.sc.data_ptr(), st.rot.data_ptr(), None
    a.viewmatrix, a.projmatrix, a.campos = view.data_ptr(), proj.data_ptr(), campos.data_ptr()
    a.scale_modifier = float(scaling_modifier)
    a.tan_fovx, a.tan_fovy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    a.prefiltered, a.debug = 0, int(bool(debug))
    a.keep_all_tiles = int(RC._state["keep_all_tiles"])       # set_keep_all_tiles(): the reference's lists and num_rendered
    st.keep = (bg, view, proj, campos, keep)
    nr_dev = b["nr_dev"]
    nr_host = RC.pinned_word()
    N.check(lib.mom_raster_forward_geometry(C.byref(a), st.geom.data_ptr(), st.img.data_ptr(), st.radii.data_ptr(), nr_dev.data_ptr(),
                                            nr_host.data_ptr(), s), "raster_geometry")
    # binning capacity exactly as diff_gaussian_rasterization._C.rasterize_gaussians sizes it
    state = RC._state
    # async with nothing to size from (no hint, no earlier frame): this one forward waits for its count, like exact mode
    blind = state["mode"] == "async" and state["cap_hint"] == 0 and state["last_R"] is None
    if state["mode"] == "exact" or blind:
        # the drop-in's default: the frame's own count decides, but the compositing is enqueued before the host waits for it
        # (RC.exact_render)
        def regeometry():
            N.check(lib.mom_raster_forward_geometry(C.byref(a), st.geom.data_ptr(), st.img.data_ptr(), st.radii.data_ptr(),
                                                    nr_dev.data_ptr(), nr_host.data_ptr(), s), "raster_geometry")
        count, st.binning = RC.exact_render(lib, a, st.geom, st.img, st.color, st.depth, nr_host, P, W, H, dev, s, regeometry)
        cap, flag = state["exact_last_capacity"], None
        if blind:
            state["cap_hint"] = int(count * 1.5) + 4096
        state["last_R"] = nr_host
        st.cap = cap
    else:
        flag = RC.overflow_flag(dev)
        RC._check_overflow(RC._FLAG_LAG)
        prev = state["last_R"]
        if prev is not None and int(prev[0]) != RC.COUNT_PENDING:
            state["cap_hint"] = max(state["cap_hint"], int(int(prev[0]) * 1.5) + 4096)
        cap = max(state["cap_hint"], 4096)
        state["last_R"] = nr_host
        st.cap = cap
        st.binning = torch.empty(lib.mom_raster_binning_bytes(P, W, H, cap), dtype=torch.uint8, device=dev)
        # the frame's status bits come back through a pinned word the compositing kernel writes (RC.post_slot): no copy, no event
        state["serial"] += 1
        a.status_post, slot = RC.post_slot(state["serial"])
        a.status_serial = state["serial"] & 0xFFFFFFFF
        N.check(lib.mom_raster_forward_render(C.byref(a), st.geom.data_ptr(), st.binning.data_ptr(), cap, st.img.data_ptr(),
                                              st.color.data_ptr(), st.depth.data_ptr(), flag.data_ptr(), s), "raster_render")
        state["pending"].append((slot, nr_host, state["serial"]))
    return st


def _field_grads(st, f, direct=True):
    """Gradient storage of the deformation field -- the planes in their channel-last storage order, then the MLP tensors, one
    flat zeroed buffer -- with the backward descriptors that point into it.  One set is cached on the field and reused from
    iteration to iteration (the reference's loop drops the gradients with zero_grad(set_to_none=True), so the parameters let go
    of it); while the parameters still hold it -- a second camera of a batch -- a temporary set is made.

    Planes that already HAVE a gradient of the right layout (the regulariser's backward ran first) are accumulated into in
    place: the kernels add with atomics anyway, and twelve elementwise additions and their launches are saved.
    Returns (plane targets, MLP targets, hp, md, in_place) -- in_place[i]: plane i's target is its own .grad."""
    field = st.field
    # through-the-graph mode (direct False): never touch a .grad, never hand out the cached buffer
    held = [p.grad if direct else None for p in st.planes]
    in_place = ops.in_place_flags(held, st.planes)
    key = (tuple(p.data_ptr() for p in st.planes), tuple(p.data_ptr() for p in st.mlp), tuple(field.aabb_host()),
           tuple(g.data_ptr() if ip else 0 for g, ip in zip(held, in_place)))
    c = getattr(field, "_fa_grads", None)
    own_busy = c is not None and (not direct or (st.mlp[0].grad is not None and st.mlp[0].grad.data_ptr() == c[3][0].data_ptr())
                                  or any(g is not None and g.data_ptr() == v.data_ptr() for g, v in zip(held, c[2]))
                                  or not ops._buffers_free(c[3]) or not ops._buffers_free(c[2], 2))   # a caller kept last iteration's gradient tensors (planes: also in `targets`)
    if c is not None and c[0] == key and not own_busy:
        c[1].zero_()
        return [held[i] if in_place[i] else c[2][i] for i in range(len(c[2]))], c[3], c[5], c[7], in_place
    n = sum(p.numel() for p in st.planes + st.mlp)
    flat = torch.zeros(n, **f)
    off, gplanes, gmlp = 0, [], []
    for p in st.planes:
        shape = ops.plane_storage(p).shape
        gplanes.append(flat[off:off + p.numel()].view(shape).permute(2, 0, 1).unsqueeze(0))
        off += p.numel()
    for p in st.mlp:
        gmlp.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    targets = [held[i] if in_place[i] else gplanes[i] for i in range(len(gplanes))]
    levels, k = [], 0
    for lv in field.grids:
        levels.append(targets[k:k + len(lv)])
        k += len(lv)
    hp, keep = ops._hexplane_desc([[p.detach() for p in lv] for lv in field.grids], field.aabb, levels, aabb_host=field.aabb_host())
    md = ops.DeformMLPFunction._desc([p.detach() for p in st.mlp], gmlp)
    if not own_busy:
        # (no reference to `held`: those are the regulariser's cached gradient views, and a reference kept here made ITS cache
        # look busy in the next iteration -- it then made new buffers, this key changed, and both caches missed every iteration:
        # 1 ms of host time per step, found with tools/host_profile.py --autograd)
        field._fa_grads = (key, flat, gplanes, gmlp, None, hp, keep, md)
    return targets, gmlp, hp, md, in_place


def _backward(st, dcolor, ddepth, direct=True):
    lib, s = N.lib(), N.current_stream()
    P, dev = st.P, st.color.device
    f = dict(dtype=torch.float32, device=dev)
    e = lambda *sh: torch.empty(*sh, **f)
    dcol = dcolor.contiguous().float()
    ddep = None if ddepth is None else ddepth.contiguous().float()
    b = st.bufs
    if "gcol" not in b:               # the backward's internal gradients and scratch: made once per pooled set
        b.update(gcol=e(P, 3), gcov=e(P, 6), dfeat=e(P, 64),
                 scratch=torch.empty(lib.mom_deform_backward_scratch_bytes(P), dtype=torch.uint8, device=dev))
    g2d, gcol, gxyz, gcov = e(P, 3), b["gcol"], e(P, 3), b["gcov"]
    gdc, grest = e(P, 1, 3), e(P, 15, 3)
    gsc, grot, gop = e(P, 3), e(P, 4), e(P, 1)
    gr = N.MomRasterGrads()
    # (scale / rotation / opacity gradients: through their activations inside the projection backward -- act_rotations_raw)
    gr.dL_dmeans2D, gr.dL_dcolors, gr.dL_dopacity = g2d.data_ptr(), gcol.data_ptr(), gop.data_ptr()
    gr.dL_dmeans3D, gr.dL_dcov3D = gxyz.data_ptr(), gcov.data_ptr()
    gr.dL_dsh, gr.dL_dsh_rest = gdc.data_ptr(), grest.data_ptr()
    gr.dL_dscales, gr.dL_drotations = gsc.data_ptr(), grot.data_ptr()
    gr.act_rotations_raw = st.rot_d.data_ptr()
    N.check(lib.mom_raster_backward(C.byref(st.a), st.radii.data_ptr(), st.geom.data_ptr(), st.binning.data_ptr(), st.cap,
                                    st.img.data_ptr(), dcol.data_ptr(), None if ddep is None else ddep.data_ptr(), C.byref(gr), s),
            "raster_bwd")
    overlap = ops.API_OVERLAP and direct
    ready = side = None
    if overlap:
        # the appearance parameters' gradients (SH, scaling, rotation, opacity) are final here: FusedAdam.step() may start their
        # update on the second stream behind this event while the deformation backward below still runs (ops.FusedAdam.step)
        side = ops.side_stream(dev).cuda_stream
        ready = ops.next_ring_mark(s)
    gplanes, gmlp, hp, md, in_place = _field_grads(st, f, direct)
    dfeat, scratch = b["dfeat"], b["scratch"]
    # pts = xyz + dx(...): d xyz starts as d pts (already in gxyz); scale / rotation residuals likewise
    if overlap:
        # with a second stream the MLP backward leaves an eighth of the chip free (for that Adam launch) and its partial-sum
        # reduction goes there too (csrc/deform_bwd_b3.hip); joined below
        N.check(lib.mom_deform_backward_split(C.byref(md), P, st.feat.data_ptr(), st.a0.data_ptr(), gxyz.data_ptr(), gsc.data_ptr(),
                                              grot.data_ptr(), dfeat.data_ptr(), scratch.data_ptr(), s, side), "deform_bwd")
    else:
        N.check(lib.mom_deform_backward(C.byref(md), P, st.feat.data_ptr(), st.a0.data_ptr(), gxyz.data_ptr(), gsc.data_ptr(),
                                        grot.data_ptr(), dfeat.data_ptr(), scratch.data_ptr(), s), "deform_bwd")
    ops.wait_reg_pending(dev)          # the regulariser's gradient kernel (second stream) adds into the plane gradients too
    porders = st.porders
    hscratch = None
    if porders is not None:
        need = lib.mom_hexplane_backward_scratch_bytes(C.byref(hp), P)
        hscratch = b.get("hscratch")
        if hscratch is None or hscratch.numel() < need:
            hscratch = b["hscratch"] = torch.empty(need, dtype=torch.uint8, device=dev)
    N.check(lib.mom_hexplane_backward(C.byref(hp), P, st.xyz.data_ptr(), None, st.cam_time,
                                      None if st.order is None else st.order.data_ptr(), dfeat.data_ptr(), gxyz.data_ptr(),
                                      None if porders is None else porders[0].data_ptr(),
                                      None if porders is None else porders[1].data_ptr(),
                                      None if hscratch is None else hscratch.data_ptr(), s), "hexplane_bwd")
    if overlap:
        ops.stream_wait_stream(s, side)                    # the MLP weight gradients are complete for whoever reads them next
    st.ready, st.side = ready, side
    # the set goes back to the free list: everything that reads it is enqueued on this stream, and so is whoever takes it next
    # (the key holds the stream).  With the second-stream overlap a kernel on the OTHER stream may still read it: that call keeps it.
    if not overlap:
        _pool_give(st.pool_key, b)
    st.bufs = None
    return g2d, gxyz, gdc, grest, gsc, grot, gop, gplanes, gmlp, in_place


class FusedRenderFunction(torch.autograd.Function):
    """(image, depth, radii) = render of one camera; inputs after the six plain arguments: the 2-D gradient holder, the six
    Gaussian parameter tensors, the 12 HexPlane planes, the 14 MLP tensors."""

    @staticmethod
    def forward(ctx, pc, cam, bg, delta_scale, scaling_modifier, debug, screenspace, xyz, f_dc, f_rest, scaling, rotation, opacity,
                *field):
        st = _forward(pc, cam, bg, delta_scale, scaling_modifier, debug)
        ctx.st, ctx.pc = st, pc
        dev = st.color.device
        ops._render_pending[dev] = ops._render_pending.get(dev, 0) + 1       # (ops.PlaneRegFunction.backward: who joins its kernel)
        ctx.mark_non_differentiable(st.radii)
        return st.color, st.depth, st.radii

    @staticmethod
    def backward(ctx, dcolor, ddepth, _dradii):
        st = ctx.st
        if st is None:
            raise RuntimeError("render(): a second backward through the same call -- its buffers were released after the first "
                               "(render again, or set pipe.per_op_autograd = True for retain_graph use)")
        ctx.st = None                                  # the call's buffers go back to the allocator after this backward
        dev = st.color.device
        ops._render_pending[dev] = max(0, ops._render_pending.get(dev, 0) - 1)
        pc = ctx.pc
        params = (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity, *st.planes, *st.mlp)
        needs = ctx.needs_input_grad[7:]
        direct = DIRECT_GRADS and ops.direct_grads_ok(params, needs)
        g2d, gxyz, gdc, grest, gsc, grot, gop, gplanes, gmlp, in_place = _backward(st, dcolor, ddepth, direct)
        if direct:
            # The 32 parameter gradients are handed to the parameters here (set, or added to what an earlier camera of the
            # batch left) instead of being returned: 32 AccumulateGrad nodes cost the autograd engine more host time than the
            # whole forward.  Only the 2-D gradient holder, a non-leaf, goes back through the graph.
            fresh = True
            for p, g in zip(params, (gxyz, gdc, grest, gsc, grot, gop, *gplanes, *gmlp)):
                if p.grad is None:
                    p.grad = g
                elif p.grad is not g:                   # (a plane accumulated into in place IS its own gradient)
                    p.grad.add_(g)
                    fresh = False
            # one camera per optimizer step (the reference's batch_size 1): tell the optimizer when the appearance gradients were
            # final, so that its step() can start their update underneath the rest of this backward (ops.FusedAdam.step checks that
            # nothing touched them in between).  A second camera accumulating into them withdraws the hint.
            opt = getattr(pc, "optimizer", None)
            if opt is not None and hasattr(opt, "early_hint"):
                app = (pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity)
                if st.ready is not None and fresh and all(p.grad is g for p, g in zip(app, (gdc, grest, gsc, grot, gop))):
                    opt.early_hint = (st.ready, st.side, [(p, p.grad, p.grad._version) for p in app])
                else:
                    opt.early_hint = None
            return (None, None, None, None, None, None, g2d) + (None,) * (6 + len(gplanes) + len(gmlp))
        # through the graph (fresh buffers: nothing is accumulated in place on this path); inputs that were not asked for get None
        out = [g if n else None for g, n in zip((gxyz, gdc, grest, gsc, grot, gop, *gplanes, *gmlp), needs)]
        return (None, None, None, None, None, None, g2d, *out)


def render(cam, pc, pipe, bg, delta_scale, scaling_modifier, screenspace_points):
    planes, mlp = field_params(pc)
    return FusedRenderFunction.apply(pc, cam, bg, delta_scale, scaling_modifier, pipe.debug,
                                     screenspace_points, pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation,
                                     pc._opacity, *planes, *mlp)
