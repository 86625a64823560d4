"""torch-facing wrappers (autograd Functions, optimizer) around the libmom4d C ABI.

Every op here runs on the GPU through hand-written HIP; given CPU tensors they raise (no fallback).
CPU-only tests of the host logic swap `BACKEND` for the oracle's torch restatement explicitly."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _native as N


def _need_cuda(t, what):
    if not t.is_cuda:
        raise N.MomError(f"{what}: libmom4d has no CPU path (got a {t.device} tensor)")


# --------------------------------------------------------------------------- HexPlane
def plane_storage(p):
    """Channel-last [H,W,C] storage view of a logical [1,C,H,W] plane parameter (or None if it is not
    laid out that way)."""
    if p.dim() == 4 and p.shape[0] == 1:
        v = p[0].permute(1, 2, 0)
        if v.is_contiguous():
            return v
    return None


def make_plane(C_, H, W, device=None):
    """A [1,C,H,W] tensor whose memory is channel-last ([H][W][C]) -- the layout libmom4d's kernels read."""
    return torch.empty(1, H, W, C_, device=device).permute(0, 3, 1, 2)


def _hexplane_desc(planes_by_level, aabb, grads_by_level=None, aabb_host=None):
    """MomHexPlane descriptor.  `aabb_host`: the six aabb floats already on the host (HexPlaneField.aabb_host());
    without it they are read back from the `aabb` tensor, which blocks the host until the GPU has drained."""
    d = N.MomHexPlane()
    d.levels = len(planes_by_level)
    d.channels = planes_by_level[0][0].shape[1]
    keep = []
    for l, planes in enumerate(planes_by_level):
        # resolutions: plane (a,b) has W = res[a], H = res[b]
        res = [0, 0, 0, 0]
        for p, (a, b) in enumerate(((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))):
            st = plane_storage(planes[p])
            if st is None:
                raise N.MomError("HexPlane planes must be channel-last (use ops.make_plane)")
            res[a], res[b] = st.shape[1], st.shape[0]
            d.planes[l][p] = st.data_ptr()
            keep.append(st)
            if grads_by_level is not None:
                gs = plane_storage(grads_by_level[l][p])
                d.grads[l][p] = gs.data_ptr()
                keep.append(gs)
        for k in range(4):
            d.res[l][k] = res[k]
    a = aabb_host if aabb_host is not None else aabb.detach().float().cpu().reshape(-1).tolist()
    for k in range(6):
        d.aabb[k] = a[k]
    return d, keep


class HexPlaneFunction(torch.autograd.Function):
    """features[P, L*32] = HexPlaneField(xyz, t) (reference scene/hexplane.py:160-183)."""

    @staticmethod
    def forward(ctx, xyz, time, aabb, n_levels, order, aabb_host, plane_orders, *planes):
        _need_cuda(xyz, "hexplane")
        ctx.plane_orders = plane_orders
        lv = [list(planes[6 * l:6 * l + 6]) for l in range(n_levels)]
        d, keep = _hexplane_desc(lv, aabb, aabb_host=aabb_host)
        ctx.aabb_host = aabb_host
        xyz_c = xyz.detach().contiguous().float()
        P = xyz_c.shape[0]
        feat = torch.empty((P, n_levels * 32), dtype=torch.float32, device=xyz.device)
        # one timestamp per camera (a python float) or per-point timestamps (a tensor, as the reference passes)
        times = time.detach().reshape(-1).contiguous().float() if torch.is_tensor(time) else None
        tval = 0.0 if times is not None else float(time)
        optr = None if order is None else order.data_ptr()
        N.check(N.lib().mom_hexplane_forward(C.byref(d), P, xyz_c.data_ptr(), None if times is None else times.data_ptr(),
                                             tval, optr, feat.data_ptr(), N.current_stream()), "mom_hexplane_forward")
        ctx.save_for_backward(xyz_c, aabb, *planes)
        ctx.times, ctx.time, ctx.n_levels, ctx.order = times, tval, n_levels, order
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        xyz_c, aabb, *planes = ctx.saved_tensors
        n_levels = ctx.n_levels
        wait_reg_pending_all()         # (AccumulateGrad adds what this returns to plane gradients the second stream may be writing)
        lv = [list(planes[6 * l:6 * l + 6]) for l in range(n_levels)]
        grads = [[torch.zeros_like(p) for p in level] for level in lv]   # preserves the channel-last strides
        d, keep = _hexplane_desc(lv, aabb, grads, aabb_host=ctx.aabb_host)
        P = xyz_c.shape[0]
        dxyz = torch.zeros_like(xyz_c) if ctx.needs_input_grad[0] else None
        dfeat = dfeat.contiguous()
        lib = N.lib()
        po, scratch = ctx.plane_orders, None
        if po is not None and ctx.times is None:        # two-pass backward: per-plane orders + the gradient-row scratch
            scratch = torch.empty(lib.mom_hexplane_backward_scratch_bytes(C.byref(d), P), dtype=torch.uint8, device=xyz_c.device)
        N.check(lib.mom_hexplane_backward(C.byref(d), P, xyz_c.data_ptr(),
                                          None if ctx.times is None else ctx.times.data_ptr(), ctx.time,
                                          None if ctx.order is None else ctx.order.data_ptr(), dfeat.data_ptr(),
                                          None if dxyz is None else dxyz.data_ptr(),
                                          None if scratch is None else po[0].data_ptr(), None if scratch is None else po[1].data_ptr(),
                                          None if scratch is None else scratch.data_ptr(), N.current_stream()),
                "mom_hexplane_backward")
        flat = [g for level in grads for g in level]
        return (dxyz, None, None, None, None, None, None, *flat)


def hexplane_features(xyz, time, aabb, planes_by_level, order=None, aabb_host=None, plane_orders=None):
    flat = [p for level in planes_by_level for p in level]
    return HexPlaneFunction.apply(xyz, time, aabb, len(planes_by_level), order, aabb_host, plane_orders, *flat)


def hexplane_orders(xyz, planes_by_level, aabb, aabb_host=None, stream=None, keepalive=None):
    """(order, inverse), int32 [3, levels, P] each: per space plane and level the permutation that sorts the points by that
    level's texel cell, and its inverse (mom_hexplane_orders) -- what the two-pass HexPlane backward walks.  Speed only.
    stream: a raw stream handle to launch on (default: the current stream); `keepalive`: a list that receives everything the launch
    refers to (scratch, the point tensor, the descriptor's tensors), for a caller that launches on another stream and must keep it
    alive until that work is done."""
    _need_cuda(xyz, "hexplane_orders")
    lib = N.lib()
    pts = xyz.detach().contiguous().float()
    P = pts.shape[0]
    d, keep = _hexplane_desc([[p.detach() for p in lv] for lv in planes_by_level], aabb, aabb_host=aabb_host)
    order = torch.empty((3, d.levels, max(P, 1)), dtype=torch.int32, device=pts.device)
    inverse = torch.empty((3, d.levels, max(P, 1)), dtype=torch.int32, device=pts.device)
    if P:
        scratch = torch.empty(lib.mom_hexplane_orders_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
        N.check(lib.mom_hexplane_orders(C.byref(d), P, pts.data_ptr(), order.data_ptr(), inverse.data_ptr(), scratch.data_ptr(),
                                        N.current_stream() if stream is None else stream), "mom_hexplane_orders")
        if keepalive is not None:
            keepalive += [scratch, pts, keep]
    return order, inverse


def morton_order(xyz, stream=None, keepalive=None):
    """uint32 permutation that walks the points along a Morton curve (int32 tensor of the same bits).  stream / keepalive: as in
    hexplane_orders."""
    _need_cuda(xyz, "morton_order")
    lib = N.lib()
    pts = xyz.detach().contiguous().float()
    P = pts.shape[0]
    order = torch.empty(P, dtype=torch.int32, device=pts.device)
    if P:
        scratch = torch.empty(lib.mom_morton_order_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
        N.check(lib.mom_morton_order(P, pts.data_ptr(), order.data_ptr(), scratch.data_ptr(),
                                     N.current_stream() if stream is None else stream), "mom_morton_order")
        if keepalive is not None:
            keepalive += [scratch, pts]
    return order


# --------------------------------------------------------------------------- deformation field in one pass
import os as _os

FUSE_FIELD = _os.environ.get("MOM_FUSE_FIELD", "1") != "0"     # HexPlane lookup fused into the MLP kernels (csrc/deform_field.hip)
FIELD_ORDER = _os.environ.get("MOM_FIELD_ORDER", "1") != "0"   # the fused kernels walk the Morton order (else the index order)
_field_scratch = {}


def field_scratch(hp, device, P=0):
    """Scratch of the fused deformation kernels, one per device: the per-frame table of time lines (73 KB at the shipped
    resolutions) and, for calls that keep no copy of the features (P > 0: no-grad render()), a [P,64] feature buffer."""
    n = N.lib().mom_deform_field_scratch_bytes(C.byref(hp), int(P))
    t = _field_scratch.get(device)
    if t is None or t.numel() < n:
        t = _field_scratch[device] = torch.empty(n, dtype=torch.uint8, device=device)
    return t


def field_forward(hp, md, P, xyz, time, order, scal, rot, flow, coef, pts, sc_d, rot_d, feat, a0, opac, sc, rot_act, op, s,
                  scratch_feat=None, scratch=None):
    """deform_network.forward for one timestamp: the fused kernel when the field's shape allows it, else HexPlane + MLP.
    feat / a0: [P,64] tensors that receive the features and relu(h0) for the backward, or None (no backward follows; the
    two-kernel path then needs `scratch_feat` for the features).
    scratch: the caller's own field scratch (mom_deform_field_scratch_bytes) -- callers that keep several frames in flight on
    different streams must not share the per-device one (the time-line table and the feature buffer live for the whole launch)."""
    lib = N.lib()
    q = lambda t: None if t is None else t.data_ptr()
    if FUSE_FIELD and lib.mom_deform_field_supported(C.byref(hp)):
        N.check(lib.mom_deform_field_forward(C.byref(hp), C.byref(md), P, xyz.data_ptr(), float(time),
                                             q(order) if FIELD_ORDER else None, scal.data_ptr(),
                                             rot.data_ptr(), flow.data_ptr(), float(coef), pts.data_ptr(), sc_d.data_ptr(),
                                             rot_d.data_ptr(), q(feat), q(a0), q(opac), q(sc), q(rot_act), q(op),
                                             (scratch if scratch is not None else
                                              field_scratch(hp, xyz.device, 0 if feat is not None else P)).data_ptr(), s), "deform_field_fwd")
        return True          # the device's field scratch now holds this frame's time lines (mom_hexplane_backward_lines)
    f = feat if feat is not None else scratch_feat
    N.check(lib.mom_hexplane_forward(C.byref(hp), P, xyz.data_ptr(), None, float(time), q(order), f.data_ptr(), s), "hexplane_fwd")
    N.check(lib.mom_deform_forward_activated(C.byref(md), P, f.data_ptr(), xyz.data_ptr(), scal.data_ptr(), rot.data_ptr(),
                                             flow.data_ptr(), float(coef), pts.data_ptr(), sc_d.data_ptr(), rot_d.data_ptr(), q(a0),
                                             q(opac), q(sc), q(rot_act), q(op), s), "deform_fwd")
    return False


# --------------------------------------------------------------------------- fused deformation MLP
class DeformMLPFunction(torch.autograd.Function):
    """(pts, scales, rots) = fused trunk + pos/scales/rotations heads + residual adds
    (reference scene/deformation.py:97-135 with the shipped config).  Parameter order:
    W0,b0, W1p,b1p,W2p,b2p, W1s,b1s,W2s,b2s, W1r,b1r,W2r,b2r."""

    @staticmethod
    def _desc(params, grads=None):
        d = N.MomDeformMLP()
        W0, b0, *rest = params
        d.W0, d.b0 = W0.data_ptr(), b0.data_ptr()
        for k in range(3):
            W1, b1, W2, b2 = rest[4 * k:4 * k + 4]
            d.W1[k], d.b1[k], d.W2[k], d.b2[k] = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr()
        if grads is not None:
            gW0, gb0, *grest = grads
            d.dW0, d.db0 = gW0.data_ptr(), gb0.data_ptr()
            for k in range(3):
                g = grest[4 * k:4 * k + 4]
                d.dW1[k], d.db1[k], d.dW2[k], d.db2[k] = (t.data_ptr() for t in g)
        return d

    @staticmethod
    def forward(ctx, feat, xyz, scaling, rotation, scene_flow, flow_coef, *params):
        _need_cuda(feat, "deform_mlp")
        lib = N.lib()
        P = feat.shape[0]
        ps = [p.detach().contiguous() for p in params]
        feat_c = feat.detach().contiguous()
        d = DeformMLPFunction._desc(ps)
        s = N.current_stream()
        pts = torch.empty_like(xyz)
        sc = torch.empty_like(scaling)
        ro = torch.empty_like(rotation)
        xyz_c, scal_c, rot_c, flow_c = (t.detach().contiguous() for t in (xyz, scaling, rotation, scene_flow))
        need_bwd = any(ctx.needs_input_grad)
        a0 = torch.empty_like(feat_c) if need_bwd else None
        N.check(lib.mom_deform_forward(C.byref(d), P, feat_c.data_ptr(), xyz_c.data_ptr(), scal_c.data_ptr(), rot_c.data_ptr(),
                                       flow_c.data_ptr(), float(flow_coef), pts.data_ptr(), sc.data_ptr(), ro.data_ptr(),
                                       None if a0 is None else a0.data_ptr(), s), "mom_deform_forward")
        ctx.save_for_backward(feat_c, a0 if a0 is not None else feat_c, *ps)
        return pts, sc, ro

    @staticmethod
    def backward(ctx, dpts, dsc, dro):
        feat_c, a0, *ps = ctx.saved_tensors
        lib = N.lib()
        P = feat_c.shape[0]
        grads = [torch.zeros_like(p) for p in ps]
        d = DeformMLPFunction._desc(ps, grads)
        dpts, dsc, dro = dpts.contiguous(), dsc.contiguous(), dro.contiguous()
        dfeat = torch.empty_like(feat_c)
        scratch = torch.empty(lib.mom_deform_backward_scratch_bytes(P), dtype=torch.uint8, device=feat_c.device)
        N.check(lib.mom_deform_backward(C.byref(d), P, feat_c.data_ptr(), a0.data_ptr(), dpts.data_ptr(), dsc.data_ptr(),
                                        dro.data_ptr(), dfeat.data_ptr(), scratch.data_ptr(), N.current_stream()),
                "mom_deform_backward")
        # identity paths: pts = xyz + ..., scales = scaling + ..., rots = rotation + ...; scene_flow has no grad
        return (dfeat, dpts, dsc, dro, None, None, *grads)


def deform_mlp(feat, xyz, scaling, rotation, scene_flow, flow_coef, params):
    return DeformMLPFunction.apply(feat, xyz, scaling, rotation, scene_flow, flow_coef, *params)


# --------------------------------------------------------------------------- L1 + PSNR
class L1LossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt):
        _need_cuda(img, "l1_loss")
        a = img.detach().contiguous().float()
        b = gt.detach().contiguous().float()
        dimg = torch.empty_like(a)
        sums = torch.empty(2, dtype=torch.float32, device=a.device)
        N.check(N.lib().mom_l1_loss(a.numel(), a.data_ptr(), b.data_ptr(), dimg.data_ptr(), sums.data_ptr(),
                                    N.current_stream()), "mom_l1_loss")
        ctx.save_for_backward(dimg)
        ctx.mark_non_differentiable(sums)
        return sums[0] / a.numel(), sums

    @staticmethod
    def backward(ctx, g, _):
        (dimg,) = ctx.saved_tensors
        return dimg * g, None


def l1_loss_with_sums(img, gt):
    """(mean |img-gt|, [sum|d|, sum d^2]) in one pass; the second output feeds psnr without re-reading the images."""
    return L1LossFunction.apply(img, gt)


# --------------------------------------------------------------------------- row selection (densify / prune)
def select_rows(mask, tensors):
    """[t[mask] for t in tensors] -- the selection the reference's optimizer surgery does per tensor (scene/gaussian_model.py:
    409-482, 511-581) -- through ONE compaction plan: the mask is scanned once, one host synchronisation reads the number of
    kept rows (the outputs have to be allocated), and one kernel gathers the rows of every tensor."""
    _need_cuda(mask, "select_rows")
    n = mask.shape[0]
    if mask.dim() != 1 or mask.dtype != torch.bool:
        raise N.MomError("select_rows: the mask must be a 1-D bool tensor")
    srcs = []
    for t_ in tensors:
        if t_.shape[0] != n or not t_.is_cuda:
            raise N.MomError("select_rows: every tensor needs one row per mask element, on the GPU")
        srcs.append(t_.detach().contiguous())
    lib, s = N.lib(), N.current_stream()
    dev = mask.device
    keep = mask.contiguous().view(torch.uint8)
    dst_index = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    count_dev = torch.empty(1, dtype=torch.int32, device=dev)
    count_host = torch.empty(1, dtype=torch.int32).pin_memory()
    scratch = torch.empty(lib.mom_select_scratch_bytes(n), dtype=torch.uint8, device=dev)
    N.check(lib.mom_select_plan(n, keep.data_ptr(), dst_index.data_ptr(), count_dev.data_ptr(), count_host.data_ptr(),
                                scratch.data_ptr(), s), "mom_select_plan")
    torch.cuda.current_stream().synchronize()          # the one sync of the round: sizes of the outputs
    m = int(count_host[0])
    outs = [torch.empty((m,) + tuple(t_.shape[1:]), dtype=t_.dtype, device=dev) for t_ in srcs]
    if m == 0:                                           # nothing kept: empty outputs (their data pointers are null)
        return outs
    for lo in range(0, len(srcs), N.SELECT_MAX_TENSORS):
        part = list(zip(srcs, outs))[lo:lo + N.SELECT_MAX_TENSORS]
        arr = (N.MomRowSelect * len(part))()
        for i, (a, b) in enumerate(part):
            rb = (a.numel() // max(n, 1)) * a.element_size() if n else 0
            arr[i].src, arr[i].dst, arr[i].row_bytes = a.data_ptr(), b.data_ptr(), rb
        N.check(lib.mom_select_apply(n, dst_index.data_ptr(), arr, len(part), s), "mom_select_apply")
    return outs


# --------------------------------------------------------------------------- densification statistics
def densify_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom, skip_flag=None, stream=None):
    """In place, for the Gaussians with radii > 0: running maximum radius, accumulated |dL/d mean2D| and its count
    (train_4DGS.py:266, scene/gaussian_model.py:713-715).  `skip_flag`: optional int32 device word; nonzero makes the
    launch a no-op (FusedAdam.skip_flag has the story)."""
    _need_cuda(radii, "densify_stats")
    P = radii.shape[0]
    for t_, n_ in ((max_radii2D, P), (xyz_gradient_accum, P), (denom, P)):
        if t_.numel() != n_ or not t_.is_contiguous() or t_.dtype != torch.float32:
            raise N.MomError("densify_stats: accumulators must be contiguous float32 tensors with one element per Gaussian")
    if radii.dtype != torch.int32 or not radii.is_contiguous():
        raise N.MomError("densify_stats: radii must be a contiguous int32 tensor")
    g = viewspace_grad
    if g.shape != (P, 3) or g.dtype != torch.float32 or not g.is_contiguous():
        raise N.MomError("densify_stats: viewspace gradient must be a contiguous float32 [P,3] tensor")
    N.check(N.lib().mom_densify_stats(P, radii.data_ptr(), g.data_ptr(), max_radii2D.data_ptr(), xyz_gradient_accum.data_ptr(),
                                      denom.data_ptr(), None if skip_flag is None else skip_flag.data_ptr(),
                                      N.current_stream() if stream is None else stream), "mom_densify_stats")


# --------------------------------------------------------------------------- SSIM
def ssim_window():
    """The 11 taps of the reference's gaussian(11, 1.5) (utils/loss_utils.py:29-31), computed the way it computes them
    (python doubles rounded to float32, then normalised in float32), as a ctypes array for the C ABI."""
    from math import exp
    g = torch.tensor([exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    g = g / g.sum()
    return (C.c_float * 11)(*g.tolist())


_SSIM_WINDOW = None


def _ssim_window():
    global _SSIM_WINDOW
    if _SSIM_WINDOW is None:
        _SSIM_WINDOW = ssim_window()
    return _SSIM_WINDOW


class SSIMFunction(torch.autograd.Function):
    """mean SSIM of two images (reference utils/loss_utils.py:52-92 with its default window), gradient w.r.t. img1."""

    @staticmethod
    def forward(ctx, img1, img2):
        _need_cuda(img1, "ssim")
        a = img1.detach().contiguous().float()
        b = img2.detach().contiguous().float()
        if a.shape != b.shape or a.dim() < 3:
            raise N.MomError(f"ssim: expected two [...,C,H,W] images of one shape, got {tuple(a.shape)} and {tuple(b.shape)}")
        H, W = a.shape[-2], a.shape[-1]
        Cn = a.numel() // max(1, H * W)
        need_grad = ctx.needs_input_grad[0]
        dm = torch.empty((3,) + tuple(a.shape), dtype=torch.float32, device=a.device) if need_grad else None
        total = torch.empty(N.SSIM_SUM_SLOTS, dtype=torch.float64, device=a.device)      # [0] = the sum, the rest scratch
        N.check(N.lib().mom_ssim_forward(Cn, H, W, _ssim_window(), a.data_ptr(), b.data_ptr(),
                                         None if dm is None else dm.data_ptr(), total.data_ptr(), N.current_stream()),
                "mom_ssim_forward")
        if need_grad:
            ctx.save_for_backward(a, b, dm)
        ctx.dims = (Cn, H, W)
        return (total[0] / max(1, a.numel())).float().reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b, dm = ctx.saved_tensors
        Cn, H, W = ctx.dims
        dimg = torch.zeros_like(a)
        g = g.detach().contiguous().float()
        N.check(N.lib().mom_ssim_backward(Cn, H, W, _ssim_window(), a.data_ptr(), b.data_ptr(), dm.data_ptr(),
                                          1.0 / max(1, a.numel()), g.data_ptr(), dimg.data_ptr(), N.current_stream()),
                "mom_ssim_backward")
        return dimg, None


def ssim(img1, img2):
    return SSIMFunction.apply(img1, img2)


# --------------------------------------------------------------------------- second stream of the autograd path
# OFF by default: measured in one process, alternating (tools/probe/api_leg.py, config 2): with it the path enqueues a step in 1.08
# instead of 0.95 ms of host time (a second Adam launch, a dozen ordering calls) and the HOST is what paces this path -- 925 against
# 1015 steps/s in async mode, 745 against 780-920 in exact mode.  It pays only where the GPU is the bottleneck (larger models, a
# faster host); MOM_API_OVERLAP=1 or ops.API_OVERLAP = True switches it on.
API_OVERLAP = _os.environ.get("MOM_API_OVERLAP", "0") == "1"
_side_streams = {}
_reg_pending = {}      # device -> True while the regulariser's gradient kernel may still run on the second stream (mark MARK_REG): the next writer of plane gradients waits
_params_ready = {}     # device -> (stream the last FusedAdam.step() ended on -- mark MARK_PARAMS --, {(id, version)} of every plane it updated)


def side_stream(device):
    """One second stream per device for the render() + loss.backward() path: it carries what does not depend on the compositing
    (the plane regularisers' two kernels), the MLP backward's partial-sum reduction and -- the largest piece -- the appearance
    parameters' Adam launch, which FusedAdam.step() starts there behind the event the backward recorded when those gradients
    became final (fused_autograd.py).  Opt-in (API_OVERLAP above): by default everything stays on the caller's stream."""
    st = _side_streams.get(device)
    if st is None:
        st = _side_streams[device] = torch.cuda.Stream(device=device)
    return st


# Ordering goes through libmom4d's stream helpers (csrc/stream_order.hip), not torch's Stream / Event objects: a dozen of these per
# iteration at 8-10 us each were a tenth of the path's host time.  Mark slots (per device): 0 = the parameters as the last
# FusedAdam.step() left them, 1 = the regulariser's gradient kernel on the second stream, 2 = the fused step's gradient bucket is cleared,
# 3 = a field's prefetched processing orders, 4.. = a ring of 60 for the backward's "appearance gradients final".
MARK_PARAMS, MARK_REG, MARK_BUCKET, MARK_ORDERS, MARK_RING0, MARK_RING_N = 0, 1, 2, 3, 4, 60
_ring = [0]


def stream_wait_stream(waiter, signaler):
    N.check(N.lib().mom_stream_wait_stream(waiter, signaler), "mom_stream_wait_stream")


def stream_mark(slot, stream):
    N.check(N.lib().mom_stream_mark(slot, stream), "mom_stream_mark")


def stream_wait_mark(stream, slot):
    N.check(N.lib().mom_stream_wait_mark(stream, slot), "mom_stream_wait_mark")


def zero_async(t, stream):
    """t.zero_() on a raw stream handle (no torch stream context: entering and leaving one costs ~10 us of host time)."""
    N.check(N.lib().mom_zero_async(t.data_ptr(), t.numel() * t.element_size(), stream), "mom_zero_async")


def next_ring_mark(stream):
    """Record the tail of `stream` under the next slot of the ring; returns the slot (valid until MARK_RING_N more have been taken)."""
    _ring[0] = (_ring[0] + 1) % MARK_RING_N
    slot = MARK_RING0 + _ring[0]
    stream_mark(slot, stream)
    return slot


def wait_reg_pending(device):
    """The current stream waits for a regulariser gradient kernel still running on the second stream (it ADDS into plane gradient
    buffers the caller is about to add into as well)."""
    if _reg_pending.pop(device, None) is not None:
        stream_wait_mark(N.current_stream(), MARK_REG)


def wait_reg_pending_all():
    """The same for every device that has such a kernel in flight: what FusedAdam.step() / zero_grad() and the per-op HexPlane
    backward call before they read or write plane gradients.  The autograd engine knows nothing of the raw second stream, so every
    consumer of the plane gradients that is not render()'s own backward joins here (the dict is empty unless API_OVERLAP is on)."""
    if _reg_pending:
        for dev in list(_reg_pending):
            with torch.cuda.device(dev):
                wait_reg_pending(dev)


# device -> number of fused render() nodes whose backward has not run yet.  The regulariser's gradient kernel may stay un-joined on
# the second stream only while such a node is still to come (its backward joins before it touches the plane gradients); with none
# pending -- the regulariser node runs last, the loss is regulariser-only, render() took the per-op path -- it joins at once.
_render_pending = {}


# --------------------------------------------------------------------------- plane regularisers
def direct_grads_ok(params, needs):
    """May a backward hand these parameters their gradients itself (p.grad = ...) instead of returning them to the engine?  Only
    if the engine would do exactly that with a returned gradient: every one of them is asked for (needs_input_grad), is a leaf
    that requires grad, and carries no tensor hook or post-accumulate hook -- those only fire on the engine's own path.
    (torch.autograd.grad() cannot be told apart from backward() inside a node: callers that use it switch the direct path off,
    fused_autograd.grads_through_graph().)"""
    for p, n in zip(params, needs):
        if not n or not p.requires_grad or not p.is_leaf:
            return False
        if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
            return False
    return True


def _buffers_free(views, held_by_cache=1):
    """True if nobody but the cache (held_by_cache references per view) and this call still refers to the cached gradient views: a
    caller that kept a parameter's .grad tensor from the last iteration must not see it zeroed and rewritten."""
    import sys
    return all(sys.getrefcount(v) <= held_by_cache + 2 for v in views)      # + the loop variable and getrefcount's argument


class PlaneRegFunction(torch.autograd.Function):
    """value = sum_p  w_smooth[p] * smooth2(plane_p) + w_l1[p] * mean|1 - plane_p|."""

    @staticmethod
    def forward(ctx, w_smooth, w_l1, *planes):
        _need_cuda(planes[0], "plane_regulation")
        # the descriptor array holds pointers, shapes and weights only: rebuilt when one of them moves (the loop hands in the same
        # twelve planes every iteration; twelve plane_storage() views and sixty field stores were 40 us of host time per call)
        key = (tuple((p.data_ptr(), p.shape, p.stride()) for p in planes), tuple(w_smooth), tuple(w_l1))
        c = PlaneRegFunction._fwd_cache
        if c is None or c[0] != key:
            arr = (N.MomRegPlane * len(planes))()
            for i, p in enumerate(planes):
                st = plane_storage(p)
                arr[i].plane, arr[i].grad = st.data_ptr(), None
                arr[i].H, arr[i].W = st.shape[0], st.shape[1]
                arr[i].w_smooth, arr[i].w_l1, arr[i].grad_scale = float(w_smooth[i]), float(w_l1[i]), 0.0
            c = PlaneRegFunction._fwd_cache = (key, arr)
        arr = c[1]
        dev = planes[0].device
        val = torch.empty(1, dtype=torch.float32, device=dev)
        if API_OVERLAP:
            # the value depends on the planes only: on the second stream it runs beside the compositing the caller's stream is
            # still busy with (the loop calls this right after render()); the caller's stream waits for it before it reads `val`.
            # The second stream need not wait for the whole of the caller's queue if the planes are exactly as the last
            # FusedAdam.step() left them (same tensors, same versions -- any torch write since would have bumped one): then the
            # event that step recorded is the only dependency
            cur, side = N.current_stream(), side_stream(dev).cuda_stream
            mark = _params_ready.get(dev)
            if mark is not None and mark[0] == cur and mark[1] >= frozenset((id(p), p._version) for p in planes):
                stream_wait_mark(side, MARK_PARAMS)
            else:
                stream_wait_stream(side, cur)
            N.check(N.lib().mom_plane_regulation(arr, len(planes), val.data_ptr(), side), "mom_plane_regulation")
            stream_wait_stream(cur, side)
        else:
            N.check(N.lib().mom_plane_regulation(arr, len(planes), val.data_ptr(), N.current_stream()), "mom_plane_regulation")
        ctx.save_for_backward(*planes)
        ctx.w = (list(w_smooth), list(w_l1))
        return val[0]

    @staticmethod
    def backward(ctx, g):
        planes = ctx.saved_tensors
        w_smooth, w_l1 = ctx.w
        # The kernel ADDS the regulariser's share into a plane's gradient.  Planes that already hold one of the right layout (the
        # render node ran first) are accumulated into in place; the others get a view of one flat zeroed buffer, which becomes
        # their gradient.  Handing the gradients to the planes here, instead of returning them, also spares the engine twelve
        # AccumulateGrad nodes and -- a plane has two consumers, this op and render() -- twelve elementwise additions
        # (DIRECT_GRADS = False returns them through the graph).  The buffer and the descriptor are cached between iterations
        # while the planes have let go of last iteration's gradients (zero_grad(set_to_none=True)); the upstream weight stays on
        # the device (float(g) was a host synchronisation in every iteration).
        direct = PlaneRegFunction.DIRECT_GRADS and direct_grads_ok(planes, ctx.needs_input_grad[2:])
        held = [p.grad if direct else None for p in planes]
        in_place = in_place_flags(held, planes)
        key = (tuple(p.data_ptr() for p in planes), tuple(w_smooth), tuple(w_l1),
               tuple(h.data_ptr() if ip else 0 for h, ip in zip(held, in_place)))
        c = PlaneRegFunction._cache
        busy = c is not None and (any(h is not None and h.data_ptr() == v.data_ptr() for h, v in zip(held, c[2]))
                                  or not _buffers_free(c[2]))
        if c is None or c[0] != key or busy:
            flat = torch.zeros(sum(p.numel() for p in planes), dtype=torch.float32, device=planes[0].device)
            off, grads = 0, []
            arr = (N.MomRegPlane * len(planes))()
            for i, p in enumerate(planes):
                st = plane_storage(p)
                gv = flat[off:off + p.numel()].view(st.shape)
                off += p.numel()
                grads.append(gv.permute(2, 0, 1).unsqueeze(0))
                target = plane_storage(held[i]) if in_place[i] else gv
                arr[i].plane, arr[i].grad = st.data_ptr(), target.data_ptr()
                arr[i].H, arr[i].W = st.shape[0], st.shape[1]
                arr[i].w_smooth, arr[i].w_l1, arr[i].grad_scale = float(w_smooth[i]), float(w_l1[i]), 1.0
            val = torch.empty(1, dtype=torch.float32, device=planes[0].device)
            c = (key, flat, grads, arr, val)
            if not busy:
                PlaneRegFunction._cache = c
        else:
            c[1].zero_()
        _, flat, grads, arr, val = c
        up = g.detach().reshape(1).float()
        dev = planes[0].device
        if API_OVERLAP and direct:
            # the engine runs this node BEFORE render()'s (it was created later): on the second stream the kernel runs beside the
            # compositing backward; whoever adds into the plane gradients next (the HexPlane backward, a second regulariser call)
            # waits for the event first (wait_reg_pending)
            wait_reg_pending(dev)
            side_t = side_stream(dev)
            cur, side = N.current_stream(), side_t.cuda_stream
            stream_wait_stream(side, cur)              # the upstream weight, and the clearing of the buffer above
            N.check(N.lib().mom_plane_regulation_grad(arr, len(planes), val.data_ptr(), up.data_ptr(), side), "mom_plane_regulation")
            up.record_stream(side_t)                   # (the engine frees it when this node returns; the second stream still reads it)
            if _render_pending.get(dev, 0) > 0:
                stream_mark(MARK_REG, side)
                _reg_pending[dev] = True
            else:
                stream_wait_stream(cur, side)          # nobody downstream is known to join: the caller's stream does, now
        else:
            N.check(N.lib().mom_plane_regulation_grad(arr, len(planes), val.data_ptr(), up.data_ptr(), N.current_stream()),
                    "mom_plane_regulation")
        if direct:
            for p, gp, ip in zip(planes, grads, in_place):
                if not ip:
                    if p.grad is None:
                        p.grad = gp
                    else:
                        wait_reg_pending(dev)          # (a torch op on the caller's stream reads what the second stream writes)
                        p.grad.add_(gp)
            return (None, None) + (None,) * len(planes)
        return (None, None, *[gp if n else None for gp, n in zip(grads, ctx.needs_input_grad[2:])])

    _cache = None
    _fwd_cache = None
    DIRECT_GRADS = True


def plane_regulation(planes, w_smooth, w_l1):
    return PlaneRegFunction.apply(w_smooth, w_l1, *planes)


# --------------------------------------------------------------------------- Adam
# numpy mirror of N.MomAdamTensor (same offsets and size): FusedAdam writes whole columns of its descriptor arrays through it
_ADAM_DT = np.dtype({"names": [n for n, _ in N.MomAdamTensor._fields_],
                     "formats": [{C.c_void_p: np.uint64, C.c_size_t: np.uint64, C.c_float: np.float32}[t] for _, t in N.MomAdamTensor._fields_],
                     "offsets": [getattr(N.MomAdamTensor, n).offset for n, _ in N.MomAdamTensor._fields_],
                     "itemsize": C.sizeof(N.MomAdamTensor)})


def _dense(t):
    """True if t's elements tile its storage span without gaps or overlap (any dim order)."""
    dims = sorted(((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1), reverse=True)
    expect = 1
    for st, sz in reversed(dims):
        if st != expect:
            return False
        expect *= sz
    return True


def _same_layout(a, b):
    """Same shape and same strides on every dimension that has more than one element."""
    if a.shape != b.shape:
        return False
    if a.stride() == b.stride():          # the common case, one tuple comparison (this runs 24 times per iteration of the API path)
        return True
    return all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


_IN_PLACE_MEMO = {}


def in_place_flags(held, planes):
    """[h is a gradient tensor the kernels can add into in place: same layout as its plane, dense] for every plane.  The answer
    depends on the two tensors' storage, shape and strides only, and the loop hands in the SAME buffers iteration after iteration
    (the gradient buffers are cached), so it is remembered per (gradient pointer, plane pointer): twelve pairs of pointer reads
    instead of twelve shape-and-stride walks, twice per iteration of the API path."""
    out = []
    for h, p in zip(held, planes):
        if h is None:
            out.append(False)
            continue
        k = (h.data_ptr(), p.data_ptr(), h.shape, h.stride())
        v = _IN_PLACE_MEMO.get(k)
        if v is None:
            if len(_IN_PLACE_MEMO) > 4096:
                _IN_PLACE_MEMO.clear()
            v = _IN_PLACE_MEMO[k] = bool(_same_layout(h, p) and _dense(h))
        out.append(v)
    return out


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False) whose step() is ONE HIP launch over all
    parameters with a gradient.  State layout ('step', 'exp_avg', 'exp_avg_sq') and param_groups are those of
    torch.optim.Adam, so the reference's optimizer-state surgery (scene/gaussian_model.py:409-482) and
    state_dict()/load_state_dict() work unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # the other keys are torch.optim.Adam's (all at their defaults: this optimizer implements exactly that configuration), so
        # that a state_dict() written here loads into the reference's torch.optim.Adam and the other way round
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))
        self._plan = None
        self._plans = {}            # "early" / "late": the plans of a step taken in two parts (step_partial)
        self._early = None          # ids of the parameters step_partial() has already advanced in this iteration
        # int32 device word or None.  While it is nonzero on the device a step() changes neither parameters nor moments:
        # the asynchronous training step points it at the rasterizer's sticky overflow word, so that a step whose image was
        # truncated never reaches the model; the host notices later and replays (train.Trainer._recover, rewind()).
        self.skip_flag = None
        # (mark slot, second stream's handle, [(param, its gradient tensor, the gradient's version)]) left by render()'s backward when the
        # appearance parameters' gradients became final, or None: see step()
        self.early_hint = None

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad without its dynamo guard, foreach grouping and profiler range (30 us per call on the
        API path's host, which sets the pace there): the same effect for dense gradients."""
        wait_reg_pending_all()         # a regulariser gradient kernel on the second stream may still be adding into what is dropped here
        for group in self.param_groups:
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    if g.grad_fn is not None:
                        g.detach_()
                    else:
                        g.requires_grad_(False)
                    g.zero_()

    def rewind(self, n):
        """Take back the host-side step counters of the last n step() calls (they were no-ops on the device)."""
        for st in self.state.values():
            if "step" in st:
                st["step"] -= float(n)

    def _build_plan(self, live, key, ranges=None):
        """Validate every (param, state) once and lay the MomAdamTensor arrays out; reused until a parameter or a moment moves
        (densify / prune / reset_opacity replace parameters and state, load_state_dict replaces state).  The gradient pointers
        are NOT part of the plan's identity: under autograd the gradients are fresh tensors every iteration, and a plan keyed on
        them was rebuilt in four iterations out of ten (0.5 ms of host time each); they are refreshed per step instead."""
        by_cfg, entries, steps, params = {}, [], [], []
        for group, p in live:
            _need_cuda(p, "FusedAdam")
            st = self.state[p]
            if not _dense(p):
                raise N.MomError("FusedAdam: parameters must be dense")
            b1, b2 = group["betas"]
            cfg = (b1, b2, group["eps"])
            slot = by_cfg.setdefault(cfg, [])
            t = N.MomAdamTensor()
            # ranges (a rank's share of a sharded step, FusedAdam.step_partial): elements [off, off + n) of the tensor in storage
            # order -- Adam is element-wise, so a slice of the parameter, its gradient and its moments is a step of its own; an empty
            # slice stays in the plan with n = 0 (no work) so that the parameter's step counter advances on every rank alike
            off, cnt = (0, p.numel()) if ranges is None else ranges[id(p)]
            if ranges is not None and (not p.is_contiguous() or not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous()):
                raise N.MomError("FusedAdam: a sharded step needs contiguous parameters and moments")
            t.param = p.data_ptr() + 4 * off
            t.exp_avg, t.exp_avg_sq = st["exp_avg"].data_ptr() + 4 * off, st["exp_avg_sq"].data_ptr() + 4 * off
            t.n = cnt
            entries.append((group, b1, b2, cfg, len(slot)))
            slot.append(t)
            steps.append(st["step"])
            params.append(p)
        arrs = {cfg: (N.MomAdamTensor * len(ts))(*ts) for cfg, ts in by_cfg.items()}
        # numpy views of the descriptor arrays (same memory): a step refreshes three float columns and one pointer column of ~32 rows
        # with four vector assignments instead of ~130 ctypes field stores (0.3 us each: a tenth of the render() path's host time)
        views = {cfg: np.frombuffer(arr, dtype=_ADAM_DT) for cfg, arr in arrs.items()}
        cfg_list = list(arrs)
        rows = {cfg: [] for cfg in cfg_list}            # per configuration: the plan-entry index of every row of its array
        for j, (_, _, _, cfg, i) in enumerate(entries):
            rows[cfg].append(j)
        groups = []
        gidx = []
        for group, _ in live:
            for k, gq in enumerate(groups):
                if gq is group:
                    gidx.append(k)
                    break
            else:
                groups.append(group)
                gidx.append(len(groups) - 1)
        # The step counters of the plan's parameters live in ONE host buffer and every state["step"] is a 0-d view into it (same
        # dtype, same values, still torch.optim.Adam's state layout): advancing and reading thirty-odd separate host tensors cost
        # 65 us per step() (torch._foreach_add_ + torch.stack), a tenth of the API path's host time; one add and one tolist() on the
        # buffer cost 3.  A counter that is replaced (load_state_dict) changes its pointer and with it the plan's key.
        buf = torch.tensor([float(t) for t in steps], dtype=torch.float32)
        for j, (_, p) in enumerate(live):
            self.state[p]["step"] = buf[j]
        return {"key": self._plan_key(live) + (() if ranges is None else (tuple(ranges[id(p)] for _, p in live),)), "entries": entries,
                "step_buf": buf, "arrs": arrs, "params": params, "offs": [0 if ranges is None else ranges[id(p)][0] for _, p in live],
                "views": views, "rows": {cfg: np.asarray(r, dtype=np.int64) for cfg, r in rows.items()}, "groups": groups,
                "gidx": np.asarray(gidx, dtype=np.int64), "offs4": np.asarray([0 if ranges is None else 4 * ranges[id(p)][0] for _, p in live], dtype=np.uint64),
                "last_grad": [0] * len(params), "betas": [(b1, b2) for _, b1, b2, _, _ in entries]}

    def _plan_key(self, live):
        return tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(),
                      self.state[p]["step"].data_ptr(), p.numel()) for _, p in live)

    def _launch(self, live, which, stream=None, ranges=None):
        state, key = self.state, []
        for _, p in live:               # one pass: create missing state, and read the pointers the plan is keyed on
            st = state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            key.append((p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel()))
        key = tuple(key)
        if ranges is not None:
            key = key + (tuple(ranges[id(p)] for _, p in live),)
        plan = self._plans.get(which)
        if plan is None or plan["key"] != key:
            plan = self._plans[which] = self._build_plan(live, key, ranges)
        if which is None:
            self._plan = plan
        # this step's gradients: pointers refreshed and strides checked every step.  (Shape, device and dtype need no check here:
        # torch refuses a .grad assignment whose size, device or dtype differs from the parameter's -- THPVariable_set_grad --
        # but it accepts any strides.)
        arrs = plan["arrs"]
        last, ptrs = plan["last_grad"], []
        for j, p in enumerate(plan["params"]):
            g = p.grad
            gp = g.data_ptr()
            if gp != last[j]:           # (the same buffer as last step -- the gradient buffers are cached -- was checked then)
                if g.stride() != p.stride() and not _same_layout(g, p):
                    raise N.MomError("FusedAdam: param/grad must be dense with identical strides")
                last[j] = gp
            ptrs.append(gp)
        plan["step_buf"] += 1
        # the step counters are host tensors the state surgery and rewind() may have touched: read them all in one go, and form
        # the two bias corrections once per distinct (betas, step) -- one or two values, not one per tensor.  Python float
        # arithmetic, exactly as torch.optim.Adam forms them (numpy's pow may round differently in the last place)
        step_vals = plan["step_buf"].tolist()
        bias = {}
        bc1, bc2 = [], []
        for (b1, b2), step in zip(plan["betas"], step_vals):
            bc = bias.get((b1, b2, step))
            if bc is None:
                bc = bias[(b1, b2, step)] = (1.0 - b1 ** step, math.sqrt(1.0 - b2 ** step))
            bc1.append(bc[0])
            bc2.append(bc[1])
        grad_col = np.asarray(ptrs, dtype=np.uint64) + plan["offs4"]
        lr_col = np.asarray([float(gq["lr"]) for gq in plan["groups"]], dtype=np.float64)[plan["gidx"]]
        bc1_col, bc2_col = np.asarray(bc1, dtype=np.float64), np.asarray(bc2, dtype=np.float64)
        for cfg, view in plan["views"].items():
            r = plan["rows"][cfg]
            view["grad"] = grad_col[r]
            view["lr"] = lr_col[r]                      # (float64 -> float32 on assignment: the rounding a ctypes c_float store makes)
            view["bias_correction1"] = bc1_col[r]
            view["bias_correction2_sqrt"] = bc2_col[r]
        for (b1, b2, eps), arr in arrs.items():
            N.check(N.lib().mom_adam_step(arr, len(arr), b1, b2, eps,
                                          None if self.skip_flag is None else self.skip_flag.data_ptr(),
                                          N.current_stream() if stream is None else stream), "mom_adam_step")

    @torch.no_grad()
    def ensure_state(self, params):
        """Create the Adam state of `params` now, on the current stream (torch.optim.Adam creates it in the first step() that sees
        a gradient; a step_partial() on a second stream would otherwise allocate the moments from that stream's pool).
        Returns True if any state was created: its zero fill is queued on the current stream, and a second stream that is to
        read the moments must first wait for THAT stream, not only for an earlier mark."""
        created = False
        for p in params:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                created = True
        return created

    @torch.no_grad()
    def step_partial(self, params, stream=None, ranges=None):
        """Advance only `params` (their gradients are final) on the CURRENT stream; the step() that follows in the same iteration
        advances the rest.  The fused training step uses it to put the Gaussians' appearance parameters -- 56 of their 59 floats,
        four fifths of Adam's bytes -- on its second stream underneath the deformation backward (fused_step.py).  Element for
        element the same update as one step(): Adam is element-wise.
        ranges: {id(p): (first element, count)} -- only that slice of each parameter (storage order) is advanced: this rank's share
        of a camera-batch shard whose ranks reduce-scatter the gradients and all-gather the updated parameters (parallel.py)."""
        ids = {id(p) for p in params}
        live = [(group, p) for group in self.param_groups for p in group["params"] if id(p) in ids and p.grad is not None]
        if not live:
            return
        self._launch(live, "early" if ranges is None else "early-sharded", stream=stream, ranges=ranges)
        self._early = {id(p) for _, p in live}

    @torch.no_grad()
    def step(self, closure=None):
        # (torch.optim.Optimizer wraps every subclass's step() in profile_hook_step -- a record_function range, two hook walks and
        # a functools wrapper: 40 us per call, 4 % of the render() path's host time.  The class is marked `hooked` below so that the
        # wrapper is not installed; registered step hooks are still honoured, here.)
        pre = self._optimizer_step_pre_hooks
        post = self._optimizer_step_post_hooks
        if pre or post or _TORCH_OPT._global_optimizer_pre_hooks or _TORCH_OPT._global_optimizer_post_hooks:
            return self._step_with_hooks(closure)
        return self._step(closure)

    def _step_with_hooks(self, closure):
        import itertools
        O = _TORCH_OPT
        for hook in itertools.chain(O._global_optimizer_pre_hooks.values(), self._optimizer_step_pre_hooks.values()):
            r = hook(self, (closure,), {})
            if r is not None:
                (closure,), _ = r if isinstance(r, tuple) and len(r) == 2 else ((closure,), {})
        out = self._step(closure)
        for hook in itertools.chain(self._optimizer_step_post_hooks.values(), O._global_optimizer_post_hooks.values()):
            hook(self, (closure,), {})
        return out

    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        hint, self.early_hint = self.early_hint, None
        wait_reg_pending_all()         # the plane gradients this step reads are complete (a regulariser kernel on the second stream)
        joined = None
        if hint is not None and self._early is None:
            # render()'s backward (fused_autograd.py) recorded an event when the appearance parameters' gradients -- SH, scaling,
            # rotation, opacity: four fifths of Adam's bytes -- were final, half-way through the backward.  If nothing has touched
            # them since (same gradient tensors, same versions: any torch in-place op bumps the version; densify / prune replace the
            # parameters and with them the optimizer's entries), their update goes to the second stream behind that event and runs
            # underneath the deformation backward the GPU is still working on, as in the fused step (fused_step.py: early_adam).
            # Element for element the same update as one step().
            slot, side, entries = hint
            mine = {id(p) for group in self.param_groups for p in group["params"]}
            if all(id(p) in mine and p.grad is g and g._version == v for p, g, v in entries):
                ps = [p for p, _, _ in entries]
                if self.ensure_state(ps):
                    # first step (or state made lazily after densify): the moments' zero fill sits on the caller's stream BEHIND the
                    # whole backward, the mark half-way through it -- the second stream waits for the fill itself this once
                    stream_wait_stream(side, N.current_stream())
                stream_wait_mark(side, slot)
                self.step_partial(ps, stream=side)
                joined = side
        early, self._early = self._early, None
        live = [(group, p) for group in self.param_groups for p in group["params"]
                if p.grad is not None and (early is None or id(p) not in early)]
        if live:
            self._launch(live, None if early is None else "late")
        if joined is not None:
            stream_wait_stream(N.current_stream(), joined)      # the step is complete for whatever the caller enqueues next
        if API_OVERLAP and live:
            # what a consumer of the parameters alone (the plane regularisers' forward) may wait for instead of the whole stream
            planes = [p for _, p in live if p.dim() == 4]
            if planes and planes[0].is_cuda:
                cur = N.current_stream()
                stream_mark(MARK_PARAMS, cur)
                _params_ready[planes[0].device] = (cur, frozenset((id(p), p._version) for p in planes))
        return loss


_TORCH_OPT = __import__("sys").modules["torch.optim.optimizer"]     # (torch.optim deletes the submodule's name from its namespace)
FusedAdam.step.hooked = True      # see FusedAdam.step: keeps torch.optim.Optimizer._patch_step_function from wrapping it


# --------------------------------------------------------------------------- backend switch
class _HipBackend:
    """The product path.  Tests of pure host logic on a GPU-less machine may install the oracle's torch
    restatement here EXPLICITLY (oracle.torch_ref.TorchBackend); nothing falls back to it by itself."""
    name = "hip"
    hexplane_features = staticmethod(hexplane_features)
    hexplane_orders = staticmethod(hexplane_orders)
    morton_order = staticmethod(morton_order)
    deform_mlp = staticmethod(deform_mlp)
    l1_loss_with_sums = staticmethod(l1_loss_with_sums)
    ssim = staticmethod(ssim)
    densify_stats = staticmethod(densify_stats)
    select_rows = staticmethod(select_rows)
    plane_regulation = staticmethod(plane_regulation)
    Adam = FusedAdam


BACKEND = _HipBackend


def set_backend(b):
    global BACKEND
    BACKEND = b
