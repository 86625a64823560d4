"""Stand-in for the reference's pybind module `diff_gaussian_rasterization._C`
(submodules/depth-diff-gaussian-rasterization/ext.cpp:14-18, rasterize_points.h:18-68):
same three functions, same argument order, same return tuples -- backed by libmom4d.so.

Differences a caller can observe, all documented in INTEGRATION.md:
* the byte buffers' internal layout is private to libmom4d;
* `num_rendered` (the first return value of rasterize_gaussians, rasterize_points.cu:35-117): in the default "exact" mode
  -- one host sync per forward, like the reference's cudaMemcpy at rasterizer_impl.cu:282 -- it is the TRUE number of
  instances this forward binned: the reference's count bit for bit under set_keep_all_tiles(True), and that count less the
  (splat, tile) pairs that cannot reach alpha >= 1/255 under the default tile cull (images and gradients are identical
  either way).  Only in the opt-in "async" mode (set_sync_mode) is it the CAPACITY the binning buffer was sized for, an
  upper bound of the count; the backward takes whichever of the two the forward returned.  This is the one place where the
  boundary's integer contract deviates, and only on request.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _native as N

from collections import deque

_state = {"mode": "exact", "cap_hint": 0, "last_R": None, "flag": None, "pending": deque(), "keep_all_tiles": False,
          "serial": 0, "verified": 0}
_FLAG_LAG = 4


class BinningOverflow(RuntimeError):
    """An async-mode forward did not fit its binning buffer.  `serial`: the number of that forward (async forwards are counted from
    1 in _state["serial"]); every forward before it is known to be complete, it and every later one were truncated and -- where
    FusedAdam.skip_flag / the statistics kernel were given overflow_flag() -- left the model untouched."""

    def __init__(self, msg, serial):
        super().__init__(msg)
        self.serial = serial


_PIN_RING = 256


def pinned_word():
    """A pinned int32 word from a ring (page-locked allocations are slow and can stall the device: none per call).  A word is
    handed out again 256 calls later -- far beyond the few calls for which the rasterizer state refers to it."""
    ring = _state.get("pin_ring")
    if ring is None:
        ring = _state["pin_ring"] = torch.zeros(_PIN_RING, dtype=torch.int32).pin_memory()
        _state["pin_next"] = 0
    i = _state["pin_next"]
    _state["pin_next"] = (i + 1) % _PIN_RING
    w = ring[i:i + 1]
    w[0] = COUNT_PENDING       # (a host store: the geometry stage overwrites it with the frame's count, which wait_count() polls for)
    return w


COUNT_PENDING = -1             # no frame has 2^32 - 1 instances


def post_slot(serial):
    """(address, slot) of the pinned 64-bit word async-mode forward number `serial` posts its status bits into
    (MomRasterArgs.status_post: the compositing kernel stores (serial << 32) | bits there itself -- no copy command and no event behind
    the forward; they were 15 us of host time per call on a path the host paces).  A slot comes round again 256 forwards later."""
    ring = _state.get("post_ring")
    if ring is None:
        ring = _state["post_ring"] = torch.zeros(_PIN_RING, dtype=torch.int64).pin_memory()
        _state["post_np"] = ring.numpy()
    slot = serial % _PIN_RING
    return ring.data_ptr() + 8 * slot, slot


def posted(slot, serial, timeout_s=60.0):
    """The status bits forward `serial` posted into `slot`; polled until they are there (wait_count explains why not an event)."""
    import time
    ring, want = _state["post_np"], serial & 0xFFFFFFFF
    v = int(ring[slot])
    if (v >> 32) & 0xFFFFFFFF != want:
        t0 = time.perf_counter()
        while True:
            v = int(ring[slot])
            if (v >> 32) & 0xFFFFFFFF == want:
                break
            if time.perf_counter() - t0 > timeout_s:
                raise N.MomError(f"async forward {serial} never posted its status (slot {slot} holds {v:#x})")
    return v & 0xFFFFFFFF


def wait_count(nr_host, ev=None, spin_s=float(__import__("os").environ.get("MOM_COUNT_SPIN_S", "0.02"))):
    """The frame's instance count, as soon as the geometry stage has written it.  The count reaches the host without a copy command:
    tile_scan stores it into this pinned, device-visible word (system scope), so the host can POLL the word instead of waiting for
    an event behind the stage -- hipEventSynchronize parks the thread and is woken by the runtime's signal handler up to a
    millisecond late (tools/probe/window20.py: a window ended by an event wait read 1100-1150 steps/s, the same window ended by a
    spinning device synchronisation 1181, twice), and exact mode pays that wait in EVERY iteration.  After `spin_s` seconds of
    polling the event (or the stream) is waited for the ordinary way."""
    import time
    v = int(nr_host[0])
    if v != COUNT_PENDING:
        return v
    t_end = time.perf_counter() + spin_s
    while time.perf_counter() < t_end:
        v = int(nr_host[0])
        if v != COUNT_PENDING:
            return v
    if ev is not None:
        ev.synchronize()
    else:
        torch.cuda.current_stream().synchronize()
    v = int(nr_host[0])
    if v == COUNT_PENDING:
        # the stream has drained and nobody wrote the word: the geometry stage never ran (a launch failed after its arguments were
        # accepted).  Sizing the binning buffer from the sentinel (-1) would follow; fail here instead.
        raise N.MomError("the geometry stage finished without writing this frame's instance count (the word still holds "
                         "COUNT_PENDING): a launch of the stage failed")
    return v


def overflow_flag(device):
    """The sticky int32 device word async-mode forwards OR their overflow bit into (created on first use).  Point
    ops.FusedAdam.skip_flag at it and no optimizer step taken after an overflow reaches the model."""
    f = _state["flag"]
    if f is None or f.device != torch.device(device):
        f = _state["flag"] = torch.zeros(1, dtype=torch.int32, device=device)
        _state["pending"].clear()
    return f


def _check_overflow(lag):
    q = _state["pending"]
    while len(q) > lag:
        slot, count, serial = q.popleft()
        if not (posted(slot, serial) & 1):
            _state["verified"] = serial
            continue
        q.clear()
        torch.cuda.current_stream().synchronize()      # nothing that still tests the word is in flight when it is cleared
        _state["flag"].zero_()
        _state["cap_hint"] = max(_state["cap_hint"], int(count[0]) * 2)
        raise BinningOverflow("libmom4d: an async-mode forward overflowed its binning capacity (instance count "
                              f"{int(count[0])}); its image and every image since were truncated.  Optimizer steps were skipped "
                              "on the device from that forward on if FusedAdam.skip_flag is overflow_flag(); the capacity "
                              "hint has been doubled -- repeat those iterations, or use set_sync_mode('exact')", serial)


def set_sync_mode(mode: str, capacity_hint: int = 0) -> None:
    """'exact': read the instance count back (blocking) and size the binning buffer exactly.
    'async': never block; size the buffer from `capacity_hint` / 1.5x the last observed count.  An overflow sets a sticky
    device word (overflow_flag) that is read back behind an event and raised a few calls later (_FLAG_LAG)."""
    assert mode in ("exact", "async")
    _state["mode"] = mode
    if capacity_hint:
        _state["cap_hint"] = int(capacity_hint)


def set_keep_all_tiles(on: bool) -> None:
    """True: bin every tile of a splat's rectangle, as the reference does -- num_rendered and the tile lists are then the
    reference's bit for bit (the parity tests of the sort use this).  False (default): instances that cannot reach
    alpha >= 1/255 anywhere in their tile are not binned; images and gradients are bit-identical either way
    (include/mom4d.h, MomRasterArgs.keep_all_tiles).  Forward and backward of one frame must run under the same setting."""
    _state["keep_all_tiles"] = bool(on)


def last_num_rendered():
    """Instance count of the most recent forward (async mode: synchronises nothing, may lag by one call)."""
    r = _state["last_R"]
    if r is None:
        return None
    v = int(r[0])
    return None if v == COUNT_PENDING else v        # (async mode: the geometry stage of that forward has not written it yet)


def _args(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
          tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug):
    a = N.MomRasterArgs()
    a.P = means3D.shape[0]
    a.D = int(degree)
    a.M = int(sh.shape[1]) if sh.numel() != 0 else 0
    a.W, a.H = int(W), int(H)
    keep = []

    def dev(t):
        if t is None or t.numel() == 0:
            return None
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.contiguous().float()
        keep.append(t)
        return t.data_ptr()

    a.background = dev(bg)
    a.means3D = dev(means3D)
    a.shs = dev(sh)
    a.colors_precomp = dev(colors)
    a.opacities = dev(opacity)
    a.scales = dev(scales)
    a.rotations = dev(rotations)
    a.cov3D_precomp = dev(cov3D_precomp)
    a.viewmatrix = dev(viewmatrix)
    a.projmatrix = dev(projmatrix)
    a.campos = dev(campos)
    a.scale_modifier = float(scale_modifier)
    a.tan_fovx, a.tan_fovy = float(tan_fovx), float(tan_fovy)
    a.prefiltered, a.debug = int(bool(prefiltered)), int(bool(debug))
    a.keep_all_tiles = int(_state["keep_all_tiles"])
    return a, keep


def exact_render(lib, a, geom, img, out_color, out_depth, nr_host, P, W, H, dev, stream, regeometry):
    """Second half of an exact-mode forward, behind mom_raster_forward_geometry on `stream`: returns (the TRUE instance count,
    the binning buffer).  The reference waits for the count here and then sizes the buffer (rasterizer_impl.cu:282-285); waiting
    first leaves the GPU idle for as long as the host needs to come back and enqueue the compositing.  So the compositing is
    enqueued FIRST, into a buffer sized from the counts of earlier exact-mode forwards, and the host then waits only for the
    geometry stage (an event between the two launches): the count is there, the GPU is already sorting and compositing.  If the
    count exceeds the guess -- the first frame, a sudden jump -- the stream is drained and the frame's rasterizer stages run again
    (regeometry(): the geometry stage, whose bucket cursors the truncated scatter has consumed; then the compositing with the exact
    size), before anything else has been enqueued: what the caller gets back is complete either way, and num_rendered is the
    frame's own count.  The buffer is merely larger than the reference's would be."""
    guess = _state.get("exact_cap", 0)
    if guess:
        binning = torch.empty((lib.mom_raster_binning_bytes(P, W, H, guess),), dtype=torch.uint8, device=dev)
        N.check(lib.mom_raster_forward_render(C.byref(a), geom.data_ptr(), binning.data_ptr(), guess, img.data_ptr(),
                                              out_color.data_ptr(), out_depth.data_ptr(), None, stream), "mom_raster_forward_render")
    # (no event between the two launches any more: the count is polled, and the rare fallback -- the poll timed out -- waits for the
    # whole stream, i.e. for the compositing too; creating and recording a torch event cost every iteration 20 us of host time)
    count = wait_count(nr_host, None)
    if count > guess or not guess:
        # (not guess: nothing was enqueued above -- also the case of a frame with NO instances before any guess exists; the
        # compositing still has to run, it writes the background image, as the reference does for num_rendered == 0)
        if guess:
            torch.cuda.current_stream().synchronize()      # the truncated pass is out of the way before its buffers are reused
            regeometry()
        guess = count
        binning = torch.empty((lib.mom_raster_binning_bytes(P, W, H, guess),), dtype=torch.uint8, device=dev)
        N.check(lib.mom_raster_forward_render(C.byref(a), geom.data_ptr(), binning.data_ptr(), guess, img.data_ptr(),
                                              out_color.data_ptr(), out_depth.data_ptr(), None, stream), "mom_raster_forward_render")
    # next frame's guess: a quarter above the largest count seen, in steps (so that the allocator sees the same size again)
    want = ((int(count * 1.25) + 65535) // 65536) * 65536
    if want > _state.get("exact_cap", 0) or _state.get("exact_cap", 0) > 4 * want:
        _state["exact_cap"] = want
    _state["exact_last_capacity"] = guess
    return count, binning


def rasterize_gaussians(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
                        debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-117)."""
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59
    if not means3D.is_cuda:
        raise RuntimeError("libmom4d has no CPU path: means3D must live on the GPU")
    lib = N.lib()
    dev = means3D.device
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    geom = torch.empty((lib.mom_raster_geom_bytes(P),), dtype=torch.uint8, device=dev)
    img = torch.empty((lib.mom_raster_image_bytes(W, H),), dtype=torch.uint8, device=dev)
    if P == 0:  # rasterize_points.cu:82
        out_color.zero_()
        out_depth.zero_()
        return 0, out_color, out_depth, radii, geom, torch.empty((0,), dtype=torch.uint8, device=dev), img
    a, keep = _args(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                    projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug)
    stream = N.current_stream()
    nr_dev = torch.empty((1,), dtype=torch.int32, device=dev)
    nr_host = pinned_word()
    N.check(lib.mom_raster_forward_geometry(C.byref(a), geom.data_ptr(), img.data_ptr(), radii.data_ptr(),
                                            nr_dev.data_ptr(), nr_host.data_ptr(), stream), "mom_raster_forward_geometry")
    # async with nothing to size from (no hint, no earlier frame): this one forward waits for its count, like exact mode
    blind = _state["mode"] == "async" and _state["cap_hint"] == 0 and _state["last_R"] is None
    if _state["mode"] == "exact" or blind:
        def regeometry():
            N.check(lib.mom_raster_forward_geometry(C.byref(a), geom.data_ptr(), img.data_ptr(), radii.data_ptr(), nr_dev.data_ptr(),
                                                    nr_host.data_ptr(), stream), "mom_raster_forward_geometry")
        cap, binning = exact_render(lib, a, geom, img, out_color, out_depth, nr_host, P, W, H, dev, stream, regeometry)
        if blind:
            _state["cap_hint"] = int(cap * 1.5) + 4096
        _state["last_R"] = nr_host
        del keep
        return cap, out_color, out_depth, radii, geom, binning, img
    else:
        flag = overflow_flag(dev)
        _check_overflow(_FLAG_LAG)
        prev = _state["last_R"]
        if prev is not None and int(prev[0]) != COUNT_PENDING:      # an earlier call's count (a sizing hint; whichever copy has landed)
            _state["cap_hint"] = max(_state["cap_hint"], int(int(prev[0]) * 1.5) + 4096)
        cap = max(_state["cap_hint"], 4096)
    _state["last_R"] = nr_host
    binning = torch.empty((lib.mom_raster_binning_bytes(P, W, H, cap),), dtype=torch.uint8, device=dev)
    tracked = _state["mode"] == "async" and not blind
    if tracked:
        # the frame's own status bits come back through a pinned word the compositing kernel writes (post_slot); the sticky device
        # word (flag) is what gates the optimizer on the device
        _state["serial"] += 1
        a.status_post, slot = post_slot(_state["serial"])
        a.status_serial = _state["serial"] & 0xFFFFFFFF
    N.check(lib.mom_raster_forward_render(C.byref(a), geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(),
                                          out_color.data_ptr(), out_depth.data_ptr(), flag.data_ptr() if tracked else None, stream),
            "mom_raster_forward_render")
    if tracked:
        _state["pending"].append((slot, nr_host, _state["serial"]))
    del keep
    return cap, out_color, out_depth, radii, geom, binning, img


def rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                 projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:119-202)."""
    lib = N.lib()
    dev = means3D.device
    P = means3D.shape[0]
    H, W = dL_dout_color.shape[1], dL_dout_color.shape[2]
    M = int(sh.shape[1]) if sh.numel() != 0 else 0
    opt = dict(dtype=torch.float32, device=dev)
    g2d = torch.empty((P, 3), **opt)
    gcol = torch.empty((P, 3), **opt)
    gop = torch.empty((P, 1), **opt)
    g3d = torch.empty((P, 3), **opt)
    gcov = torch.empty((P, 6), **opt)
    have_sr = scales.numel() != 0
    gsh = torch.empty((P, M, 3), **opt) if (M and colors.numel() == 0) else torch.zeros((P, M, 3), **opt)
    gsc = torch.empty((P, 3), **opt) if have_sr else torch.zeros((P, 3), **opt)
    grot = torch.empty((P, 4), **opt) if have_sr else torch.zeros((P, 4), **opt)
    if P == 0:
        return g2d, gcol, gop, g3d, gcov, gsh, gsc, grot
    # opacity is not an input of the reference's backward; the forward kept it inside the geometry records
    a, keep = _args(bg, means3D, colors, means3D.new_ones((1,)), scales, rotations, scale_modifier, cov3D_precomp,
                    viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, False, debug)
    g = N.MomRasterGrads()
    g.dL_dmeans2D, g.dL_dcolors, g.dL_dopacity = g2d.data_ptr(), gcol.data_ptr(), gop.data_ptr()
    g.dL_dmeans3D, g.dL_dcov3D = g3d.data_ptr(), gcov.data_ptr()
    g.dL_dsh = gsh.data_ptr() if gsh.numel() else None
    g.dL_dscales, g.dL_drotations = gsc.data_ptr(), grot.data_ptr()
    dcol = dL_dout_color.contiguous().float()
    ddep = None if dL_dout_depth is None else dL_dout_depth.contiguous().float()
    N.check(lib.mom_raster_backward(C.byref(a), radii.data_ptr(), geomBuffer.data_ptr(), binningBuffer.data_ptr(), int(R),
                                    imageBuffer.data_ptr(), dcol.data_ptr(), None if ddep is None else ddep.data_ptr(),
                                    C.byref(g), N.current_stream()), "mom_raster_backward")
    del keep
    return g2d, gcol, gop, g3d, gcov, gsh, gsc, grot


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:204-223)."""
    lib = N.lib()
    P = means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P:
        m = means3D.contiguous().float()
        v = viewmatrix.contiguous().float()
        p = projmatrix.contiguous().float()
        N.check(lib.mom_mark_visible(P, m.data_ptr(), v.data_ptr(), p.data_ptr(), present.data_ptr(), N.current_stream()),
                "mom_mark_visible")
    return present
