"""The render script's body (reference render_4DGS.py:45-91: render_set / render_sets) on libmom4d.

What render_set produces is the reference's: <model_path>/frame_result/<name>/%05d.png -- every frame of the path, quantised
like torchvision.utils.save_image -- and, when imageio is installed, <model_path>/vid_result/<name>.mp4 of the frames cropped by
32 pixels.  How it gets there differs: the reference encodes each PNG inside the render loop, on the thread that also launches
the kernels, so its printed FPS is the PNG encoder's (render_4DGS.py:60-71).  Here the loop only enqueues GPU work -- render,
one quantisation kernel (mom_image_to_rgb8), an async copy into a ring of pinned host buffers -- and a pool of threads encodes
finished frames behind it (PIL releases the GIL inside zlib).  `scripted=True` keeps the reference's blocking order for
comparison.  Frames whose binning buffer overflowed in async mode are rendered again at the end (FusedRender.overflowed)."""
import contextlib
import os
import queue
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _native as N
from . import gaussian_renderer as GR
from .gaussian_renderer import render
from .utils.image_io import save_image, save_uint8


# zlib level of the asynchronous writer's PNGs.  PNG is lossless at every level: the decoded pixels are the same bytes, only the
# files are larger (1.2 MB instead of 1.0 MB per 960x540 frame of the synthetic scene) -- and deflate is what the as-scripted
# render loop waits for (a frame takes 0.26 ms to render and ~35 ms of one host core to compress at level 6, torchvision's / PIL's
# default, which the blocking order `scripted=True` keeps).
PNG_COMPRESS_LEVEL = int(os.environ.get("MOM_PNG_LEVEL", "1"))


class AsyncPNGWriter:
    """submit(image [3,H,W] on the GPU, path): quantise on the device, copy to a pinned slot behind the frame's kernels, encode on
    a worker thread.  At most `slots` frames are in flight; submit() blocks only when all slots are busy."""

    def __init__(self, H, W, C=3, slots=48, workers=None, compress_level=None):
        self.H, self.W, self.C = H, W, C
        self.level = PNG_COMPRESS_LEVEL if compress_level is None else int(compress_level)
        self.host = [torch.empty((H, W, C), dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.dev = [None] * slots
        self.free = queue.Queue()
        for i in range(slots):
            self.free.put(i)
        self.pool = ThreadPoolExecutor(max_workers=workers or min(slots, max(4, (os.cpu_count() or 8) // 2)))
        self.futures = []
        list(self.pool.map(lambda _: time.sleep(0.002), range(self.pool._max_workers)))     # start the threads now, not per frame

    def submit(self, image, path):
        slot = self.free.get()
        if self.dev[slot] is None or self.dev[slot].device != image.device:
            self.dev[slot] = torch.empty((self.H, self.W, self.C), dtype=torch.uint8, device=image.device)
        img = image if (image.is_contiguous() and image.dtype == torch.float32) else image.contiguous().float()
        N.check(N.lib().mom_image_to_rgb8(self.C, self.H, self.W, img.data_ptr(), self.dev[slot].data_ptr(), N.current_stream()),
                "mom_image_to_rgb8")
        self.host[slot].copy_(self.dev[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.futures.append(self.pool.submit(self._encode, slot, ev, path))

    def _encode(self, slot, ev, path):
        try:
            ev.synchronize()
            save_uint8(self.host[slot].numpy(), path, self.level)
        finally:
            self.free.put(slot)

    def drain(self):
        for f in self.futures:
            f.result()
        self.futures.clear()

    def close(self):
        self.drain()
        self.pool.shutdown()


# Streams render_set deals consecutive frames to when it writes them asynchronously (1: every frame on the current stream).  Two,
# not three: the device has four hardware queues, and a process that has trained (the step's two streams) and then renders on three
# more lands two of them on one queue -- 4400 frames/s instead of 5700, where two streams give 5300 every time
# (tools/probe/fps_leg.py).
RENDER_STREAMS = 2

to8b = lambda x: (255 * np.clip(x.cpu().numpy(), 0, 1)).astype(np.uint8)      # render_4DGS.py:44


def render_set(model_path, name, iteration, views, gaussians, pipeline, background, cam_type, delta_scale=1, scripted=False,
               video=True, writer=None):
    """Returns {"frames", "seconds", "fps"}: fps = (frames - 1) / seconds from the start of the first render to the last PNG on
    disk (the reference's definition, render_4DGS.py:61,70-71, made honest about the writes it overlaps)."""
    render_path = os.path.join(model_path, 'frame_result', name)
    os.makedirs(render_path, exist_ok=True)
    print("point nums:", gaussians._xyz.shape[0])
    views = list(views)
    fr = None
    streams_before = GR.render_streams()
    if gaussians._xyz.is_cuda:
        # the forward-only launch sequence reports async-mode overflows per frame; collect them instead of raising mid-loop
        from .fused_render import FusedRender, FusedRenderPool
        if not scripted and RENDER_STREAMS > 1:
            # consecutive frames on alternating streams (FusedRenderPool): every frame is quantised and copied to the host on the
            # stream it was rendered on, so nothing here ever waits for a frame on another stream
            GR.set_render_streams(RENDER_STREAMS)
            fr = getattr(gaussians, "_fused_render_pool", None)
            if fr is None or fr.n != RENDER_STREAMS:
                fr = gaussians._fused_render_pool = FusedRenderPool(gaussians, RENDER_STREAMS)
        else:
            GR.set_render_streams(1)
            fr = getattr(gaussians, "_fused_render", None)
            if fr is None:
                fr = gaussians._fused_render = FusedRender(gaussians)
        fr.collect, fr.bad = True, []
    try:
        return _render_set_body(model_path, name, views, gaussians, pipeline, background, cam_type, delta_scale, scripted, video, writer,
                                render_path, fr)
    finally:
        GR.set_render_streams(streams_before)
        if fr is not None:
            fr.collect = False


def _render_set_body(model_path, name, views, gaussians, pipeline, background, cam_type, delta_scale, scripted, video, writer,
                     render_path, fr):
    own_writer = None
    crop = 32
    frames, images = [], []
    with torch.no_grad():
        t0 = time.time()
        pooled = fr is not None and hasattr(fr, "slots")
        first = fr.count if pooled else (fr.serial if fr is not None else 0)      # the loop's frame 0 in the renderer's numbering
        for idx, view in enumerate(views):
            out = render(view, gaussians, pipeline, background, cam_type=cam_type, delta_scale=delta_scale)
            rendering = out["render"]
            path = os.path.join(render_path, '{0:05d}'.format(idx) + ".png")
            if scripted or not rendering.is_cuda:
                save_image(rendering, path)                      # blocking, inside the loop: the reference's order
            else:
                if writer is None and own_writer is None:
                    own_writer = AsyncPNGWriter(rendering.shape[1], rendering.shape[2])
                with (torch.cuda.stream(out["stream"]) if "stream" in out else contextlib.nullcontext()):
                    (writer or own_writer).submit(rendering, path)
            frames.append((idx, view, path))
            if video:
                images.append(rendering)
        w = writer or own_writer
        if w is not None:
            # async binning: frames flagged as overflowed are rendered again (the capacity was raised), before the clock stops
            bad = (fr.bad + fr.overflowed()) if fr is not None else []
            if fr is not None:
                fr.bad = []
            if bad:
                from .diff_gaussian_rasterization import _C as RC
                mode = RC._state["mode"]
                RC._state["mode"] = "exact"                       # the repairs size their buffer from their own instance count
                GR.set_render_streams(1)                          # ... on the current stream, one after the other
                torch.cuda.synchronize()
                try:
                    for s in bad:
                        idx = (s - first) if pooled else (s - first - 1)       # a pool numbers frames from 0, a FusedRender's serial counts from 1
                        if 0 <= idx < len(views):
                            rendering = render(views[idx], gaussians, pipeline, background, cam_type=cam_type, delta_scale=delta_scale)["render"]
                            w.submit(rendering, frames[idx][2])
                            if video:
                                images[idx] = rendering
                finally:
                    RC._state["mode"] = mode
            w.drain()
        if rendering.is_cuda:
            torch.cuda.synchronize()
        dt = time.time() - t0
    fps = (len(views) - 1) / dt if dt > 0 else float("inf")
    print("FPS:", fps)
    if own_writer is not None:
        own_writer.close()
    if video:
        try:
            import imageio
            video_path = os.path.join(model_path, 'vid_result')
            os.makedirs(video_path, exist_ok=True)
            imageio.mimwrite(os.path.join(video_path, name + '.mp4'),
                             [to8b(im).transpose(1, 2, 0)[crop:-crop, crop:-crop] for im in images], fps=30)
        except ImportError:
            print("imageio is not installed: skipping vid_result/" + name + ".mp4 (the PNG frames are complete)")
    return {"frames": len(views), "seconds": dt, "fps": fps}


def render_sets(dataset, hyperparam, iteration, pipeline, skip_train, skip_test, skip_video, TrainData_path, Gaussian_path, **kw):
    """render_4DGS.py:77-91.  One AsyncPNGWriter serves the four trajectories."""
    from .scene import GaussianModel, Scene
    with torch.no_grad():
        gaussians = GaussianModel(dataset.sh_degree, hyperparam)
        scene = Scene(TrainData_path, Gaussian_path, dataset, gaussians, load_iteration=iteration, shuffle=False)
        dev = gaussians._xyz.device
        background = torch.tensor([1, 1, 1] if dataset.white_background else [0, 0, 0], dtype=torch.float32, device=dev)
        out = {}
        writer = None
        if dev.type == "cuda" and not kw.get("scripted") and "writer" not in kw:
            c0 = scene.getVideoCameras_up()[0]
            writer = kw["writer"] = AsyncPNGWriter(int(c0.image_height), int(c0.image_width))
        try:
            for name, cams in (("up_down", scene.getVideoCameras_up()), ("side", scene.getVideoCameras_side()),
                               ("zoom", scene.getVideoCameras_zoom()), ("circle", scene.getVideoCameras_circle())):
                out[name] = render_set(Gaussian_path, name, scene.loaded_iter, cams, gaussians, pipeline, background,
                                       scene.dataset_type, **kw)
        finally:
            if writer is not None:
                writer.close()
    return out
