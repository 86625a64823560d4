"""The per-iteration body of the reference's scene_reconstruction (train_4DGS.py:119-301), in the same order:
LR update -> SH-degree bump -> camera draw -> render -> loss -> backward -> densification statistics ->
densify / prune / opacity reset -> Adam step.  Host syncs the reference pays every iteration (loss.item(),
isnan, the num_rendered read-back, H2D of matrices and ground truth) are optional here."""
from collections import deque
from random import randint

import torch

from .diff_gaussian_rasterization import _C as RC
from .gaussian_renderer import render
from .utils.image_utils import psnr
from .utils.loss_utils import l1_loss, psnr_from_last_l1, ssim


class Trainer:
    def __init__(self, scene, gaussians, opt, hyper, pipe, stage="fine", delta_scale=1, white_background=False,
                 sync_every_step=True, fused=False, gc_freeze=False):
        self.scene, self.g, self.opt, self.hyper, self.pipe, self.stage = scene, gaussians, opt, hyper, pipe, stage
        self.delta_scale = delta_scale
        dev = gaussians._xyz.device
        self.background = torch.tensor([1, 1, 1] if white_background else [0, 0, 0], dtype=torch.float32, device=dev)
        gaussians.training_setup(opt)
        self.cams = list(scene.getTrainCameras() if stage == "coarse" else scene.getTrainCameras_2())
        self.stack = list(self.cams)
        self.sync_every_step = sync_every_step
        self.ema_loss, self.ema_psnr = 0.0, 0.0
        self.last = {}
        self.dist = None     # set by parallel.attach(): camera-batch shard, one camera per rank
        # fused=True: the iteration runs as an explicit launch sequence (fused_step.py) instead of render()+autograd
        self.fused = None
        if fused and stage == "fine" and opt.batch_size == 1:
            from .fused_step import FusedStep
            self.fused = FusedStep(gaussians, opt, hyper, self.background)
        # every fused step since the last verified one: (serial, iteration, camera); and the overflow-word read-backs in flight:
        # (serial of the last step they cover, ring slot, event)
        self._log = deque()
        self._checks = deque()
        self._alog = deque()     # async autograd path: (first forward, last forward, iteration, cameras) of the unverified iterations
        self._serial = 0
        self.replayed = 0    # iterations replayed after a binning overflow (diagnostics)
        if gc_freeze:
            # (opt-in: a training script sets it once its scene and model exist; it is process-wide.)  Everything alive now --
            # the scene, its cameras, the model, torch itself -- stays for the whole run: move it out of the cyclic collector's
            # reach.  Otherwise a full collection walks that heap every few hundred iterations and the
            # training loop stands still for 100-150 ms each time (tools/probe/autograd_rate.py: 4.9 against 1.5 ms per
            # iteration over a 40-iteration window that contains one).
            import gc
            gc.collect()
            gc.freeze()

    # The host enqueues a fused step without knowing whether the step's binning buffer will be large enough (it is sized
    # from earlier frames; waiting for the count would be the reference's per-iteration sync, rasterizer_impl.cu:282).
    # A step that does not fit leaves its tag in a sticky device word; from then on every Adam / statistics launch is a no-op
    # (FusedStep.flags).  The host copies the word back every CHECK_EVERY steps and looks at a copy FLAG_LAG read-backs later
    # -- a fixed lag, so that all ranks of a multi-GPU run reach the same decision at the same step -- and, if it is set,
    # clears it and replays the logged iterations from the tagged one on with exactly sized buffers.  The model therefore
    # always equals what a fully synchronous run would have produced.
    CHECK_EVERY, FLAG_LAG = 4, 2

    def _boundary(self, iteration):
        """Iterations whose host logic reads or restructures model state (densify / prune / opacity reset / SH bump): every
        earlier step must be verified first, and the step itself runs with an exactly sized buffer."""
        o = self.opt
        return (iteration % o.densification_interval == 0 or iteration % o.pruning_interval == 0
                or iteration % o.opacity_reset_interval == 0 or iteration % 1000 == 0)

    def _post_check(self):
        if self._log and (not self._checks or self._checks[-1][0] != self._log[-1][0]):
            slot = len(self._checks) % self.fused.RING if not self._checks else (self._checks[-1][1] + 1) % self.fused.RING
            self._checks.append((self._log[-1][0], slot, self.fused.post_flag(slot)))

    def _poll(self, lag):
        while len(self._checks) > lag:
            upto, slot, ev = self._checks.popleft()
            ev.synchronize()
            tag = int(self.fused.flag_ring[slot])
            if tag != 0:
                self._recover(tag)
                return
            while self._log and self._log[0][0] <= upto:
                self._log.popleft()

    def drain(self):
        """Every step enqueued so far has been applied to the model (replaying the ones an overflow skipped)."""
        if self.fused is not None:
            self._post_check()
            self._poll(0)
        elif RC._state["mode"] == "async" and self._alog:
            self.drain_autograd()

    def save(self, iteration, stage=None):
        """scene.save() behind drain(): a fused (or async) run may hold a few iterations the device skipped after a binning
        overflow until the host replays them -- anything that reads the model between steps (checkpoints, evaluation renders)
        drains first.  Trainer.step() does so itself at every iteration that restructures the model."""
        self.drain()
        if self.fused is not None:
            self.fused.gather_moments()
        self.scene.save(iteration, stage or self.stage)

    def _recover(self, tag):
        torch.cuda.synchronize()
        self._checks.clear()
        # the step tagged `tag` overflowed; it and every later one were no-ops on the device; all earlier ones are in
        entries = list(self._log)
        self._log.clear()
        first = next((i for i, e in enumerate(entries) if ((e[0] & 0x7FFFFFFF) + 1) == tag), 0)
        redo = entries[first:]
        self.g.optimizer.rewind(len(redo))      # the host counted steps the device skipped
        self.fused.flags.zero_()
        for _, it, cam in redo:
            self.g.update_learning_rate(it)
            self.fused.exact_next()
            self._step_fused(it, cam, replay=True)
            self.replayed += 1

    def _draw(self):
        cams = []
        while len(cams) < self.opt.batch_size:
            cams.append(self.stack.pop(randint(0, len(self.stack) - 1)))
            if not self.stack:
                self.stack = list(self.cams)
        return cams

    def step(self, iteration, cams=None):
        g, opt, hyper = self.g, self.opt, self.hyper
        if self.fused is not None:
            # before anything of this iteration touches host state: earlier fused steps are verified (and replayed if an
            # overflow skipped them) -- all of them at an iteration that restructures the model, else up to a fixed lag
            if self._boundary(iteration):
                self.drain()
                self.fused.exact_next()
                self.fused.gather_moments()        # (sharded Adam: the round below reads / restructures every slice's moments)
            else:
                if len(self._log) and len(self._log) % self.CHECK_EVERY == 0:
                    self._post_check()
                self._poll(self.FLAG_LAG)
        elif RC._state["mode"] == "async" and self._boundary(iteration):
            self.drain_autograd()      # async autograd path: the same rule
        g.update_learning_rate(iteration)
        if iteration % 1000 == 0:
            g.oneupSHdegree()
        cams = cams or self._draw()
        if self.fused is not None and len(cams) == 1:
            return self._step_fused(iteration, cams[0])
        if RC._state["mode"] == "async":
            # the rasterizer sizes its binning buffer from earlier frames and reports an overflow a few forwards later (render()
            # raises BinningOverflow): replay what the overflow skipped (_recover_autograd), then take this iteration again
            first = RC._state["serial"] + 1
            try:
                loss = self._step_autograd(iteration, cams)
            except RC.BinningOverflow as e:
                self._recover_autograd(e.serial)
                g.update_learning_rate(iteration)
                g.optimizer.zero_grad(set_to_none=True)
                first = RC._state["serial"] + 1
                loss = self._step_autograd(iteration, cams)       # the capacity hint has been doubled
            self._alog.append((first, RC._state["serial"], iteration, list(cams)))
            while self._alog and self._alog[0][1] <= RC._state["verified"]:
                self._alog.popleft()
            return loss
        return self._step_autograd(iteration, cams)

    def _recover_autograd(self, serial):
        """Async forward number `serial` overflowed: the optimizer steps and statistics updates of its iteration and of every
        later one were no-ops on the device (FusedAdam.skip_flag / the statistics kernel test RC.overflow_flag()).  Replay those
        iterations with exactly sized buffers, as _recover does for the fused step."""
        redo = [e for e in self._alog if e[1] >= serial]
        self._alog.clear()
        self.g.optimizer.rewind(len(redo))          # the host counted steps the device skipped
        RC._state["mode"] = "exact"
        try:
            for _, _, it, cams in redo:
                self.g.update_learning_rate(it)
                self._step_autograd(it, cams)
                self.replayed += 1
        finally:
            RC._state["mode"] = "async"

    def drain_autograd(self):
        """Async autograd path: every iteration enqueued so far has been applied (or replayed)."""
        try:
            RC._check_overflow(0)
        except RC.BinningOverflow as e:
            self._recover_autograd(e.serial)
        self._alog.clear()

    def _step_autograd(self, iteration, cams):
        g, opt, hyper = self.g, self.opt, self.hyper
        self._skip = None
        if RC._state["mode"] == "async":
            self._skip = g.optimizer.skip_flag = RC.overflow_flag(g._xyz.device)
        images, gts, radii_l, vis_l, vsp_l = [], [], [], [], []
        for cam in cams:
            pkg = render(cam, g, self.pipe, self.background, stage=self.stage, cam_type=self.scene.dataset_type,
                         delta_scale=self.delta_scale)
            images.append(pkg["render"].unsqueeze(0))
            gts.append(cam.device_tensors(self.background.device)[3].unsqueeze(0))
            radii_l.append(pkg["radii"].unsqueeze(0))
            vis_l.append(pkg["visibility_filter"].unsqueeze(0))
            vsp_l.append(pkg["viewspace_points"])
        if len(cams) == 1:      # a batch of one: the reference's cat / max / any over the batch dimension are identities
            radii, visibility, image, gt = radii_l[0][0], vis_l[0][0], images[0], gts[0]
        else:
            radii = torch.cat(radii_l, 0).max(dim=0).values
            visibility = torch.cat(vis_l).any(dim=0)
            image = torch.cat(images, 0)
            gt = torch.cat(gts, 0)

        Ll1 = l1_loss(image, gt[:, :3, :, :])
        loss = Ll1
        if self.stage == "fine" and hyper.time_smoothness_weight != 0:
            loss = loss + g.compute_regulation(hyper.time_smoothness_weight, hyper.l1_time_planes, hyper.plane_tv_weight)
        if opt.lambda_dssim != 0:
            loss = loss + opt.lambda_dssim * (1.0 - ssim(image, gt))
        loss.backward()

        vsp_grad = torch.zeros_like(vsp_l[0])
        for v in vsp_l:
            vsp_grad = vsp_grad + v.grad
        if self.dist is not None:
            self.dist.sync_param_grads(g.optimizer)
            radii, visibility, vsp_grad = self.dist.sync_stats(radii, vsp_grad)
            self.dist.seed_for(iteration)

        with torch.no_grad():
            if self.sync_every_step:
                # the reference pays these syncs every iteration (train_4DGS.py:224,234-235)
                if torch.isnan(loss).any():
                    raise FloatingPointError("loss is nan")
                self.ema_loss = 0.4 * loss.item() + 0.6 * self.ema_loss
                # the reference logs psnr(image, gt).mean(): the mean of the per-image PSNRs (train_4DGS.py:214), which the pooled
                # squared error of the fused L1 pass only equals for a batch of one
                ps = psnr_from_last_l1() if image.shape[0] == 1 else psnr(image.detach(), gt).mean()
                self.ema_psnr = 0.4 * float(ps) + 0.6 * self.ema_psnr
            self.last = {"loss": loss.detach(), "l1": Ll1.detach(), "points": g._xyz.shape[0]}
        loss = self._after_backward(iteration, loss.detach(), radii, visibility, vsp_grad)
        self._skip = None
        return loss

    def _early_adam(self, params, stream=None, ranges=None):
        """Runs on the fused step's second stream (`stream`: its raw handle; None: the current stream), right after the activation
        backward: Adam for the appearance parameters and -- it only needs the radii and the screen-space gradient, both final by
        then -- the densification statistics."""
        self.g.optimizer.skip_flag = self.fused.flags
        self.g.optimizer.step_partial(params, stream=stream, ranges=ranges)       # ranges: this rank's slice (sharded Adam, parallel.py)
        if self._early_iter < self.opt.densify_until_iter:
            self.g.update_densification_stats(self.fused.radii, self.fused.g2d, skip_flag=self.fused.flags, stream=stream)
            self._stats_done = True

    def _step_fused(self, iteration, cam, replay=False):
        with torch.no_grad():
            if not replay:
                self._serial += 1
            self.fused.next_tag = (self._serial & 0x7FFFFFFF) + 1
            # Adam for the appearance parameters starts inside the step, beside the deformation backward -- on iterations whose
            # host logic touches no parameter between backward and optimizer.step() (a densify / prune round replaces them and
            # the reference's step() then skips them: train_4DGS.py:266-297)
            early = None
            self._stats_done = False
            if iteration < self.opt.iterations and not self._boundary(iteration):
                early = self._early_adam
                self._early_iter = iteration
                g_ = self.g
                g_.optimizer.ensure_state([g_._features_dc, g_._features_rest, g_._scaling, g_._rotation, g_._opacity])
            loss, radii, vsp_grad = self.fused.forward_backward(cam, self.delta_scale, early_adam=early)
            self.g.optimizer.skip_flag = self._skip = self.fused.flags
            if self.dist is not None:
                # the step began its all-reduces as each bucket became final (fused_step.py); radii and vsp_grad come
                # back reduced in place (largest radius over the ranks, mean screen-space gradient)
                self.dist.finish()
                if self._boundary(iteration):
                    # (the one random draw a step can make is densify_and_split's, at a boundary iteration: two torch calls -- 25 us
                    # of a rank's host time -- spared on the others)
                    self.dist.seed_for(iteration)
                if not getattr(self.dist, "verified", True):
                    # first step of a world > 1: what the collectives should have made identical on every rank IS identical, or the job stops
                    fs = self.fused
                    torch.cuda.synchronize()
                    chk = [fs.ibucket, fs._dg_flat] + ([fs.pflat, fs.g2d_flat] if (fs._chunk and early is not None) else [fs.early_bucket])
                    if self.dist.mode == "tile-row":
                        chk += [fs.pts, fs.rot_d]            # the gathered deformed state (first and last of the five arrays)
                    self.dist.verify_replicas(chk)
            visibility = None          # the statistics kernel derives it from the radii (update_densification_stats)
            if self.sync_every_step:
                if torch.isnan(loss.tensor()).any():
                    raise FloatingPointError("loss is nan")
                self.ema_loss = 0.4 * loss.item() + 0.6 * self.ema_loss
            self.last = {"loss": loss, "points": self.g._xyz.shape[0]}
        loss = self._after_backward(iteration, loss, radii, visibility, vsp_grad)
        self._skip = None
        if not replay:
            self._log.append((self._serial, iteration, cam))
        return loss

    def _after_backward(self, iteration, loss, radii, visibility, vsp_grad):
        g, opt = self.g, self.opt
        with torch.no_grad():

            if iteration < opt.densify_until_iter:
                # same values as the reference's boolean-mask indexing (train_4DGS.py:266), without the host sync a
                # nonzero() costs: invisible entries keep their old value
                if not getattr(self, "_stats_done", False):      # (the fused step may have run them on its second stream)
                    g.update_densification_stats(radii, vsp_grad, skip_flag=getattr(self, "_skip", None))
                self._stats_done = False
                if self.stage == "coarse":
                    op_thr, de_thr = opt.opacity_threshold_coarse, opt.densify_grad_threshold_coarse
                else:
                    f = iteration / opt.densify_until_iter
                    op_thr = opt.opacity_threshold_fine_init - f * (opt.opacity_threshold_fine_init - opt.opacity_threshold_fine_after)
                    de_thr = opt.densify_grad_threshold_fine_init - f * (opt.densify_grad_threshold_fine_init - opt.densify_grad_threshold_after)
                n = g.get_xyz.shape[0]
                size_thr = 20 if iteration > opt.opacity_reset_interval else None
                if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0 and n < 360000:
                    g.densify(de_thr, op_thr, self.scene.cameras_extent, size_thr, 5, 5, self.scene.model_path, iteration, self.stage)
                n = g.get_xyz.shape[0]
                if iteration > opt.pruning_from_iter and iteration % opt.pruning_interval == 0 and n > 200000:
                    g.prune(de_thr, op_thr, self.scene.cameras_extent, size_thr)
                if iteration % opt.opacity_reset_interval == 0:
                    g.reset_opacity()

            if iteration < opt.iterations:
                g.optimizer.step()
                g.optimizer.zero_grad(set_to_none=True)
        return loss
