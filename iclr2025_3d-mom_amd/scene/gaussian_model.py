"""GaussianModel: parameter store, Adam groups, LR schedules, densify / prune / opacity reset with optimizer-state
surgery, PLY + .pth checkpoints -- the surface train_4DGS.py / render_4DGS.py / gaussian_renderer use
(reference scene/gaussian_model.py:28-769), device-agnostic and with the per-step work on libmom4d:
one-launch Adam (ops.FusedAdam), fused HexPlane regularisers, HIP distCUDA2."""
import os

import numpy as np
import torch
from torch import nn

from .. import ops
from ..utils.general_utils import (build_rotation, build_scaling_rotation, get_expon_lr_func, inverse_sigmoid,
                                   strip_symmetric)
from ..utils.sh_utils import RGB2SH
from .deformation import deform_network
from .regulation import compute_plane_smoothness  # noqa: F401  (re-exported like the reference module)

_PER_POINT = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


class GaussianModel:
    EMPTY_CACHE_AFTER_PRUNE = False

    def setup_functions(self):
        def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
            L = build_scaling_rotation(scaling_modifier * scaling, rotation)
            return strip_symmetric(L @ L.transpose(1, 2))

        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.covariance_activation = build_covariance_from_scaling_rotation
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    def __init__(self, sh_degree: int, args, device="cuda"):
        self.device = torch.device(device)
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self._deformation = deform_network(args)
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self._deformation_table = e
        self._scene_flow = e
        self.iter_lr = 1.0
        self.setup_functions()

    # ------------------------------------------------------------------ checkpoint tuple (gaussian_model.py:72-115)
    def capture(self):
        return (self.active_sh_degree, self._xyz, self._deformation.state_dict(), self._deformation_table,
                self._scene_flow, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity,
                self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(),
                self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, self._xyz, deform_state, self._deformation_table, self._scene_flow, self._features_dc,
         self._features_rest, self._scaling, self._rotation, self._opacity, self.max_radii2D, xyz_gradient_accum, denom,
         opt_dict, self.spatial_lr_scale) = model_args
        self._deformation.load_state_dict(deform_state)
        self.training_setup(training_args)
        self.xyz_gradient_accum = xyz_gradient_accum
        self.denom = denom
        self.optimizer.load_state_dict(opt_dict)

    # ------------------------------------------------------------------ accessors
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_flow(self):
        return self._scene_flow

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    def get_aabb(self):
        return self._deformation.get_aabb

    def get_covariance(self, scaling_modifier=1):
        return self.covariance_activation(self.get_scaling, scaling_modifier, self._rotation)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ initialisation (gaussian_model.py:153-187)
    def create_from_pcd(self, pcd, spatial_lr_scale: float, time_line: int, TrainData_path: str, flow_scale: int,
                        scene_flow=None):
        """`scene_flow` ([3,P] tensor) may be given directly; otherwise it is read from
        dirname(TrainData_path)/scene_flow.pth exactly like the reference (:183)."""
        from ..simple_knn._C import distCUDA2
        dev = self.device
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).float().to(dev)
        col = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(dev))
        n = pts.shape[0]
        feats = torch.zeros((n, 3, (self.max_sh_degree + 1) ** 2), device=dev)
        feats[:, :3, 0] = col
        print("Number of points at initialisation : ", n)
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=dev)
        rots[:, 0] = 1
        opac = inverse_sigmoid(0.1 * torch.ones((n, 1), dtype=torch.float, device=dev))
        self._xyz = nn.Parameter(pts.requires_grad_(True))
        self._deformation = self._deformation.to(dev)
        self._features_dc = nn.Parameter(feats[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(feats[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opac.requires_grad_(True))
        self.max_radii2D = torch.zeros((n,), device=dev)
        self._deformation_table = torch.gt(torch.ones((n,), device=dev), 0)
        if scene_flow is None:
            scene_flow = torch.load(os.path.join(os.path.dirname(TrainData_path), 'scene_flow.pth'), map_location="cpu")
        flow = (scene_flow.T.float().to(dev) * flow_scale).contiguous()
        print("flow_scale: ", flow_scale)
        self._scene_flow = flow.detach().requires_grad_(False)

    # ------------------------------------------------------------------ optimizer (gaussian_model.py:190-298)
    def training_setup(self, training_args):
        dev = self._xyz.device
        n = self._xyz.shape[0]
        self.percent_dense = training_args.percent_dense
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self._deformation_accum = torch.zeros((n, 3), device=dev)
        s, ta = self.spatial_lr_scale, training_args
        groups = [
            {'params': [self._xyz], 'lr': ta.position_lr_init * s, "name": "xyz"},
            {'params': list(self._deformation.get_mlp_parameters()), 'lr': ta.deformation_lr_init * s, "name": "deformation"},
            {'params': list(self._deformation.get_grid_parameters()), 'lr': ta.grid_lr_init * s, "name": "grid"},
            {'params': [self._features_dc], 'lr': ta.feature_lr, "name": "f_dc"},
            {'params': [self._features_rest], 'lr': ta.feature_lr / 20.0, "name": "f_rest"},
            {'params': [self._opacity], 'lr': ta.opacity_lr, "name": "opacity"},
            {'params': [self._scaling], 'lr': ta.scaling_lr, "name": "scaling"},
            {'params': [self._rotation], 'lr': ta.rotation_lr, "name": "rotation"},
        ]
        self.optimizer = ops.BACKEND.Adam(groups, lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(ta.position_lr_init * s, ta.position_lr_final * s,
                                                    lr_delay_mult=ta.position_lr_delay_mult, max_steps=ta.position_lr_max_steps)
        self.deformation_scheduler_args = get_expon_lr_func(ta.deformation_lr_init * s, ta.deformation_lr_final * s,
                                                            lr_delay_mult=ta.deformation_lr_delay_mult,
                                                            max_steps=ta.position_lr_max_steps)
        self.grid_scheduler_args = get_expon_lr_func(ta.grid_lr_init * s, ta.grid_lr_final * s,
                                                     lr_delay_mult=ta.deformation_lr_delay_mult,
                                                     max_steps=ta.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        """Only xyz / grid / deformation are scheduled (gaussian_model.py:284-298)."""
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g['lr'] = self.xyz_scheduler_args(iteration)
            if "grid" in g["name"]:
                g['lr'] = self.grid_scheduler_args(iteration)
            elif g["name"] == "deformation":
                g['lr'] = self.deformation_scheduler_args(iteration)

    # ------------------------------------------------------------------ PLY / deformation checkpoints (:300-407)
    def construct_list_of_attributes(self):
        names = ['x', 'y', 'z', 'nx', 'ny', 'nz']
        names += [f'f_dc_{i}' for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f'f_rest_{i}' for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append('opacity')
        names += [f'scale_{i}' for i in range(self._scaling.shape[1])]
        names += [f'rot_{i}' for i in range(self._rotation.shape[1])]
        return names

    def load_model(self, path):
        print("loading model from exists{}".format(path))
        dev = self.device
        self._scene_flow = torch.load(os.path.join(path, "scene_flow.pth"), map_location=dev)
        self._deformation.load_state_dict(torch.load(os.path.join(path, "deformation.pth"), map_location=dev))
        self._deformation = self._deformation.to(dev)
        n = self.get_xyz.shape[0]
        self._deformation_table = torch.gt(torch.ones((n,), device=dev), 0)
        self._deformation_accum = torch.zeros((n, 3), device=dev)
        for name in ("deformation_table", "deformation_accum"):
            f = os.path.join(path, name + ".pth")
            if os.path.exists(f):
                setattr(self, "_" + name, torch.load(f, map_location=dev))
        self.max_radii2D = torch.zeros((n,), device=dev)

    def save_deformation(self, path):
        torch.save(self._deformation.state_dict(), os.path.join(path, "deformation.pth"))
        torch.save(self._deformation_table, os.path.join(path, "deformation_table.pth"))
        torch.save(self._deformation_accum, os.path.join(path, "deformation_accum.pth"))
        torch.save(self._scene_flow, os.path.join(path, "scene_flow.pth"))

    def _ply_columns(self):
        xyz = self._xyz.detach().cpu().numpy()
        cols = [xyz, np.zeros_like(xyz),
                self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
                self._opacity.detach().cpu().numpy(), self._scaling.detach().cpu().numpy(),
                self._rotation.detach().cpu().numpy()]
        return np.concatenate(cols, axis=1).astype(np.float32)

    def save_ply(self, path):
        """binary_little_endian PLY, one float32 property per attribute, as plyfile writes it for the reference."""
        from ..utils.ply_io import write_ply
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        write_ply(path, self.construct_list_of_attributes(), self._ply_columns())

    def load_ply(self, path):
        from ..utils.ply_io import read_ply
        names, data = read_ply(path)
        col = {n: i for i, n in enumerate(names)}
        dev = self.device

        def pick(prefix):
            ks = sorted([n for n in names if n.startswith(prefix)], key=lambda x: int(x.split('_')[-1]))
            return np.stack([data[:, col[k]] for k in ks], axis=1)

        xyz = np.stack([data[:, col[k]] for k in ("x", "y", "z")], axis=1)
        n = xyz.shape[0]
        f_dc = np.stack([data[:, col[f"f_dc_{i}"]] for i in range(3)], axis=1).reshape(n, 3, 1)
        f_rest = pick("f_rest_")
        assert f_rest.shape[1] == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_rest = f_rest.reshape(n, 3, (self.max_sh_degree + 1) ** 2 - 1)

        def par(a):
            return nn.Parameter(torch.tensor(a, dtype=torch.float, device=dev).requires_grad_(True))

        self._xyz = par(xyz)
        self._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float, device=dev).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(f_rest, dtype=torch.float, device=dev).transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = par(data[:, col["opacity"]][..., None])
        self._scaling = par(pick("scale_"))
        self._rotation = par(pick("rot"))
        self.active_sh_degree = self.max_sh_degree

    # ------------------------------------------------------------------ optimizer-state surgery (:409-482)
    def _single_groups(self):
        return [g for g in self.optimizer.param_groups if len(g["params"]) == 1]

    def replace_tensor_to_optimizer(self, tensor, name):
        out = {}
        for g in self.optimizer.param_groups:
            if g["name"] != name:
                continue
            old = g['params'][0]
            st = self.optimizer.state.get(old, None)
            st["exp_avg"] = torch.zeros_like(tensor)
            st["exp_avg_sq"] = torch.zeros_like(tensor)
            del self.optimizer.state[old]
            g["params"][0] = nn.Parameter(tensor.requires_grad_(True))
            self.optimizer.state[g['params'][0]] = st
            out[name] = g["params"][0]
        return out

    def _rebuild(self, fn_param, fn_state):
        """Apply fn_param to every per-point parameter and fn_state to both of its Adam moments."""
        out = {}
        for g in self._single_groups():
            old = g["params"][0]
            st = self.optimizer.state.get(old, None)
            new = nn.Parameter(fn_param(g["name"], old).requires_grad_(True))
            if st is not None:
                st["exp_avg"] = fn_state(g["name"], st["exp_avg"])
                st["exp_avg_sq"] = fn_state(g["name"], st["exp_avg_sq"])
                del self.optimizer.state[old]
                self.optimizer.state[new] = st
            g["params"][0] = new
            out[g["name"]] = new
        return out

    def _prune_optimizer(self, mask, extra=()):
        """Keep the rows with mask set in every per-point parameter and both of its Adam moments (reference :409-440), plus in
        the `extra` tensors, all through ONE backend selection; returns (new parameters by group name, selected extras)."""
        groups = self._single_groups()
        flat = []
        for g in groups:
            old = g["params"][0]
            st = self.optimizer.state.get(old, None)
            flat.append(old.detach())
            if st is not None:
                flat += [st["exp_avg"], st["exp_avg_sq"]]
        sel = iter(ops.BACKEND.select_rows(mask, flat + list(extra)))

        def take(n, t):
            return next(sel)
        out = self._rebuild(take, take)
        return out, [next(sel) for _ in extra]

    def cat_tensors_to_optimizer(self, tensors_dict):
        return self._rebuild(lambda n, p: torch.cat((p, tensors_dict[n]), dim=0),
                             lambda n, s: torch.cat((s, torch.zeros_like(tensors_dict[n])), dim=0))

    def _adopt(self, t):
        self._xyz, self._features_dc, self._features_rest = t["xyz"], t["f_dc"], t["f_rest"]
        self._opacity, self._scaling, self._rotation = t["opacity"], t["scaling"], t["rotation"]

    def prune_points(self, mask):
        keep = ~mask
        params, extra = self._prune_optimizer(keep, (self._deformation_accum, self.xyz_gradient_accum, self._deformation_table,
                                                      self.denom, self.max_radii2D, self._scene_flow))
        self._adopt(params)
        (self._deformation_accum, self.xyz_gradient_accum, self._deformation_table, self.denom, self.max_radii2D,
         self._scene_flow) = extra

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation,
                              new_deformation_table, new_sceneflow):
        d = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
             "scaling": new_scaling, "rotation": new_rotation, "scene_flow": new_sceneflow}
        self._adopt(self.cat_tensors_to_optimizer(d))
        dev, n = self._xyz.device, self._xyz.shape[0]
        self._deformation_table = torch.cat([self._deformation_table, new_deformation_table], -1)
        # all statistics restart after every densification (:505-508)
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self._deformation_accum = torch.zeros((n, 3), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n,), device=dev)
        self._scene_flow = torch.cat([self._scene_flow, new_sceneflow])

    # ------------------------------------------------------------------ densify / prune (:511-581,681-715)
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        dev, n0 = self._xyz.device, self.get_xyz.shape[0]
        padded = torch.zeros((n0,), device=dev)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        if not sel.any():
            return
        # the selected rows of every tensor the split needs, through one backend selection (the reference indexes each with the
        # mask: scaling twice, rotation twice, xyz, both SH parts, opacity, the deformation table, the flow)
        xyz_s, fdc_s, frest_s, opac_s, scal_raw_s, rot_s, table_s, flow_s = ops.BACKEND.select_rows(
            sel, [self._xyz.detach(), self._features_dc.detach(), self._features_rest.detach(), self._opacity.detach(),
                  self._scaling.detach(), self._rotation.detach(), self._deformation_table, self.get_flow])
        scal_s = self.scaling_activation(scal_raw_s)
        stds = scal_s.repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
        rots = build_rotation(rot_s).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + xyz_s.repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(scal_s.repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, fdc_s.repeat(N, 1, 1), frest_s.repeat(N, 1, 1), opac_s.repeat(N, 1), new_scaling,
                                   rot_s.repeat(N, 1), table_s.repeat(N), flow_s.repeat(N, 1))
        self.prune_points(torch.cat((sel, torch.zeros(N * xyz_s.shape[0], device=dev, dtype=bool))))

    def densify_and_clone(self, grads, grad_threshold, scene_extent, density_threshold=20, displacement_scale=20,
                          model_path=None, iteration=None, stage=None):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & \
              (torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        picked = ops.BACKEND.select_rows(sel, [self._xyz.detach(), self._features_dc.detach(), self._features_rest.detach(),
                                               self._opacity.detach(), self._scaling.detach(), self._rotation.detach(),
                                               self._deformation_table, self._scene_flow])
        self.densification_postfix(*picked)

    def prune(self, max_grad, min_opacity, extent, max_screen_size):
        mask = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            mask = mask | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(mask)
        # The reference empties the caching allocator here (scene/gaussian_model.py:692).  It changes no result; on this path it
        # made the next densify round re-acquire its buffers from the driver -- 30 ms per large allocation, 80-90 ms per round,
        # a third of the training time at a 100-iteration cadence (tools/boundary_cost.py).  EMPTY_CACHE_AFTER_PRUNE restores it.
        if self.EMPTY_CACHE_AFTER_PRUNE and torch.cuda.is_available():
            torch.cuda.empty_cache()

    def densify(self, max_grad, min_opacity, extent, max_screen_size, density_threshold, displacement_scale, model_path=None,
                iteration=None, stage=None):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent, density_threshold, displacement_scale, model_path, iteration, stage)
        self.densify_and_split(grads, max_grad, extent)

    def grow(self, *a, **k):
        """Not part of the claimed surface: the reference's grow() (scene/gaussian_model.py:647-680) cannot run
        either -- add_point_by_mask passes densification_postfix seven of its eight arguments (:629 against :476), and downsample_point /
        addpoint need open3d, which the reference's environment does not install -- and opt.add_point is False in every shipped
        configuration (train_4DGS.py:285 is the only caller)."""
        raise NotImplementedError("GaussianModel.grow(): opt.add_point=True does not run in the reference either (gaussian_model.py:629 "
                                  "passes densification_postfix 7 of its 8 arguments; downsample_point needs open3d)")

    def reset_opacity(self):
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity = self.replace_tensor_to_optimizer(new, "opacity")["opacity"]

    def update_densification_stats(self, radii, viewspace_grad, skip_flag=None, stream=None):
        """The per-iteration bookkeeping of train_4DGS.py:266 and add_densification_stats (:713-715) for the Gaussians with
        radii > 0, in one backend call (one HIP kernel), in place.  skip_flag: see ops.densify_stats."""
        g = viewspace_grad.detach()
        if g.dim() != 2 or g.shape[1] != 3 or not g.is_contiguous():
            g = g.reshape(-1, 3).contiguous()
        kw = {}
        if skip_flag is not None:
            kw["skip_flag"] = skip_flag
        if stream is not None:             # (a raw stream handle: the fused step's second stream, HIP backend only)
            kw["stream"] = stream
        ops.BACKEND.densify_stats(radii.contiguous(), g.float(), self.max_radii2D, self.xyz_gradient_accum, self.denom, **kw)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        # masked accumulate == the reference's boolean-mask indexing (:713-715) without nonzero()'s host sync
        m = update_filter.unsqueeze(-1).to(self.xyz_gradient_accum.dtype)
        self.xyz_gradient_accum += torch.norm(viewspace_point_tensor[:, :2], dim=-1, keepdim=True) * m
        self.denom += m

    @torch.no_grad()
    def update_deformation_table(self, threshold):
        self._deformation_table = torch.gt(self._deformation_accum.max(dim=-1).values / 100, threshold)

    # ------------------------------------------------------------------ regularisers (:730-769)
    def _reg_terms(self):
        grids = self._deformation.deformation_net.grid.grids
        return [(g, i) for g in grids if len(g) != 3 for i in range(6)]

    def _plane_regulation(self):
        return sum(compute_plane_smoothness(g[i]) for g, i in self._reg_terms() if i in (0, 1, 3))

    def _time_regulation(self):
        return sum(compute_plane_smoothness(g[i]) for g, i in self._reg_terms() if i in (2, 4, 5))

    def _l1_regulation(self):
        return sum(torch.abs(1 - g[i]).mean() for g, i in self._reg_terms() if i in (2, 4, 5))

    def compute_regulation(self, time_smoothness_weight, l1_time_planes_weight, plane_tv_weight):
        """plane_tv * sum smooth2(space planes) + time_smoothness * sum smooth2(space-time planes)
        + l1_time_planes * sum mean|1 - space-time planes|, as one fused HIP pass over the 12 planes."""
        # the planes and their weights, remembered until a plane object or a weight changes: walking the nn.ParameterLists costs
        # 2 us per plane (container.__getitem__), 30 us per iteration of the render() path
        grid = self._deformation.deformation_net.grid
        key = (time_smoothness_weight, l1_time_planes_weight, plane_tv_weight)
        c = getattr(self, "_reg_cache", None)
        if c is None or c[0] is not grid or c[1] != key or (c[2] and c[2][0] is not grid.grids[0][0]):
            planes, ws, wl = [], [], []
            for g, i in self._reg_terms():
                planes.append(g[i])
                ws.append(time_smoothness_weight if i in (2, 4, 5) else plane_tv_weight)
                wl.append(l1_time_planes_weight if i in (2, 4, 5) else 0.0)
            # (a replaced plane object -- a rebuilt grid; load_state_dict copies into the existing ones -- shows in the first plane)
            c = self._reg_cache = (grid, key, planes, ws, wl)
        planes, ws, wl = c[2], c[3], c[4]
        if not planes:
            return 0.0
        return ops.BACKEND.plane_regulation(planes, ws, wl)
