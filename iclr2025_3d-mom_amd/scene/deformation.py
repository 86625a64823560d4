"""deform_network / Deformation with the reference's module tree, state_dict keys and forward signature
(reference scene/deformation.py:16-242).  The HexPlane lookup is a fused HIP op; the positional encodings the
reference computes and then discards (poc_fre, deformation.py:205-207 -- only the un-encoded leading columns
are ever read) are not computed."""
import torch
import torch.nn as nn
import torch.nn.init as init

from .hexplane import HexPlaneField


def poc_fre(input_data, poc_buf):
    """deformation.py:236-242 (kept for API parity; not on the hot path)."""
    emb = (input_data.unsqueeze(-1) * poc_buf).flatten(-2)
    return torch.cat([input_data, emb.sin(), emb.cos()], -1)


def _head(width, out):
    return nn.Sequential(nn.ReLU(), nn.Linear(width, width), nn.ReLU(), nn.Linear(width, out))


class Deformation(nn.Module):
    def __init__(self, D=8, W=256, input_ch=27, input_ch_time=9, grid_pe=0, skips=[], args=None):
        super().__init__()
        self.D, self.W, self.input_ch, self.input_ch_time, self.skips, self.grid_pe = D, W, input_ch, input_ch_time, skips, grid_pe
        self.no_grid = args.no_grid
        self.grid = HexPlaneField(args.bounds, args.kplanes_config, args.multires)
        self.args = args
        if getattr(args, "empty_voxel", False):
            raise NotImplementedError("empty_voxel=True (DenseGrid) is unreachable with the shipped defaults")
        if args.static_mlp:
            self.static_mlp = _head(self.W, 1)
        self.ratio = 0
        self.create_net()

    @property
    def get_aabb(self):
        return self.grid.get_aabb

    def set_aabb(self, xyz_max, xyz_min):
        print("Deformation Net Set aabb", xyz_max, xyz_min)
        self.grid.set_aabb(xyz_max, xyz_min)

    def create_net(self):
        grid_out = self.grid.feat_dim * (3 if self.grid_pe != 0 else 1)
        layers = [nn.Linear(4 if self.no_grid else grid_out, self.W)]
        for _ in range(self.D - 1):
            layers += [nn.ReLU(), nn.Linear(self.W, self.W)]
        self.feature_out = nn.Sequential(*layers)
        self.pos_deform = _head(self.W, 3)
        self.scales_deform = _head(self.W, 3)
        self.rotations_deform = _head(self.W, 4)
        self.opacity_deform = _head(self.W, 1)
        self.shs_deform = _head(self.W, 16 * 3)

    def query_time(self, rays_pts_emb, scales_emb, rotations_emb, time_feature, time_emb):
        if self.no_grid:
            t = time_emb[:, :1] if torch.is_tensor(time_emb) else torch.full_like(rays_pts_emb[:, :1], float(time_emb))
            hidden = torch.cat([rays_pts_emb[:, :3], t], -1)
        else:
            t = time_emb[:, :1] if torch.is_tensor(time_emb) else time_emb
            hidden = self.grid(rays_pts_emb[:, :3], t)
            if self.grid_pe > 1:
                hidden = poc_fre(hidden, self.grid_pe)
        return self.feature_out(hidden)

    @property
    def get_empty_ratio(self):
        return self.ratio

    def forward(self, rays_pts_emb, scales_emb=None, rotations_emb=None, opacity=None, shs_emb=None, time_feature=None,
                time_emb=None, scene_flow=None, frame_num=None, delta_scale=None):
        if time_emb is None:
            return self.forward_static(rays_pts_emb[:, :3])
        return self.forward_dynamic(rays_pts_emb, scales_emb, rotations_emb, opacity, shs_emb, time_feature, time_emb,
                                    scene_flow, frame_num, delta_scale)

    def forward_static(self, rays_pts_emb):
        return rays_pts_emb[:, :3] + self.static_mlp(self.grid(rays_pts_emb[:, :3]))

    def _fusable(self):
        a = self.args
        return (self.W == 64 and self.D == 0 and self.grid.feat_dim == 64 and self.grid_pe == 0 and not a.no_grid
                and not a.static_mlp and not a.no_dx and not a.no_ds and not a.no_dr and a.no_do and a.no_dshs
                and not a.apply_rotation)

    def _fused_params(self):
        """The 14 tensors of the trunk and the three heads, in the kernels' order.  Cached while the first and the last are the
        objects they were (walking four nn.Sequential containers costs 33 us per call, and a training step asks once)."""
        c = self.__dict__.get("_fused_params_cache")
        if c is not None and c[0] is self.feature_out[0].weight and c[-1] is self.rotations_deform[3].bias:
            return list(c)
        ps = [self.feature_out[0].weight, self.feature_out[0].bias]
        for head in (self.pos_deform, self.scales_deform, self.rotations_deform):
            ps += [head[1].weight, head[1].bias, head[3].weight, head[3].bias]
        self.__dict__["_fused_params_cache"] = tuple(ps)
        return ps

    def forward_dynamic(self, rays_pts_emb, scales_emb, rotations_emb, opacity_emb, shs_emb, time_feature, time_emb,
                        scene_flow, frame_num, delta_scale):
        if self._fusable():
            # shipped configuration: HexPlane lookup + trunk + three heads + residuals as two fused HIP ops
            from .. import ops
            t = time_emb[:, :1] if torch.is_tensor(time_emb) else time_emb
            feat = self.grid(rays_pts_emb[:, :3], t)
            coef = delta_scale * frame_num
            coef = float(coef) if not torch.is_tensor(coef) else float(coef.item())
            pts, scales, rotations = ops.BACKEND.deform_mlp(feat, rays_pts_emb[:, :3], scales_emb[:, :3], rotations_emb[:, :4],
                                                            scene_flow, coef, self._fused_params())
            return pts, scales, rotations, opacity_emb[:, :1], shs_emb
        hidden = self.query_time(rays_pts_emb, scales_emb, rotations_emb, time_feature, time_emb)
        a = self.args
        mask = self.static_mlp(hidden) if a.static_mlp else None   # default: mask == 1 (deformation.py:103)

        def masked(x):
            return x if mask is None else x * mask

        xyz = rays_pts_emb[:, :3]
        if a.no_dx:
            pts = xyz
        else:
            # residual on top of the scene-flow motion prior (deformation.py:113-116)
            dx = self.pos_deform(hidden) + delta_scale * (frame_num * scene_flow)
            pts = masked(xyz) + dx
        scales = scales_emb[:, :3] if a.no_ds else masked(scales_emb[:, :3]) + self.scales_deform(hidden)
        if a.no_dr:
            rotations = rotations_emb[:, :4]
        elif a.apply_rotation:
            from ..utils.graphics_utils import batch_quaternion_multiply
            rotations = batch_quaternion_multiply(rotations_emb, self.rotations_deform(hidden))
        else:
            rotations = rotations_emb[:, :4] + self.rotations_deform(hidden)
        opacity = opacity_emb[:, :1] if a.no_do else masked(opacity_emb[:, :1]) + self.opacity_deform(hidden)
        if a.no_dshs:
            shs = shs_emb
        else:
            dshs = self.shs_deform(hidden).reshape([shs_emb.shape[0], 16, 3])
            shs = (shs_emb if mask is None else shs_emb * mask.unsqueeze(-1)) + dshs
        return pts, scales, rotations, opacity, shs

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" in n]


class deform_network(nn.Module):
    def __init__(self, args):
        super().__init__()
        times_ch = 2 * args.timebase_pe + 1
        self.timenet = nn.Sequential(nn.Linear(times_ch, args.timenet_width), nn.ReLU(),
                                     nn.Linear(args.timenet_width, args.timenet_output))
        self.deformation_net = Deformation(W=args.net_width, D=args.defor_depth, input_ch=3 + 3 * args.posebase_pe * 2,
                                           grid_pe=args.grid_pe, input_ch_time=args.timenet_output, args=args)
        for name, n in (("time_poc", args.timebase_pe), ("pos_poc", args.posebase_pe),
                        ("rotation_scaling_poc", args.scale_rotation_pe), ("opacity_poc", args.opacity_pe)):
            self.register_buffer(name, torch.FloatTensor([2 ** i for i in range(n)]))
        self.apply(initialize_weights)

    def forward(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None, scene_flow=None,
                frame_num=None, delta_scale=None):
        return self.forward_dynamic(point, scales, rotations, opacity, shs, times_sel, scene_flow, frame_num, delta_scale)

    @property
    def get_aabb(self):
        return self.deformation_net.get_aabb

    @property
    def get_empty_ratio(self):
        return self.deformation_net.get_empty_ratio

    def forward_static(self, points):
        return self.deformation_net(points)

    def forward_dynamic(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None,
                        scene_flow=None, frame_num=None, delta_scale=None):
        # The reference encodes point/scales/rotations with poc_fre and then reads back only their first 3/4
        # columns, i.e. the inputs themselves (deformation.py:205-207,103-135).
        return self.deformation_net(point, scales, rotations, opacity, shs, None, times_sel, scene_flow, frame_num,
                                    delta_scale)

    def get_mlp_parameters(self):
        return self.deformation_net.get_mlp_parameters() + list(self.timenet.parameters())

    def get_grid_parameters(self):
        return self.deformation_net.get_grid_parameters()


def initialize_weights(m):
    # deformation.py:229-235: xavier on the weight, applied twice when a bias exists; biases keep nn.Linear's init
    if isinstance(m, nn.Linear):
        init.xavier_uniform_(m.weight, gain=1)
        if m.bias is not None:
            init.xavier_uniform_(m.weight, gain=1)
