"""render(): one camera, one frame (reference gaussian_renderer/__init__.py:22-178) -- same signature, same
returned dict, same maths; the deformation field and the rasterizer underneath are libmom4d HIP kernels, the
camera matrices are staged on the device once per camera instead of every call, and the timestamp travels as
a scalar instead of a [P,1] tensor."""
import math

import torch

from .. import ops
from ..diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from ..scene.gaussian_model import GaussianModel  # noqa: F401  (render_4DGS.py:22 imports it from here)
from ..utils.sh_utils import eval_sh
from . import network_gui  # noqa: F401  (train_4DGS.py:17)


_RENDER_STREAMS = 1


def set_render_streams(n):
    """n > 1: no-grad render() deals consecutive frames to n alternating streams (fused_render.FusedRenderPool: +42 % frames/s
    with two, +54 % with three at config 2).  The returned dict then carries "stream" and "ready"; the caller's stream is NOT made
    to wait for the frame -- see FusedRenderPool for how to consume it.  n = 1 (default): the reference's behaviour, every frame on
    the current stream."""
    global _RENDER_STREAMS
    _RENDER_STREAMS = max(1, int(n))


def render_streams():
    return _RENDER_STREAMS


def _nograd_fast_path_applies(cam, pc, pipe, stage, override_color, cam_type):
    """Forward-only launch sequence (fused_render.py): no gradients wanted, fine stage, the shipped deformation configuration,
    SH colours and covariances computed by the rasterizer, an ordinary camera, everything on the GPU."""
    if torch.is_grad_enabled() or stage != "fine" or override_color is not None or cam_type == "PanopticSports":
        return False
    if pipe.compute_cov3D_python or pipe.convert_SHs_python or not hasattr(cam, "device_tensors"):
        return False
    if not pc.get_xyz.is_cuda or pc.get_xyz.shape[0] == 0 or ops.BACKEND.name != "hip":
        return False
    dn = getattr(pc._deformation, "deformation_net", None)
    return dn is not None and hasattr(dn, "_fusable") and dn._fusable() and pc._features_rest.shape[1] == 15


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, stage="fine",
           cam_type=None, delta_scale=None):
    """Background tensor (bg_color) must be on the GPU."""
    means3D = pc.get_xyz
    dev = means3D.device
    if stage != "coarse" and delta_scale is None:
        # the reference forms `delta_scale * (frame_num * scene_flow)` with whatever it was given (scene/deformation.py:114 via
        # gaussian_renderer/__init__.py:101-103): None raises there, so it raises here, on every path
        raise TypeError("unsupported operand type(s) for *: 'NoneType' and 'Tensor' (render(): delta_scale is required "
                        "outside the coarse stage)")
    if _RENDER_STREAMS > 1 and _nograd_fast_path_applies(viewpoint_camera, pc, pipe, stage, override_color, cam_type):
        pool = getattr(pc, "_fused_render_pool", None)
        if pool is None or pool.n != _RENDER_STREAMS:
            from ..fused_render import FusedRenderPool
            pool = pc._fused_render_pool = FusedRenderPool(pc, _RENDER_STREAMS)
        image, depth, radii, visible, stream, ready = pool.render(viewpoint_camera, bg_color, delta_scale, scaling_modifier, pipe.debug)
        return {"render": image, "viewspace_points": pool.zero_points(), "visibility_filter": visible, "radii": radii,
                "depth": depth, "flow_loss": 0, "stream": stream, "ready": ready}
    if _nograd_fast_path_applies(viewpoint_camera, pc, pipe, stage, override_color, cam_type):
        fr = getattr(pc, "_fused_render", None)
        if fr is None:
            from ..fused_render import FusedRender
            fr = pc._fused_render = FusedRender(pc)
        image, depth, radii = fr.render(viewpoint_camera, bg_color, delta_scale, scaling_modifier,
                                        pipe.debug)
        return {"render": image, "viewspace_points": torch.zeros_like(means3D), "visibility_filter": radii > 0, "radii": radii,
                "depth": depth, "flow_loss": 0}
    # gradient holder for the 2D means (read back by the densification statistics, train_4DGS.py:227-229)
    screenspace_points = torch.zeros_like(means3D, dtype=means3D.dtype, requires_grad=True, device=dev) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    from .. import fused_autograd
    if fused_autograd.applies(viewpoint_camera, pc, pipe, stage, override_color, cam_type):
        # the whole fine-stage forward of this camera as ONE autograd node (fused_autograd.py); pipe.per_op_autograd = True keeps
        # the op-by-op path below
        image, depth, radii = fused_autograd.render(viewpoint_camera, pc, pipe, bg_color, delta_scale, scaling_modifier,
                                                    screenspace_points)
        return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
                "depth": depth, "flow_loss": 0}

    if cam_type != "PanopticSports":
        if hasattr(viewpoint_camera, "device_tensors"):
            view, proj, campos, _ = viewpoint_camera.device_tensors(dev)
        else:
            view, proj, campos = (viewpoint_camera.world_view_transform.to(dev), viewpoint_camera.full_proj_transform.to(dev),
                                  viewpoint_camera.camera_center.to(dev))
        raster_settings = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
            scale_modifier=scaling_modifier, viewmatrix=view, projmatrix=proj, sh_degree=pc.active_sh_degree, campos=campos,
            prefiltered=False, debug=pipe.debug)
        time = float(viewpoint_camera.time)
    else:
        raster_settings = viewpoint_camera['camera']
        time = float(viewpoint_camera['time'])
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means2D = screenspace_points
    opacity = pc._opacity
    shs = pc.get_features
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc._scaling, pc._rotation

    if stage == "coarse":
        means3D_final, scales_final, rotations_final, opacity_final, shs_final = means3D, scales, rotations, opacity, shs
    else:
        # fine stage: HexPlane + MLP residual on top of frame_num * scene_flow (:101-103)
        means3D_final, scales_final, rotations_final, opacity_final, shs_final = pc._deformation(
            means3D, scales, rotations, opacity, shs, time, pc.get_flow, viewpoint_camera.frame_num, delta_scale)
    flow_loss = 0

    scales_final = pc.scaling_activation(scales_final)
    rotations_final = pc.rotation_activation(rotations_final)
    opacity = pc.opacity_activation(opacity_final)

    colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - campos.repeat(pc.get_features.shape[0], 1)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp / dir_pp.norm(dim=1, keepdim=True))
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
            shs_final = None
    else:
        colors_precomp = override_color
        shs_final = None

    rendered_image, radii, depth = rasterizer(means3D=means3D_final, means2D=means2D, shs=shs_final,
                                              colors_precomp=colors_precomp, opacities=opacity, scales=scales_final,
                                              rotations=rotations_final, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth, "flow_loss": flow_loss}
