"""A second, independent statement of the reference rasterizer's forward -- TEST INFRASTRUCTURE ONLY.

SURVEY.md section 8(c)(3) asks for the C restatement (oracle/raster_oracle.c) to be pinned by "autograd through a slow
pure-torch re-expression of the forward".  This module is that re-expression.  It shares no code and no structure with
raster_oracle.c: where the C file walks Gaussians, tiles, sorted lists and pixels in loops like the CUDA kernels do, this one
states the same image as dense tensor algebra --

    projection and EWA covariance in matrix form (Sigma = R S^2 R^T,  cov2D = J V Sigma V^T J^T + 0.3 I),
    ONE global stable sort by depth (a tile's list is the sub-sequence of it whose rectangles cover the tile),
    a [pixels, Gaussians] alpha matrix and an exclusive cumulative product for the transmittance,

-- and gets its gradients from torch.autograd instead of from hand-written derivative code.  What it follows in the reference
(submodules/depth-diff-gaussian-rasterization/cuda_rasterizer/):
    forward.cu:20-70    computeColorFromSH (real SH basis to degree 3, +0.5, clamp at 0)
    forward.cu:73-114   computeCov2D (frustum clamp at 1.3 tan(fov), low-pass 0.3)
    forward.cu:119-152  computeCov3D (quaternion used as given, not normalised)
    forward.cu:156-256  preprocessCUDA (near cull z <= 0.2, det == 0, radius = ceil(3 sqrt(lambda_max)), tile rectangle)
    auxiliary.h:38-58   ndc2Pix, getRect (truncating casts, 16-pixel tiles)
    rasterizer_impl.cu:70-111,301-318  key = tile << 32 | float32 depth bits, stable sort: order by depth, ties by index
    forward.cu:261-379  renderCUDA (power > 0 skipped, alpha = min(0.99, o exp(power)), alpha < 1/255 skipped, stop BEFORE
                        the splat that would take T below 1e-4, colour + T bg, depth, final_T, n_contrib)

Only tests/ may import this module (oracle/ is the checker, never the product).
"""
import math
import struct

import torch

TILE = 16


def f32(x):
    """The value of the C literal `<x>f`: the reference writes its constants as float literals (0.3f, 1.3f, 0.99f, 0.0001f,
    0.0000001f, the SH table), and an fp64 evaluation that wants to agree with it to 1e-10 must use those values, not the
    decimal ones (0.3f = 0.30000001192...)."""
    return struct.unpack("f", struct.pack("f", x))[0]


# real spherical-harmonics constants in closed form (auxiliary.h:22-41 holds the same numbers as decimals)
_SPI = math.sqrt(math.pi)
C0 = f32(1.0 / (2.0 * _SPI))
C1 = f32(math.sqrt(3.0) / (2.0 * _SPI))
C2 = tuple(f32(v) for v in (math.sqrt(15.0) / (2.0 * _SPI), -math.sqrt(15.0) / (2.0 * _SPI), math.sqrt(5.0) / (4.0 * _SPI),
      -math.sqrt(15.0) / (2.0 * _SPI), math.sqrt(15.0) / (4.0 * _SPI)))
C3 = tuple(f32(v) for v in (-math.sqrt(35.0 / (2.0 * math.pi)) / 4.0, math.sqrt(105.0 / math.pi) / 2.0, -math.sqrt(21.0 / (2.0 * math.pi)) / 4.0,
      math.sqrt(7.0 / math.pi) / 4.0, -math.sqrt(21.0 / (2.0 * math.pi)) / 4.0, math.sqrt(105.0 / math.pi) / 4.0,
      -math.sqrt(35.0 / (2.0 * math.pi)) / 4.0))


def sh_colour(deg, sh, d):
    """sh [P, M, 3], unit directions d [P, 3] -> [P, 3] before the +0.5 and the clamp."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    c = C0 * sh[:, 0]
    if deg > 0:
        c = c - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        c = (c + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6] + C2[3] * xz * sh[:, 7]
             + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        c = (c + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
             + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
             + C3[5] * z * (xx - yy) * sh[:, 14] + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return c


def quaternion_matrix(q):
    """Rotation matrix of (r, x, y, z) as given -- the reference does not normalise (forward.cu:128)."""
    r, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)


def render(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, bg, shs=None, sh_degree=0,
           colors_precomp=None, scales=None, rotations=None, scale_modifier=1.0, means2D=None):
    """All arguments torch tensors of one floating dtype (float64 for the autograd comparisons); viewmatrix / projmatrix are the
    transposed 4x4s the rasterizer takes (a point is a ROW vector times them).  `means2D` [P, 3] is the reference's
    screen-space gradient holder: it enters as an offset in NDC, so its gradient is dL/d(ndc) = dL/d(pixel) * (W/2, H/2), which
    is what RasterizeGaussiansBackwardCUDA returns as dL_dmeans2D (backward.cu:573-579).
    Returns a dict: color [3,H,W], depth [1,H,W], radii [P], final_T [H*W], n_contrib [H*W], tiles_touched [P], num_rendered."""
    P = means3D.shape[0]
    dt, dev = means3D.dtype, means3D.device
    view, proj = viewmatrix.reshape(4, 4), projmatrix.reshape(4, 4)
    ones = torch.ones(P, 1, dtype=dt, device=dev)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ view[:, :3]
    depth = p_view[:, 2]
    p_hom = hom @ proj
    p_w = 1.0 / (p_hom[:, 3] + f32(0.0000001))
    ndc = p_hom[:, :2] * p_w[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    size = torch.tensor([W, H], dtype=dt, device=dev)
    pix = ((ndc + 1.0) * size - 1.0) * 0.5                                        # ndc2Pix

    # world covariance and its projection
    Rq = quaternion_matrix(rotations)
    S = scale_modifier * scales
    Sigma = Rq @ torch.diag_embed(S * S) @ Rq.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = f32(1.3) * tanfovx, f32(1.3) * tanfovy
    tx = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(P, 2, 3)
    V = view[:3, :3].t()                                                          # rotation of the world-to-view map, column form
    M = J @ V
    cov = M @ Sigma @ M.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + f32(0.3), cov[:, 0, 1], cov[:, 1, 1] + f32(0.3)
    det = a * c - b * b
    safe = torch.where(det == 0, torch.ones_like(det), det)
    conic = torch.stack([c / safe, -b / safe, a / safe], 1)

    with torch.no_grad():
        mid = 0.5 * (a + c)
        root = torch.sqrt(torch.clamp(mid * mid - det, min=f32(0.1)))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + root, mid - root)))
        gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        x0 = torch.clamp(torch.trunc((pix[:, 0] - radius) / TILE), 0, gx)
        x1 = torch.clamp(torch.trunc((pix[:, 0] + radius + TILE - 1) / TILE), 0, gx)
        y0 = torch.clamp(torch.trunc((pix[:, 1] - radius) / TILE), 0, gy)
        y1 = torch.clamp(torch.trunc((pix[:, 1] + radius + TILE - 1) / TILE), 0, gy)
        tiles = ((x1 - x0) * (y1 - y0)).to(torch.int64)
        live = (depth > f32(0.2)) & (det != 0) & (tiles > 0)
        tiles = torch.where(live, tiles, torch.zeros_like(tiles))
        radii = torch.where(live, radius, torch.zeros_like(radius)).to(torch.int32)
        # the global order every tile list is a sub-sequence of: float32 depth bits (positive floats order like their bits),
        # ties by index (the sort is stable and the keys are emitted in index order)
        order = torch.sort(depth.to(torch.float32), stable=True).indices
        # membership: tile (tx, ty) x Gaussian
        txs = torch.arange(gx, device=dev, dtype=dt)[None, :, None]
        tys = torch.arange(gy, device=dev, dtype=dt)[:, None, None]
        member = (live[None, None] & (txs >= x0) & (txs < x1) & (tys >= y0) & (tys < y1))[:, :, order]       # [gy, gx, P] in depth order
        position = torch.cumsum(member.to(torch.int64), 2)                          # 1-based position in the tile's list

    if colors_precomp is None:
        d = means3D - campos.reshape(1, 3)
        d = d / d.norm(dim=1, keepdim=True)
        colour = torch.clamp_min(sh_colour(sh_degree, shs, d) + 0.5, 0.0)
    else:
        colour = colors_precomp

    # dense compositing in depth order
    o = order
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    pxf, pyf = xs.reshape(-1).to(dt), ys.reshape(-1).to(dt)
    in_list = member[ys.reshape(-1) // TILE, xs.reshape(-1) // TILE]              # [N, P]
    pos = position[ys.reshape(-1) // TILE, xs.reshape(-1) // TILE]
    dx = pix[o, 0][None, :] - pxf[:, None]
    dy = pix[o, 1][None, :] - pyf[:, None]
    power = -0.5 * (conic[o, 0] * dx * dx + conic[o, 2] * dy * dy) - conic[o, 1] * dx * dy
    alpha = torch.clamp_max(opacities.reshape(-1)[o][None, :] * torch.exp(torch.clamp_max(power, 0.0)), f32(0.99))
    counted = in_list & ~(power > 0) & ~(alpha < 1.0 / 255.0)
    alpha = torch.where(counted, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - alpha
    T_before = torch.cumprod(torch.cat([torch.ones_like(one_minus[:, :1]), one_minus[:, :-1]], 1), 1)
    with torch.no_grad():
        stops = counted & (T_before * one_minus < f32(0.0001))
        running = torch.cumsum(stops.to(torch.int64), 1) == 0                      # still compositing at (and including) this entry
    contributes = counted & running
    w = torch.where(contributes, alpha * T_before, torch.zeros_like(alpha))       # [N, P]
    T_final = torch.prod(torch.where(contributes, one_minus, torch.ones_like(one_minus)), 1)
    image = w @ colour[o] + T_final[:, None] * bg.reshape(1, 3)
    depth_img = w @ depth[o]
    with torch.no_grad():
        n_contrib = torch.where(contributes, pos, torch.zeros_like(pos)).max(dim=1).values
    return {"color": image.t().reshape(3, H, W), "depth": depth_img.reshape(1, H, W), "radii": radii, "final_T": T_final,
            "n_contrib": n_contrib, "tiles_touched": tiles, "num_rendered": int(tiles.sum()), "order": order}
