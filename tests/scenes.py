"""Seeded synthetic inputs shared by the parity tests (numpy only, no torch needed)."""
import math
import numpy as np


def projection_matrix(znear, zfar, fovx, fovy):
    """Row-major P of utils/graphics_utils.py:51-71 (reference), as float32."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera(W, H, R=None, T=None, focal=None):
    """Returns dict(viewmatrix, projmatrix (both transposed, as the rasterizer wants), campos, tanfovx, tanfovy)."""
    R = np.eye(3) if R is None else np.asarray(R, np.float64)
    T = np.zeros(3) if T is None else np.asarray(T, np.float64)
    focal = 582.69 if focal is None else focal  # train_motion.py:52-56 style intrinsics
    fovx = 2 * math.atan(W / (2 * focal * W / H)) if False else 2 * math.atan(W / (2 * focal))
    fovy = 2 * math.atan(H / (2 * focal))
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.T
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    w2c = np.float32(Rt)
    view = w2c.T.copy()
    proj = projection_matrix(0.01, 100.0, fovx, fovy).T.copy()
    full = (view @ proj).astype(np.float32)
    campos = np.linalg.inv(view.astype(np.float64))[3, :3].astype(np.float32)
    return dict(viewmatrix=view, projmatrix=full, campos=campos, tanfovx=math.tan(fovx * 0.5),
                tanfovy=math.tan(fovy * 0.5), W=W, H=H)


def random_gaussians(P, seed=0, W=128, H=96, zrange=(1.0, 6.0), scale=(-4.5, -2.0), sh_coeffs=16, focal=None):
    """Random Gaussians spread over (and slightly beyond) the frustum of camera(W,H)."""
    rng = np.random.default_rng(seed)
    cam = camera(W, H, focal=focal)
    z = rng.uniform(*zrange, P)
    # a few behind / at the near plane to exercise the cull (auxiliary.h:154)
    n_near = max(1, P // 50)
    z[:n_near] = rng.uniform(-1.0, 0.25, n_near)
    x = rng.uniform(-1.3, 1.3, P) * cam["tanfovx"] * z
    y = rng.uniform(-1.3, 1.3, P) * cam["tanfovy"] * z
    means = np.stack([x, y, z], 1).astype(np.float32)
    scales = np.exp(rng.uniform(scale[0], scale[1], (P, 3))).astype(np.float32)
    rots = rng.normal(size=(P, 4)).astype(np.float32)
    rots /= np.linalg.norm(rots, axis=1, keepdims=True)
    opac = (1 / (1 + np.exp(-rng.normal(0, 2, (P, 1))))).astype(np.float32)
    shs = (rng.normal(0, 0.3, (P, sh_coeffs, 3))).astype(np.float32)
    shs[:, 0, :] += rng.uniform(0, 2.0, (P, 3)).astype(np.float32)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    return dict(means3D=means, scales=scales, rotations=rots, opacities=opac, shs=shs, bg=bg, **cam)
