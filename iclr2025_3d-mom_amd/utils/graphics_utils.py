"""Mirror of the reference's utils/graphics_utils.py:17-75 (camera matrices)."""
import math
from typing import NamedTuple

import numpy as np
import torch


class BasicPointCloud(NamedTuple):
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


def getWorld2View2(R, t, translate=np.array([.0, .0, .0]), scale=1.0):
    """World-to-camera 4x4 (float32) after recentring/scaling the camera position (graphics_utils.py:38-49).
    R is the camera-to-world rotation (stored transposed, as in 3DGS), t the w2c translation."""
    w2c = np.eye(4)
    w2c[:3, :3] = np.asarray(R).T
    w2c[:3, 3] = np.asarray(t)
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(c2w))


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """graphics_utils.py:51-71: z maps to [0,1], w = z."""
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    right, top = tx * znear, ty * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - -right)
    P[1, 1] = 2.0 * znear / (top - -top)
    P[0, 2] = (right + -right) / (right - -right)
    P[1, 2] = (top + -top) / (top - -top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def batch_quaternion_multiply(q1, q2):
    """Hamilton product of [N,4] (w,x,y,z) batches, normalised (graphics_utils.py:107-132)."""
    w1, x1, y1, z1 = q1.unbind(1)
    w2, x2, y2, z2 = q2.unbind(1)
    q = torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2), dim=1)
    return q / torch.norm(q, dim=1, keepdim=True)
