"""Synthetic scene S(P, F, W, H, seed) of SURVEY.md section 8(d): the stand-in for a stage-1 output
(MOM/train_data.pth + MOM/scene_flow.pth + MOM/video/*.png), since neither datasets nor the stage-1 checkpoints
are available.  It reproduces the reference's data contract: intrinsics as train_motion.py:52-56 /
dataset_readers.py:994-1002 derive them, a point cloud unprojected from a depth map (train_motion.py:221-222),
F video frames from the centre view (time = f/(F-1), frame_num = f; dataset_readers.py:802-869) and a few
multi-view frames at time 0."""
import math

import numpy as np
import torch

from ..utils.graphics_utils import BasicPointCloud
from .cameras import Camera

FOCAL = 582.69


def _box_blur(img, k=9):
    pad = k // 2
    x = torch.nn.functional.pad(img[None], (pad, pad, pad, pad), mode="replicate")
    w = torch.ones(3, 1, k, k) / (k * k)
    return torch.nn.functional.conv2d(x, w, groups=3)[0]


class SyntheticScene:
    dataset_type = "blender"

    def __init__(self, P, F, W, H, seed=6666, n_views=5, model_path="", image_device="cpu"):
        g = torch.Generator("cpu").manual_seed(seed)
        self.P, self.F, self.W, self.H, self.model_path = P, F, W, H, model_path
        fy = FOCAL
        fx = FOCAL * W / H
        self.FovX, self.FovY = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
        # ---- point cloud on a sub-grid of the image, depth = 3 + sum of 4 sinusoids
        hs = max(1, int(round(math.sqrt(P * H / W))))
        ws = (P + hs - 1) // hs
        uu = (torch.arange(ws, dtype=torch.float64) + 0.5) * (W / ws)
        vv = (torch.arange(hs, dtype=torch.float64) + 0.5) * (H / hs)
        v, u = torch.meshgrid(vv, uu, indexing="ij")
        u, v = u.reshape(-1)[:P], v.reshape(-1)[:P]
        amp = torch.rand(4, generator=g, dtype=torch.float64) * 0.4 + 0.1
        mag = (torch.rand(4, generator=g, dtype=torch.float64) * 6 * math.pi + 2 * math.pi) / W
        ang = torch.rand(4, generator=g, dtype=torch.float64) * 2 * math.pi
        phi = torch.rand(4, generator=g, dtype=torch.float64) * 2 * math.pi
        d = torch.full_like(u, 3.0)
        for k in range(4):
            d = d + amp[k] * torch.sin(mag[k] * (math.cos(ang[k]) * u + math.sin(ang[k]) * v) + phi[k])
        x = (u - W / 2) / fx * d
        y = (v - H / 2) / fy * d
        pts = torch.stack([x, y, d], 1).float()
        cols = torch.rand(P, 3, generator=g)
        self.point_cloud = BasicPointCloud(points=pts.numpy(), colors=cols.numpy(), normals=np.zeros((P, 3), np.float32))
        self.scene_flow = torch.randn(3, P, generator=g) * 1e-3     # MOM/scene_flow.pth layout [3,P]
        self.xyz_max, self.xyz_min = pts.max(0).values.tolist(), pts.min(0).values.tolist()
        # ---- ground truth frames: blurred noise (content does not matter for timing; the loss stays finite)
        n_img = min(F + n_views, 8)
        bank = [_box_blur(torch.rand(3, H, W, generator=g)).clamp(0, 1).to(image_device) for _ in range(n_img)]
        # ---- cameras
        eye = np.eye(3)
        self._video = [Camera(colmap_id=f, R=eye, T=np.zeros(3), FoVx=self.FovX, FoVy=self.FovY, image=bank[f % n_img],
                              gt_alpha_mask=None, image_name=f"{f}", uid=f, data_device=image_device,
                              time=(f / (F - 1) if F > 1 else 0.0), frame_num=f) for f in range(F)]
        self._views = []
        ctr = np.array([0.0, 0.0, 3.0])
        for i, (yaw, pitch) in enumerate([(0, 0), (5, 0), (-5, 0), (0, 5), (0, -5)][:n_views]):
            cy, sy, cp, sp = math.cos(math.radians(yaw)), math.sin(math.radians(yaw)), math.cos(math.radians(pitch)), math.sin(math.radians(pitch))
            Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
            R_c2w = Ry @ Rx
            cam_pos = ctr - R_c2w @ ctr          # orbit the scene centre
            T = -R_c2w.T @ cam_pos               # w2c translation
            self._views.append(Camera(colmap_id=F + i, R=R_c2w, T=T, FoVx=self.FovX, FoVy=self.FovY,
                                      image=bank[(F + i) % n_img], gt_alpha_mask=None, image_name=f"v{i}", uid=F + i,
                                      data_device=image_device, time=0.0, frame_num=0))
        centres = np.stack([c.camera_center.numpy() for c in self._views + self._video[:1]])
        self.cameras_extent = float(1.1 * np.linalg.norm(centres - centres.mean(0, keepdims=True), axis=1).max())
        if self.cameras_extent == 0:
            self.cameras_extent = 1.0

    # the accessors train_4DGS.py / render_4DGS.py use (scene/__init__.py:96-114)
    def getTrainCameras(self):
        return self._views

    def getTrainCameras_2(self):
        return self._video + self._views

    def getTestCameras(self):
        return self._views

    def getVideoCameras_up(self):
        return self._video

    getVideoCameras_zoom = getVideoCameras_circle = getVideoCameras_up

    @staticmethod
    def side_trajectory():
        """(R[60,3,3], t[60,3]) of the reference's `side` render path (test_trajectory/side_{R,t}_list): no rotation,
        the camera slides along x from +0.09 to -0.09 in 60 equal steps.  tests/golden/g9_side_trajectory.npz holds
        the reference's own lists; tests/test_golden_cpu.py checks this restatement against them."""
        R = np.broadcast_to(np.eye(3, dtype=np.float32), (60, 3, 3)).copy()
        t = np.zeros((60, 3), dtype=np.float32)
        t[:, 0] = np.linspace(0.09, -0.09, 60, dtype=np.float64).astype(np.float32)
        return R, t

    def getVideoCameras_side(self):
        """The 59 cameras render_4DGS.py renders for the "side" video (scene/dataset_readers.py:1003-1018: pose idx,
        time of video frame idx, frame_num = idx, and the last of the 60 poses is dropped)."""
        if getattr(self, "_side", None) is None:
            R, t = self.side_trajectory()
            F = len(self._video)
            self._side = [Camera(colmap_id=i, R=R[i].astype(np.float64), T=t[i].astype(np.float64), FoVx=self.FovX, FoVy=self.FovY,
                                 image=self._video[i % F]._image_host, gt_alpha_mask=None, image_name=f"side{i}", uid=i,
                                 data_device=self._video[0].data_device, time=((i % F) / (F - 1) if F > 1 else 0.0),
                                 frame_num=i % F) for i in range(59)]
        return self._side

    def init_gaussians(self, gaussians, flow_scale=2):
        """What Scene.__init__ does with a fresh model (scene/__init__.py:78-89)."""
        gaussians._deformation.deformation_net.set_aabb(self.xyz_max, self.xyz_min)
        gaussians.create_from_pcd(self.point_cloud, self.cameras_extent, self.F, "", flow_scale, scene_flow=self.scene_flow)
        return gaussians

    def make_trained_like(self, gaussians, seed=6666):
        """The benchmark state of SURVEY 8(d): perturb the initial model so that it looks like a partly trained one."""
        g = torch.Generator("cpu").manual_seed(seed)
        dev = gaussians._xyz.device
        with torch.no_grad():
            gaussians._scaling += (torch.randn(gaussians._scaling.shape, generator=g) * 0.5).to(dev)
            gaussians._rotation.copy_(torch.randn(gaussians._rotation.shape, generator=g).to(dev))
            gaussians._opacity.copy_((torch.randn(gaussians._opacity.shape, generator=g) * 2).to(dev))
            gaussians._features_rest.copy_((torch.randn(gaussians._features_rest.shape, generator=g) * 0.1).to(dev))
        gaussians.active_sh_degree = gaussians.max_sh_degree
        return gaussians
