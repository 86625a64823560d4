"""The stage-1 -> stage-2 data contract of the reference (scene/dataset_readers.py:33-83,802-869,994-1018,1128-1202):
`MOM/train_data.pth` (a torch pickle holding intrinsics, the point cloud and the multi-view frames as PIL images),
`MOM/video/*.png` (the animated centre view, one image per timestamp) and the four fixed render trajectories.

Only the Blender-style loader train_4DGS.py / render_4DGS.py actually call (`sceneLoadTypeCallbacks["Blender"]` =
readNerfSyntheticInfo) exists; the COLMAP / dynerf / nerfies / PanopticSports readers of the reference are unreachable from
those scripts and are not built.  Differences a caller can observe:
  * train_data.pth is read ONCE per call (the reference reads it three times);
  * the render trajectories (test_trajectory/{up-down,side,zoom-in,circle}_{R,t}_list in the reference's working directory)
    are closed forms here (trajectory()), pinned against the reference's lists by tests/golden/g9_trajectories.npz; files in
    ./test_trajectory are used instead when they exist;
  * nothing is moved to "cuda" at load time: images stay on the host, as in the reference, matrices go to the device on first
    use (Camera.device_tensors).
"""
import math
import os
from typing import NamedTuple

import numpy as np
import torch

from ..utils.graphics_utils import BasicPointCloud, getWorld2View2

FOCAL = 5.8269e+02      # dataset_readers.py:996, train_motion.py:52: the focal length every stage-1 output is built with


class CameraInfo(NamedTuple):
    uid: int
    R: np.ndarray
    T: np.ndarray
    FovY: float
    FovX: float
    image: torch.Tensor
    image_path: str
    image_name: str
    width: int
    height: int
    time: float
    mask: object
    frame_num: int


class SceneInfo(NamedTuple):
    point_cloud: BasicPointCloud
    train_cameras: list
    train_cameras_2: list
    test_cameras: list
    video_cameras_up: list
    video_cameras_side: list
    video_cameras_zoom: list
    video_cameras_circle: list
    nerf_normalization: dict
    ply_path: str
    maxtime: float


def getNerfppNorm(cam_info):
    """Centre and 1.1 x radius of the camera positions (dataset_readers.py:62-83).  The reference's dtypes are kept -- the
    world-to-view matrices are float32, so the centres, their mean, the distances and `diagonal * 1.1` are float32 arithmetic
    (under NumPy 2's promotion rules the product stays float32): the radius becomes spatial_lr_scale, and fixture g12 pins it
    to the last bit."""
    centres = np.hstack([np.linalg.inv(getWorld2View2(c.R, c.T))[:3, 3:4] for c in cam_info])
    centre = np.mean(centres, axis=1, keepdims=True)
    diagonal = np.max(np.linalg.norm(centres - centre, axis=0, keepdims=True))
    return {"translate": -centre.flatten(), "radius": diagonal * 1.1}


def frame_timeline(num_frames):
    """{timestamp (np.float32 in [0, 2]) -> normalised time in [0, 1]} (dataset_readers.py:1150-1158)."""
    line = np.linspace(0, 2, num_frames, dtype=np.float32)
    top = max(line)
    return {t: t / top for t in line}


def read_timeline(path=None):
    """The fixed 60-stamp timeline (dataset_readers.py:1128-1148): (mapper, max time)."""
    line = np.linspace(0, 2, 60, dtype=np.float32)
    top = max(line)
    return {t: t / top for t in line}, top


def trajectory(name):
    """(R [n,3,3], t [n,3]) float32 of a render path: what the reference loads from test_trajectory/<name>_{R,t}_list.
    side: x from +0.09 to -0.09; up-down: y from +0.08 to -0.08; zoom-in: z from 0 to -0.24 (60 poses each, equal steps);
    circle: 90 poses, (x, y) on a circle of radius 0.04 at 8 degrees per pose, z = 0.09 cos(4 degrees x pose).  No rotation."""
    for d in (os.path.join(os.getcwd(), "test_trajectory"),):
        fr, ft = os.path.join(d, name + "_R_list"), os.path.join(d, name + "_t_list")
        if os.path.exists(fr) and os.path.exists(ft):
            R = torch.load(fr, map_location="cpu", weights_only=False)
            t = torch.load(ft, map_location="cpu", weights_only=False)
            return np.stack([np.asarray(r, np.float32) for r in R]), np.stack([np.asarray(x, np.float32) for x in t])
    n = 90 if name == "circle" else 60
    t = np.zeros((n, 3), np.float64)
    if name == "side":
        t[:, 0] = np.linspace(0.09, -0.09, n)
    elif name == "up-down":
        t[:, 1] = np.linspace(0.08, -0.08, n)
    elif name == "zoom-in":
        t[:, 2] = np.linspace(0.0, -0.24, n)
    elif name == "circle":
        k = np.arange(n)
        t[:, 0] = -0.04 * np.cos(np.deg2rad(8.0 * k))
        t[:, 1] = -0.04 * np.sin(np.deg2rad(8.0 * k))
        t[:, 2] = 0.09 * np.cos(np.deg2rad(4.0 * k))
    else:
        raise ValueError(f"unknown trajectory {name!r}")
    return np.broadcast_to(np.eye(3, dtype=np.float32), (n, 3, 3)).copy(), t.astype(np.float32)


def _pose_from_c2w(transform_matrix):
    """OpenGL/Blender camera-to-world -> (R, T) as the rasterizer wants them: flip y and z, invert, R stored transposed
    (dataset_readers.py:1037-1045)."""
    c2w = np.array(transform_matrix, dtype=np.float64)
    c2w[:3, 1:3] *= -1
    w2c = np.linalg.inv(c2w)
    return np.transpose(w2c[:3, :3]), w2c[:3, 3]


def _image_tensor(pil_image, white_background):
    """RGBA compositing over the background, float32 CHW in [0, 1] (dataset_readers.py:1048-1056)."""
    data = np.array(pil_image.convert("RGBA")) / 255.0
    bg = np.array([1, 1, 1]) if white_background else np.array([0, 0, 0])
    arr = data[:, :, :3] * data[:, :, 3:4] + bg * (1 - data[:, :, 3:4])
    return torch.Tensor(arr).permute(2, 0, 1)


def load_train_data(TrainData_path):
    """MOM/train_data.pth: {camera_angle_x, camera_angle_y, W, H, pcd_points [3,P], pcd_colors [P,3], pcd_masks, frames: [{image:
    PIL, transform_matrix: 4x4 c2w, mask: PIL, ...}]} (train_motion.py:251-260,354-364).  A pickle with PIL images inside."""
    data = torch.load(TrainData_path, map_location="cpu", weights_only=False)
    missing = [k for k in ("camera_angle_x", "camera_angle_y", "W", "H", "pcd_points", "pcd_colors", "frames") if k not in data]
    if missing:
        raise KeyError(f"{TrainData_path}: not a stage-1 train_data.pth (missing {missing})")
    if len(data["frames"]) < 3:
        raise ValueError("train_data.pth needs at least 3 frames: the video frames take the pose of frame 2 (dataset_readers.py:821)")
    return data


def multiview_cameras(data, white_background, time):
    """The multi-view frames, all at one timestamp, frame_num 0 (readCamerasFromTransforms_MVS, dataset_readers.py:1022-1058)."""
    out, image = [], None
    for idx, frame in enumerate(data["frames"]):
        R, T = _pose_from_c2w(frame["transform_matrix"])
        image = _image_tensor(frame["image"], white_background)
        out.append(CameraInfo(uid=idx, R=R, T=T, FovY=data["camera_angle_y"], FovX=data["camera_angle_x"], image=image, image_path='',
                              image_name='', width=image.shape[2], height=image.shape[1], time=time, mask=None, frame_num=0))
    return out, image


def video_cameras(data, TrainData_path, white_background):
    """The animated centre view: one camera per image of MOM/video (sorted by name) at the pose of frame 2, time = its place on
    the timeline, frame_num = its index; followed by the multi-view frames at time 0 (readCamerasFromTransforms_Wframe,
    dataset_readers.py:802-869).  Returns (cameras, time_line, mapper)."""
    from PIL import Image
    folder = os.path.join(os.path.dirname(TrainData_path), 'video')
    names = sorted(n for n in os.listdir(folder) if n.endswith((".jpg", ".jpeg", ".png")))
    mapper = frame_timeline(len(names))
    time_line = np.linspace(0, 2, len(names), dtype=np.float32)
    R, T = _pose_from_c2w(data["frames"][2]["transform_matrix"])
    out = []
    for idx, name in enumerate(names):
        image = _image_tensor(Image.open(os.path.join(folder, name)), white_background)
        out.append(CameraInfo(uid=idx, R=R, T=T, FovY=data["camera_angle_y"], FovX=data["camera_angle_x"], image=image, image_path='',
                              image_name='', width=image.shape[2], height=image.shape[1], time=mapper[time_line[idx]], mask=None,
                              frame_num=idx))
    views, _ = multiview_cameras(data, white_background, mapper[time_line[0]])
    return out + views, time_line, mapper


def trajectory_cameras(name, sample_image, time_line, mapper, width, height):
    """Cameras of one render path (generateCamerasFromTransforms_one_path, dataset_readers.py:990-1018): intrinsics from the
    fixed focal length, pose idx at the time of video frame idx with frame_num = idx; at most 60, and the last pose of the list
    is never used."""
    R_list, T_list = trajectory(name)
    fx, fy = FOCAL * (width / height), FOCAL
    fovx, fovy = 2 * np.arctan(width / (2 * fx)), 2 * np.arctan(height / (2 * fy))
    out = []
    for idx in range(len(R_list)):
        if idx >= 60 or idx == len(R_list) - 1:
            break
        out.append(CameraInfo(uid=idx, R=R_list[idx].copy(), T=T_list[idx].copy(), FovY=fovy, FovX=fovx, image=sample_image,
                              image_path=None, image_name=None, width=width, height=height, time=mapper[time_line[idx]], mask=None,
                              frame_num=idx))
    return out


def readNerfSyntheticInfo(TrainData_path, slr_path=None, path=None, white_background=False, eval=False, viewcrafter=False,
                          extension=".png"):
    """dataset_readers.py:1160-1202: SceneInfo + the video's timeline from a stage-1 output directory."""
    mapper60, max_time = read_timeline(path)
    data = load_train_data(TrainData_path)
    print("Reading Training Transforms")
    train, sample_image = multiview_cameras(data, white_background, mapper60[np.float32(0)])
    print("Stage 1 data: ", len(train))
    train2, time_line, mapper = video_cameras(data, TrainData_path, white_background)
    print("Stage 2 data: ", len(train2))
    W, H = data['W'], data['H']
    paths = {k: trajectory_cameras(n, sample_image, time_line, mapper, W, H)
             for k, n in (("up", "up-down"), ("side", "side"), ("zoom", "zoom-in"), ("circle", "circle"))}
    pts = data['pcd_points']
    pts = pts.numpy() if torch.is_tensor(pts) else np.asarray(pts)
    cols = data['pcd_colors']
    cols = cols.numpy() if torch.is_tensor(cols) else np.asarray(cols)
    pcd = BasicPointCloud(points=pts.T, colors=cols, normals=None)
    info = SceneInfo(point_cloud=pcd, train_cameras=train, train_cameras_2=train2, test_cameras=train,
                     video_cameras_up=paths["up"], video_cameras_side=paths["side"], video_cameras_zoom=paths["zoom"],
                     video_cameras_circle=paths["circle"], nerf_normalization=getNerfppNorm(train), ply_path=None, maxtime=max_time)
    return info, time_line


def add_points(pointsclouds, xyz_min, xyz_max):
    raise NotImplementedError("args.add_points is False in every shipped configuration (arguments/__init__.py:65)")


sceneLoadTypeCallbacks = {"Blender": readNerfSyntheticInfo}
