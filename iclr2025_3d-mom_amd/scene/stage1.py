"""Writes a stage-1 output directory in the reference's layout (train_motion.py:251-260,354-364,463-464):

    <dir>/MOM/train_data.pth    torch pickle: intrinsics, point cloud [3,P], colours [P,3], masks, frames (PIL image, 4x4 OpenGL
                                camera-to-world, PIL mask, the hint / flow lists)
    <dir>/MOM/video/00000.png   the animated centre view, one PNG per timestamp
    <dir>/MOM/scene_flow.pth    [3,P] Eulerian flow (gaussian_model.py:183)

from a SyntheticScene, so that the whole stage-2 path -- Scene(TrainData_path=...), train, render -- can be driven from files
exactly as train_4DGS.py / render_4DGS.py do, without the stage-1 networks (ZoeDepth, the motion estimator) that are out of
scope.  Also the validator the reference lacks: check_stage1_dir()."""
import os

import numpy as np
import torch


def _to_pil(img_chw):
    from PIL import Image
    return Image.fromarray(np.round(img_chw.permute(1, 2, 0).clamp(0, 1).numpy() * 255.0).astype(np.uint8))


def _c2w_opengl(cam):
    """Inverse of dataset_readers.py:1037-1045: (R stored transposed, T) -> OpenGL camera-to-world."""
    w2c = np.eye(4)
    w2c[:3, :3] = np.transpose(np.asarray(cam.R, np.float64))
    w2c[:3, 3] = np.asarray(cam.T, np.float64)
    c2w = np.linalg.inv(w2c)
    c2w[:3, 1:3] *= -1
    return c2w


def write_stage1_outputs(input_dir, scene):
    """scene: SyntheticScene.  The multi-view frames become train_data['frames'] (at least three: the reader takes the video's
    pose from frame 2), the video frames MOM/video/*.png.  Returns the path of train_data.pth."""
    from PIL import Image
    mom = os.path.join(input_dir, "MOM")
    os.makedirs(os.path.join(mom, "video"), exist_ok=True)
    views = list(scene._views)
    while len(views) < 3:
        views.append(views[-1])
    # frame 2 must carry the pose the video was rendered from (the centre view): SyntheticScene's view 0 is that pose
    views[0], views[2] = views[2], views[0]
    H, W = scene.H, scene.W
    ones = Image.fromarray(np.full((H, W, 3), 255, np.uint8))
    pts = np.asarray(scene.point_cloud.points, np.float32)
    data = {"camera_angle_x": scene.FovX, "camera_angle_y": scene.FovY, "W": W, "H": H,
            "pcd_points": pts.T.copy(), "pcd_colors": np.asarray(scene.point_cloud.colors, np.float32),
            "pcd_masks": np.ones((pts.shape[0], 3), np.float32), "frames": []}
    for cam in views:
        data["frames"].append({"image": _to_pil(cam._image_host), "transform_matrix": _c2w_opengl(cam).tolist(), "mask": ones,
                               "final_hint_start_x": [], "final_hint_start_y": [], "final_hint_end_x": [], "final_hint_end_y": [],
                               "T2C_flow": [], "our_flow": []})
    for f, cam in enumerate(scene._video):
        _to_pil(cam._image_host).save(os.path.join(mom, "video", f"{f:05d}.png"))
    path = os.path.join(mom, "train_data.pth")
    torch.save(data, path)
    torch.save(scene.scene_flow.clone(), os.path.join(mom, "scene_flow.pth"))
    return path


def check_stage1_dir(input_dir):
    """Raises with a precise message when <input_dir>/MOM is not a usable stage-1 output; returns (P, n_frames, n_video)."""
    from .dataset_readers import load_train_data
    mom = os.path.join(input_dir, "MOM")
    data = load_train_data(os.path.join(mom, "train_data.pth"))
    pts = np.asarray(data["pcd_points"])
    if pts.ndim != 2 or pts.shape[0] != 3:
        raise ValueError(f"pcd_points must be [3, P], got {pts.shape}")
    P = pts.shape[1]
    if np.asarray(data["pcd_colors"]).shape != (P, 3):
        raise ValueError("pcd_colors must be [P, 3]")
    flow = torch.load(os.path.join(mom, "scene_flow.pth"), map_location="cpu", weights_only=False)
    if tuple(flow.shape) != (3, P):
        raise ValueError(f"scene_flow.pth must be [3, {P}], got {tuple(flow.shape)}")
    video = [n for n in os.listdir(os.path.join(mom, "video")) if n.endswith((".jpg", ".jpeg", ".png"))]
    if not video:
        raise ValueError("MOM/video holds no frames")
    for fr in data["frames"]:
        if fr["image"].size != (data["W"], data["H"]):
            raise ValueError("frame image size differs from (W, H)")
        if np.asarray(fr["transform_matrix"]).shape != (4, 4):
            raise ValueError("transform_matrix must be 4x4")
    return P, len(data["frames"]), len(video)
