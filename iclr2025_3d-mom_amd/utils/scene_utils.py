"""utils/scene_utils.py of the reference: render_training_image, the periodic debug dump of train_4DGS.py (only with
dataset.render_process): the rendered image next to the ground truth and a depth map, as <model_path>/<stage>_render/images."""
import os

import torch

from .image_io import save_image


@torch.no_grad()
def render_training_image(scene, gaussians, viewpoints, render_func, pipe, background, stage, iteration, time_now, dataset_type,
                          delta_scale=1):
    folder = os.path.join(scene.model_path, f"{stage}_render", "images")
    os.makedirs(folder, exist_ok=True)
    for idx, vp in enumerate(viewpoints):
        pkg = render_func(vp, gaussians, pipe, background, stage=stage.replace("test", "").replace("train", ""), cam_type=dataset_type,
                          delta_scale=delta_scale)
        image, depth = pkg["render"].clamp(0, 1), pkg["depth"]
        gt = vp.original_image.to(image.device)[:3]
        depth = (depth / (depth.max() + 1e-12)).repeat(3, 1, 1)
        save_image(torch.cat((gt, image, depth), dim=2), os.path.join(folder, f"{iteration}_{idx}.jpg".replace(".jpg", ".png")))
