"""MI355X-native hot path of the 4D Gaussian Splatting train/render loop of cvsp-lab/ICLR2025_3D-MOM.

The directory name is not a Python identifier; load it with
    importlib.import_module("iclr2025_3d-mom_amd")
or call `install_dropin()` (below) and import the reference's own module names
(`diff_gaussian_rasterization`, `simple_knn`, `gaussian_renderer`, `scene.gaussian_model`, ...).
"""
import importlib
import sys

from . import _native
from ._native import build, lib, MomError  # noqa: F401

# reference module name -> module of this package.  Everything the reference's train_4DGS.py / render_4DGS.py import from
# the repository itself (third-party packages such as imageio / mmcv / torchvision are the caller's business).
_DROPIN = {
    "diff_gaussian_rasterization": ".diff_gaussian_rasterization",
    "diff_gaussian_rasterization._C": ".diff_gaussian_rasterization._C",
    "simple_knn": ".simple_knn",
    "simple_knn._C": ".simple_knn._C",
    "gaussian_renderer": ".gaussian_renderer",
    "gaussian_renderer.network_gui": ".gaussian_renderer.network_gui",
    "scene": ".scene",
    "scene.gaussian_model": ".scene.gaussian_model",
    "scene.deformation": ".scene.deformation",
    "scene.hexplane": ".scene.hexplane",
    "scene.regulation": ".scene.regulation",
    "scene.cameras": ".scene.cameras",
    "scene.dataset": ".scene.dataset",
    "scene.dataset_readers": ".scene.dataset_readers",
    "arguments": ".arguments",
    "utils": ".utils",
    "utils.loss_utils": ".utils.loss_utils",
    "utils.image_utils": ".utils.image_utils",
    "utils.general_utils": ".utils.general_utils",
    "utils.graphics_utils": ".utils.graphics_utils",
    "utils.sh_utils": ".utils.sh_utils",
    "utils.system_utils": ".utils.system_utils",
    "utils.camera_utils": ".utils.camera_utils",
    "utils.params_utils": ".utils.params_utils",
    "utils.timer": ".utils.timer",
    "utils.loader_utils": ".utils.loader_utils",
    "utils.scene_utils": ".utils.scene_utils",
}


def install_dropin(only=None) -> None:
    """Register this package's modules under the names the reference scripts import (`from scene import Scene, GaussianModel`,
    `from gaussian_renderer import render, network_gui`, `from utils.loss_utils import l1_loss, ssim`, ...), so that the
    reference's own train_4DGS.py / render_4DGS.py run against libmom4d unchanged.  `only`: an iterable of public names to
    register instead of all of them."""
    names = _DROPIN if only is None else {k: _DROPIN[k] for k in only}
    for public, rel in names.items():
        sys.modules[public] = importlib.import_module(rel, __name__)
