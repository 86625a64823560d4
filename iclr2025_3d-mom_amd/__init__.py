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

_DROPIN = {
    "diff_gaussian_rasterization": ".diff_gaussian_rasterization",
    "diff_gaussian_rasterization._C": ".diff_gaussian_rasterization._C",
}


def install_dropin(extra: bool = True) -> None:
    """Register this package's modules under the names the reference scripts import."""
    for public, rel in _DROPIN.items():
        sys.modules[public] = importlib.import_module(rel, __name__)
