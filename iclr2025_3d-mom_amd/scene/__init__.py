"""Host-side mirror of the reference's `scene` package (only what the 4DGS train / render scripts reach): Scene (reference
scene/__init__.py:23-114), GaussianModel, and the synthetic stand-in for a stage-1 output."""
import os

from .dataset import FourDGSdataset
from .dataset_readers import sceneLoadTypeCallbacks
from .gaussian_model import GaussianModel  # noqa: F401
from .synthetic import SyntheticScene  # noqa: F401


def searchForMaxIteration(folder):
    return max(int(name.split("_")[-1]) for name in os.listdir(folder))


class Scene:
    """Loads a stage-1 output (TrainData_path = <input_dir>/MOM/train_data.pth), builds the camera sets and either initialises
    the Gaussians from the point cloud or loads a saved iteration (scene/__init__.py:27-94)."""

    def __init__(self, TrainData_path, Gaussian_path, args, gaussians, flow_scale=1, viewcrafter=False, load_iteration=None,
                 shuffle=True, resolution_scales=[1.0], load_coarse=False):
        self.model_path = Gaussian_path
        self.loaded_iter = None
        self.gaussians = gaussians
        if load_iteration:
            self.loaded_iter = (searchForMaxIteration(os.path.join(self.model_path, "point_cloud")) if load_iteration == -1
                                else load_iteration)
            print("Loading trained model at iteration {}".format(self.loaded_iter))
        info, time_line = sceneLoadTypeCallbacks["Blender"](TrainData_path, args.source_path, args.white_background, args.eval,
                                                            viewcrafter, args.extension)
        self.dataset_type = "blender"
        self.time_line = time_line
        self.maxtime = info.maxtime
        self.cameras_extent = info.nerf_normalization["radius"]
        print("Loading Training Cameras")
        self.train_camera = FourDGSdataset(info.train_cameras, args, self.dataset_type)
        self.train_camera_2 = FourDGSdataset(info.train_cameras_2, args, self.dataset_type)
        print("Loading Test Cameras")
        self.test_camera = FourDGSdataset(info.test_cameras, args, self.dataset_type)
        print("Loading Video Cameras")
        self.video_cameras_up = FourDGSdataset(info.video_cameras_up, args, self.dataset_type)
        self.video_cameras_side = FourDGSdataset(info.video_cameras_side, args, self.dataset_type)
        self.video_cameras_zoom = FourDGSdataset(info.video_cameras_zoom, args, self.dataset_type)
        self.video_cameras_circle = FourDGSdataset(info.video_cameras_circle, args, self.dataset_type)
        xyz_max = info.point_cloud.points.max(axis=0)
        xyz_min = info.point_cloud.points.min(axis=0)
        if getattr(args, "add_points", False):
            from .dataset_readers import add_points
            info = info._replace(point_cloud=add_points(info.point_cloud, xyz_max=xyz_max, xyz_min=xyz_min))
        self.gaussians._deformation.deformation_net.set_aabb(xyz_max, xyz_min)
        if self.loaded_iter:
            folder = os.path.join(self.model_path, "point_cloud", "iteration_" + str(self.loaded_iter))
            self.gaussians.load_ply(os.path.join(folder, "point_cloud.ply"))
            self.gaussians.load_model(folder)
        else:
            self.gaussians.create_from_pcd(info.point_cloud, self.cameras_extent, self.maxtime, TrainData_path, flow_scale)

    def save(self, iteration, stage):
        folder = os.path.join(self.model_path, "point_cloud/iteration_{}".format(iteration))
        self.gaussians.save_ply(os.path.join(folder, "point_cloud.ply"))
        self.gaussians.save_deformation(folder)

    def getTrainCameras(self, scale=1.0):
        return self.train_camera

    def getTrainCameras_2(self, scale=1.0):
        return self.train_camera_2

    def getTestCameras(self, scale=1.0):
        return self.test_camera

    def getTimeline(self):
        return self.time_line

    def getVideoCameras_up(self, scale=1.0):
        return self.video_cameras_up

    def getVideoCameras_side(self, scale=1.0):
        return self.video_cameras_side

    def getVideoCameras_zoom(self, scale=1.0):
        return self.video_cameras_zoom

    def getVideoCameras_circle(self, scale=1.0):
        return self.video_cameras_circle
