"""utils/camera_utils.py of the reference: CameraInfo -> Camera, and the cameras.json entry."""
import numpy as np

from ..scene.cameras import Camera


def loadCam(args, id, cam_info, resolution_scale):
    return Camera(colmap_id=cam_info.uid, R=cam_info.R, T=cam_info.T, FoVx=cam_info.FovX, FoVy=cam_info.FovY, image=cam_info.image,
                  gt_alpha_mask=None, image_name=cam_info.image_name, uid=id, data_device=args.data_device, time=cam_info.time)


def cameraList_from_camInfos(cam_infos, resolution_scale, args):
    return [loadCam(args, i, c, resolution_scale) for i, c in enumerate(cam_infos)]


def camera_to_JSON(id, camera):
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = camera.R.transpose()
    Rt[:3, 3] = camera.T
    Rt[3, 3] = 1.0
    W2C = np.linalg.inv(Rt)
    from .graphics_utils import fov2focal
    return {'id': id, 'img_name': camera.image_name, 'width': camera.width, 'height': camera.height,
            'position': W2C[:3, 3].tolist(), 'rotation': [x.tolist() for x in W2C[:3, :3]],
            'fy': fov2focal(camera.FovY, camera.height), 'fx': fov2focal(camera.FovX, camera.width)}
