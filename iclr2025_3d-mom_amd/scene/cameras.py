"""Camera / MiniCam with the reference's attributes (reference scene/cameras.py:17-83)."""
import collections
import weakref

import numpy as np
import torch
from torch import nn

from ..utils.graphics_utils import getProjectionMatrix, getWorld2View2


class Camera(nn.Module):
    def __init__(self, colmap_id, R, T, FoVx, FoVy, image, gt_alpha_mask, image_name, uid, trans=np.array([0.0, 0.0, 0.0]),
                 scale=1.0, data_device="cuda", time=0, mask=None, frame_num=0, depth=None):
        super().__init__()
        self.uid, self.colmap_id, self.R, self.T = uid, colmap_id, R, T
        self.FoVx, self.FoVy, self.image_name, self.time = FoVx, FoVy, image_name, time
        try:
            self.data_device = torch.device(data_device)
        except Exception as e:
            print(e)
            print(f"[Warning] Custom device {data_device} failed, fallback to default cuda device")
            self.data_device = torch.device("cuda")
        self._image_host = image.clamp(0.0, 1.0)[:3, :, :]         # the master copy stays on the host like the reference's (:39-40)
        self.image_width, self.image_height = self._image_host.shape[2], self._image_host.shape[1]
        if gt_alpha_mask is not None:
            self._image_host = self._image_host * gt_alpha_mask
        self.depth, self.mask, self.frame_num = depth, mask, frame_num
        self.zfar, self.znear, self.trans, self.scale = 100.0, 0.01, trans, scale
        self.world_view_transform = torch.tensor(getWorld2View2(R, T, trans, scale)).transpose(0, 1)
        self.projection_matrix = getProjectionMatrix(znear=self.znear, zfar=self.zfar, fovX=FoVx, fovY=FoVy).transpose(0, 1)
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0))).squeeze(0)
        self.camera_center = self.world_view_transform.inverse()[3, :3]
        self._dev_cache = None
        self._gt_dev = None

    # `original_image` is what the reference's loop uploads every iteration (`viewpoint_cam.original_image.cuda()`,
    # train_4DGS.py:194: 6 MB over PCIe behind a pageable-memory staging copy at 960x540).  With a GPU present the attribute hands
    # out the device-resident copy of the bounded cache below, so that the script's own .cuda() is a no-op; .cpu(), indexing,
    # save_image and arithmetic work on it as on the host tensor.  GT_ON_DEVICE = False restores the host tensor.
    GT_ON_DEVICE = True

    @property
    def original_image(self):
        if Camera.GT_ON_DEVICE and self.data_device.type == "cuda" and torch.cuda.is_available():
            return self.device_tensors(self.data_device if self.data_device.index is not None
                                       else torch.device("cuda", torch.cuda.current_device()))[3]
        return self._image_host

    @original_image.setter
    def original_image(self, image):
        self._image_host = image.detach().to("cpu") if image.device.type != "cpu" else image
        self._drop_gt()

    # Ground-truth images resident on the device, over all cameras: bounded (least recently used first out), because a real
    # multi-view video is tens of GB while the three matrices of a camera are 140 bytes.  The reference uploads the image of
    # the drawn camera every iteration (train_4DGS.py:194); within the budget an image is uploaded once.
    GT_CACHE_BYTES = 24 << 30
    _gt_lru = collections.OrderedDict()        # id(camera) -> (weak reference to the camera, bytes)
    _gt_bytes = 0

    def device_tensors(self, device):
        """(view, full_proj, camera_center, gt image) on `device`: the matrices staged once per camera -- the reference
        re-uploads them every iteration (gaussian_renderer/__init__.py:49-52) -- the image through the bounded cache above."""
        device = torch.device(device)
        if self._dev_cache is None or self._dev_cache[0].device != device:
            # contiguous: world_view_transform is a transposed view, and the C ABI takes raw pointers
            self._dev_cache = (self.world_view_transform.to(device).contiguous(), self.full_proj_transform.to(device).contiguous(),
                               self.camera_center.to(device).contiguous())
            self._drop_gt()
        cls = Camera
        if self._gt_dev is None:
            img = self._image_host.to(device).contiguous()
            self._gt_dev = img
            if img.device.type != "cpu":
                nbytes = img.numel() * img.element_size()
                cls._gt_lru[id(self)] = (weakref.ref(self), nbytes)
                cls._gt_bytes += nbytes
                while cls._gt_bytes > cls.GT_CACHE_BYTES and len(cls._gt_lru) > 1:
                    key, (ref, nb) = next(iter(cls._gt_lru.items()))
                    victim = ref()
                    if victim is not None:
                        victim._drop_gt()
                    else:
                        del cls._gt_lru[key]
                        cls._gt_bytes -= nb
        elif id(self) in cls._gt_lru:
            cls._gt_lru.move_to_end(id(self))
        return self._dev_cache + (self._gt_dev,)

    def __del__(self):
        try:
            self._drop_gt()
        except Exception:
            pass

    def _drop_gt(self):
        ent = Camera._gt_lru.pop(id(self), None)
        if ent is not None:
            Camera._gt_bytes -= ent[1]
        self._gt_dev = None


class MiniCam:
    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform, time):
        self.image_width, self.image_height, self.FoVy, self.FoVx = width, height, fovy, fovx
        self.znear, self.zfar = znear, zfar
        self.world_view_transform, self.full_proj_transform = world_view_transform, full_proj_transform
        self.camera_center = torch.inverse(self.world_view_transform)[3][:3]
        self.time = time
