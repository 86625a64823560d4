"""FourDGSdataset (reference scene/dataset.py:10-51): a list of CameraInfo that yields Camera objects.  The reference builds a
new Camera -- three matrix products and an inverse -- on every __getitem__ and puts `frame_num` on the GPU as a 0-d tensor
(dataset.py:39); here each Camera is built once and cached (train_4DGS.py:93 copies the whole list anyway) and frame_num stays
a python int, so that render() can form delta_scale * frame_num without a device read-back."""
from torch.utils.data import Dataset

from .cameras import Camera


class FourDGSdataset(Dataset):
    def __init__(self, dataset, args, dataset_type):
        self.dataset, self.args, self.dataset_type = dataset, args, dataset_type
        self._cams = {}

    def __getitem__(self, index):
        if self.dataset_type == "PanopticSports":
            return self.dataset[index]
        if isinstance(index, slice):
            return [self[i] for i in range(*index.indices(len(self)))]
        if index < 0:
            index += len(self)
        cam = self._cams.get(index)
        if cam is None:
            info = self.dataset[index]
            cam = self._cams[index] = Camera(colmap_id=index, R=info.R, T=info.T, FoVx=info.FovX, FoVy=info.FovY, image=info.image,
                                             gt_alpha_mask=None, image_name=f"{index}", uid=index,
                                             data_device=getattr(self.args, "data_device", "cuda"), time=info.time, mask=info.mask,
                                             frame_num=int(info.frame_num))
        return cam

    def __len__(self):
        return len(self.dataset)
