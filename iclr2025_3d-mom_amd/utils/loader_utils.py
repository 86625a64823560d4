"""utils/loader_utils.py of the reference, the two helpers train_4DGS.py imports.  Both are only reached with options that are
off in every shipped configuration (opt.zerostamp_init, opt.custom_sampler) and with a multi-view video dataset that carries
`.dataset.poses`; the stage-1 data contract of this path has neither."""
import random

import torch
from torch.utils.data.sampler import Sampler


def get_stamp_list(dataset, timestamp):
    poses = len(dataset.dataset.poses)
    frame_length = int(len(dataset) / poses)
    if timestamp > frame_length:
        raise IndexError("input timestamp bigger than total timestamp.")
    return [dataset[i * frame_length + timestamp] for i in range(poses)]


class FineSampler(Sampler):
    def __init__(self, dataset):
        self.len_dataset, self.len_pose = len(dataset), len(dataset.dataset.poses)
        self.frame_length = int(self.len_dataset / self.len_pose)
        sample_list = []
        for i in range(self.frame_length):
            for _ in range(4):
                idx = torch.randperm(self.len_pose) * self.frame_length + i
                now, cnt = [], 0
                for item in idx.tolist():
                    now.append(item)
                    cnt += 1
                    if cnt % 2 == 0 and len(sample_list) > 2:
                        now += random.sample(sample_list, 2)
            # once per frame, AFTER the four draws: the reference extends the list outside its `for j in range(4)` loop
            # (utils/loader_utils.py:27-42), so only the fourth permutation of a frame survives
            sample_list += now
        self.sample_list = sample_list

    def __iter__(self):
        return iter(self.sample_list)

    def __len__(self):
        return len(self.sample_list)
