"""Test-time data side of the animation entry: `build_dataset`,
`build_dataloader` and the (source frame, driving video, anchor) dataset `animate.py` iterates
(reference `basicsr/data/__init__.py:25-88`, `basicsr/data/frames_dataset.py:178-306`).

Same option keys, same returned dict, same quirks (the first frame of the driving folder is
skipped, `frames_dataset.py:252-255`; `video_name` strips 4 characters from each path component).
PNG frames are decoded by the in-tree codec (no cv2 in this image); `.jpg` folders are listed like
the reference does but cannot be decoded here and raise."""
import glob
import os
from copy import deepcopy

import numpy as np
import torch
from torch.utils.data import Dataset

from .img_util import imfrombytes, img2tensor, resize_linear
from .registry import DATASET_REGISTRY

__all__ = ["build_dataset", "build_dataloader"]


def build_dataset(dataset_opt):
    dataset_opt = deepcopy(dataset_opt)
    return DATASET_REGISTRY.get(dataset_opt["type"])(dataset_opt)


def build_dataloader(dataset, dataset_opt, num_gpu=1, dist=False, sampler=None, seed=None):
    """val/test phases only (batch 1, in order, no workers: `data/__init__.py:77-78`)."""
    phase = dataset_opt.get("phase", "test")
    if phase not in ("val", "test"):
        raise ValueError(f"Wrong dataset phase: {phase}. The MI355X-native build has the 'val' and 'test' loaders "
                         "(training is SURVEY row N2).")
    return torch.utils.data.DataLoader(dataset=dataset, batch_size=1, shuffle=False, num_workers=0,
                                       pin_memory=dataset_opt.get("pin_memory", False))


def _read(path):
    with open(path, "rb") as f:
        return imfrombytes(f.read(), float32=True)


def _normalize_(t, mean, std):
    m = torch.tensor(mean, dtype=t.dtype).view(-1, 1, 1)
    s = torch.tensor(std, dtype=t.dtype).view(-1, 1, 1)
    return t.sub_(m).div_(s)


@DATASET_REGISTRY.register()
class FramesMotionTransferTestDataset_CrossID_videopair_anchor(Dataset):
    """pairs csv columns: source (frame path), driving (folder of frames), optional anchor (frame
    path), optional anchor_idx (index into the returned driving list)."""

    def __init__(self, opt):
        super().__init__()
        import pandas as pd
        self.opt = opt
        self.root_dir = opt.get("root_dir")
        self.gt_size = opt.get("gt_size", 512)
        self.mean = opt.get("mean", [0.5, 0.5, 0.5])
        self.std = opt.get("std", [0.5, 0.5, 0.5])
        self.max_frame = opt.get("max_frame", None)
        pairs_list = opt.get("pairs_list", None)
        if pairs_list is None:
            raise NotImplementedError("Shoule provide cross id pairs for dataset.")
        pairs = pd.read_csv(pairs_list)
        self.source = pairs["source"].tolist()
        self.driving = pairs["driving"].tolist()
        self.anchors = pairs["anchor"].tolist() if "anchor" in pairs else None
        self.anchor_idx = pairs["anchor_idx"].tolist() if "anchor_idx" in pairs else None

    def __len__(self):
        return len(self.source)

    def __getitem__(self, idx):
        path_source, path_driving = self.source[idx], self.driving[idx]
        path_anchor = self.anchors[idx] if self.anchors is not None else None
        anchor_idx = self.anchor_idx[idx] if self.anchor_idx is not None else None
        video_name = (os.path.basename(os.path.dirname(path_source))[:-4] + "_" + os.path.basename(path_source)[:-4]
                      + "_" + os.path.basename(path_driving)[:-4])
        source = _read(path_source)
        driving, driving_name = [], []
        if os.path.isdir(path_driving):
            frames = sorted(glob.glob(path_driving + "/*.png"))
            if len(frames) == 0:
                frames = sorted(glob.glob(path_driving + "/*.jpg"))
            num_frames = len(frames)
            if self.max_frame is not None and num_frames - 1 > self.max_frame:
                num_frames = self.max_frame + 1
            for i in range(num_frames - 1):
                driving.append(_read(frames[i + 1]))
                driving_name.append(os.path.basename(frames[i + 1]))
        if path_anchor is not None:
            anchor = _read(path_anchor)
        else:
            anchor, anchor_idx = driving[0], 0
        if source.shape[-2] != self.gt_size:
            sz = (int(self.gt_size), int(self.gt_size))
            source = resize_linear(source, sz)
            driving = [resize_linear(f, sz) for f in driving]
            anchor = resize_linear(anchor, sz)
        source, anchor = img2tensor([np.ascontiguousarray(source), np.ascontiguousarray(anchor)], bgr2rgb=True, float32=True)
        driving = [img2tensor(np.ascontiguousarray(f), bgr2rgb=True, float32=True) for f in driving]
        _normalize_(source, self.mean, self.std)
        _normalize_(anchor, self.mean, self.std)
        for f in driving:
            _normalize_(f, self.mean, self.std)
        return {"source": source, "driving_video": driving, "anchor": anchor, "video_name": video_name,
                "driving_name_list": driving_name, "anchor_idx": anchor_idx}
