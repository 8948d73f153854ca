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


def _list_driving_frames(folder, max_frame):
    """frame files of a driving folder in the reference's order and with its quirks (`frames_dataset.py:248-255`):
    png first, jpg only if no png; the FIRST file is skipped; at most `max_frame` frames are kept."""
    if not os.path.isdir(folder):
        return []
    files = sorted(glob.glob(os.path.join(folder, "*.png"))) or sorted(glob.glob(os.path.join(folder, "*.jpg")))
    files = files[1:]
    return files if max_frame is None else files[:max_frame]


class _FramePrep:
    """decode -> (resize to gt_size) -> CHW RGB float -> (x - mean) / std, one frame at a time."""

    def __init__(self, gt_size, mean, std):
        self.size = int(gt_size)
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
        self.resize = None                 # decided by the source frame, applied to every frame of the item

    def decode(self, path):
        with open(path, "rb") as f:
            return imfrombytes(f.read(), float32=True)

    def finish(self, img):
        if self.resize:
            img = resize_linear(img, (self.size, self.size))
        t = img2tensor(np.ascontiguousarray(img), bgr2rgb=True, float32=True)
        return t.sub_(self.mean).div_(self.std)


def _video_name(path_source, path_driving):
    """`<source folder minus 4 chars>_<source file minus 4 chars>_<driving folder minus 4 chars>`
    (`frames_dataset.py:244-246`: the reference strips a 4-character suffix from each component)."""
    parts = (os.path.basename(os.path.dirname(path_source)), os.path.basename(path_source), os.path.basename(path_driving))
    return "_".join(p[:-4] for p in parts)


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
        if opt.get("pairs_list") is None:
            raise NotImplementedError("a pairs_list csv (source, driving[, anchor, anchor_idx]) is required for this dataset")
        table = pd.read_csv(opt["pairs_list"])
        column = lambda name: table[name].tolist() if name in table else None
        self.source, self.driving = column("source"), column("driving")
        self.anchors, self.anchor_idx = column("anchor"), column("anchor_idx")

    def __len__(self):
        return len(self.source)

    def __getitem__(self, idx):
        src_path, drv_path = self.source[idx], self.driving[idx]
        prep = _FramePrep(self.gt_size, self.mean, self.std)
        src_img = prep.decode(src_path)
        prep.resize = src_img.shape[-2] != self.gt_size          # the reference tests the source only (`:268`)
        files = _list_driving_frames(drv_path, self.max_frame)
        item = {"source": prep.finish(src_img),
                "driving_video": [prep.finish(prep.decode(f)) for f in files],
                "video_name": _video_name(src_path, drv_path),
                "driving_name_list": [os.path.basename(f) for f in files]}
        if self.anchors is not None:
            item["anchor"] = prep.finish(prep.decode(self.anchors[idx]))
            item["anchor_idx"] = self.anchor_idx[idx] if self.anchor_idx is not None else None
        else:                                                     # no anchor column: the first driving frame
            item["anchor"], item["anchor_idx"] = item["driving_video"][0].clone(), 0
        return item


def _frame_name(path_source, path_driving):
    """`<src folder -4>_<src file -4>_<drv folder -4>_<drv file -4>` (`frames_dataset.py:367`)."""
    parts = (os.path.basename(os.path.dirname(path_source)), os.path.basename(path_source),
             os.path.basename(os.path.dirname(path_driving)), os.path.basename(path_driving))
    return "_".join(p[:-4] for p in parts)


@DATASET_REGISTRY.register()
class FramesMotionTransferTestDataset_PairsList(Dataset):
    """frame-PAIR evaluation set of `test.py` (reference `data/frames_dataset.py:309-399`): pairs csv with columns source, driving
    and optionally anchor (frame paths; no anchor column: the driving frame).  Every frame is resized on its OWN size test
    (`:372-384`, unlike the video dataset) and normalised; -> {'source', 'driving', 'anchor' [3,S,S], 'frame_name'}."""

    def __init__(self, opt):
        super().__init__()
        import pandas as pd
        self.opt = opt
        self.root_dir = opt.get("root_dir")
        self.gt_size = opt.get("gt_size", 512)
        self.mean = opt.get("mean", [0.5, 0.5, 0.5])
        self.std = opt.get("std", [0.5, 0.5, 0.5])
        if opt.get("pairs_list") is None:
            raise NotImplementedError("Shoule provide pairs_list for dataset.")
        table = pd.read_csv(opt["pairs_list"])
        self.source, self.driving = table["source"].tolist(), table["driving"].tolist()
        self.anchors = table["anchor"].tolist() if "anchor" in table else self.driving

    def __len__(self):
        return len(self.source)

    def __getitem__(self, idx):
        out = {}
        for key, path in (("source", self.source[idx]), ("driving", self.driving[idx]), ("anchor", self.anchors[idx])):
            prep = _FramePrep(self.gt_size, self.mean, self.std)
            img = prep.decode(path)
            prep.resize = img.shape[-2] != self.gt_size
            out[key] = prep.finish(img)
        out["frame_name"] = _frame_name(self.source[idx], self.driving[idx])
        return out
