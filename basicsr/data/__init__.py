"""Drop-in alias of `basicsr.data` (reference `basicsr/data/__init__.py`): test-phase
`build_dataset` / `build_dataloader` and the anchor video-pair dataset of `animate.py`."""
from synergize_motion_appearance_amd.data import build_dataset, build_dataloader  # noqa: F401
from synergize_motion_appearance_amd.data import FramesMotionTransferTestDataset_CrossID_videopair_anchor  # noqa: F401

__all__ = ["build_dataset", "build_dataloader"]
