from synergize_motion_appearance_amd.archs import build_network, ARCH_REGISTRY  # noqa: F401
from synergize_motion_appearance_amd.archs import AppMotionCompFormer, Motion_Estimator_keypoint_aware  # noqa: F401

__all__ = ["build_network"]
