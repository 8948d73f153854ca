from synergize_motion_appearance_amd.registry import *  # noqa: F401,F403
from synergize_motion_appearance_amd.registry import Registry, ARCH_REGISTRY  # noqa: F401
