"""`basicsr.archs` plugin surface: ARCH_REGISTRY names + build_network
(reference `basicsr/archs/__init__.py:19-25`)."""
from copy import deepcopy

from ..registry import ARCH_REGISTRY
from .appmotioncodebook_arch import AppMotionCompFormer  # noqa: F401  (registers)
from .motion_estimator_arch import Motion_Estimator_keypoint_aware, KPDetector, DenseMotionNetwork  # noqa: F401  (register)
from .vqgan_arch import VQGANDiscriminator  # noqa: F401  (registers)

__all__ = ["build_network", "ARCH_REGISTRY"]


def build_network(opt):
    """deep-copies `opt`, pops 'type', instantiates ARCH_REGISTRY[type](**opt)."""
    opt = deepcopy(opt)
    network_type = opt.pop("type")
    return ARCH_REGISTRY.get(network_type)(**opt)
