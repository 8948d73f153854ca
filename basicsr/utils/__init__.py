from synergize_motion_appearance_amd.img_util import (img2tensor, tensor2img, imwrite, imfrombytes,  # noqa: F401
                                                      mimsave)
from synergize_motion_appearance_amd.registry import (ARCH_REGISTRY, MODEL_REGISTRY, LOSS_REGISTRY,  # noqa: F401
                                                      METRIC_REGISTRY, DATASET_REGISTRY)

__all__ = ["img2tensor", "tensor2img", "imwrite", "imfrombytes", "mimsave"]
