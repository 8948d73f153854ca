"""Drop-in alias of `basicsr.models` (reference `basicsr/models/__init__.py`): `build_model`, the
model registry and the inference surface of `AppMotionCompModel`."""
from synergize_motion_appearance_amd.models import build_model, AppMotionCompModel  # noqa: F401
from synergize_motion_appearance_amd.registry import MODEL_REGISTRY  # noqa: F401

__all__ = ["build_model"]
