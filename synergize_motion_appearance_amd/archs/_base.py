"""Common behaviour of the HIP-backed arch modules: parameters live in a ParamNode tree
(checkpoint contract); the packed-weight engine is rebuilt lazily whenever parameters may
have changed (load_state_dict, .cuda()/.to())."""
from torch import nn

from ..lib import SmxError
from ..paramtree import attach


class HipArch(nn.Module):
    def __init__(self, manifest):
        super().__init__()
        attach(self, manifest)
        self._engine = None
        self.compute_dtype = "f32"       # "f32": BASELINE configs[1] (reference arithmetic); "bf16": configs[2]

    def set_compute_dtype(self, name):
        """'f32' (default; fp32 storage, fp32 MFMA: the reference's arithmetic) or 'bf16' (BASELINE configs[2]: bf16
        activation / weight storage on the bf16 MFMA with fp32 accumulate; keypoint, flow, grid, normalisation-statistic and
        softmax math stay fp32).  Drops the packed engine; parameters keep their fp32 master copy."""
        if name not in ("f32", "bf16"):
            raise ValueError(f"compute dtype must be 'f32' or 'bf16', got {name!r}")
        if name != self.compute_dtype:
            self.compute_dtype = name
            self._engine = None
        return self

    # any of these may change parameter storage -> drop the packed engine
    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def refresh(self):
        """re-pack weights after in-place parameter edits."""
        self._engine = None

    def _params_on_device(self):
        P = dict(self.state_dict(keep_vars=False))
        dev = next(iter(P.values())).device
        if dev.type != "cuda":
            raise SmxError(f"{type(self).__name__}: parameters are on '{dev}'. The hot path runs as HIP kernels on an "
                           "MI355X only (call .cuda()); there is no CPU fallback.")
        return {k: (v.detach().float() if v.is_floating_point() else v) for k, v in P.items()}
