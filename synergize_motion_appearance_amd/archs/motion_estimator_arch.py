"""`Motion_Estimator_keypoint_aware` -- drop-in for reference
`basicsr/archs/motion_estimator_arch.py:14-51` (+ KPDetector / DenseMotionNetwork it owns),
executing `engine_motion.MotionEngine` (HIP) instead of ATen modules."""
import torch

from ..engine_motion import MotionEngine
from ..manifest import motion_estimator_manifest
from .. import ops
from ..registry import ARCH_REGISTRY
from ._base import HipArch


@ARCH_REGISTRY.register()
class Motion_Estimator_keypoint_aware(HipArch):
    def __init__(self, common_params, dense_motion_params, kp_detector_params):
        if kp_detector_params is None:
            raise NotImplementedError("Shoule have kp_detector.")
        if dense_motion_params is None:
            raise NotImplementedError("Shoule have dense_motion_network.")
        self._cfg = (dict(common_params), dict(dense_motion_params), dict(kp_detector_params))
        dm, kp = self._cfg[1], self._cfg[2]
        if dm.get("scale_factor", 1) != 0.25 or kp.get("scale_factor", 1) != 0.25 or dm.get("multi_mask", False) \
                or not kp.get("estimate_jacobian", False) or not dm.get("estimate_occlusion_map", False):
            raise NotImplementedError("only the options/test.yml motion-estimator configuration has a HIP plan")
        super().__init__(motion_estimator_manifest(*self._cfg))
        from ..synth import antialias_kernel
        nc = self._cfg[0]["num_channels"]
        with torch.no_grad():                       # fixed Gaussian buffers (motion_estimator_util.py:603-632)
            self.kp_detector.down.weight.copy_(antialias_kernel(nc, 0.25))
            self.dense_motion_network.down.weight.copy_(antialias_kernel(nc, 0.25))
        self.aux_outputs = True                     # also return mask / sparse_deformed / kp_heatmap / source
        self._src_key, self._src64 = None, None

    def engine(self):
        if self._engine is None:
            self._engine = MotionEngine(self._params_on_device(), *self._cfg)
            self._src_key = None
        return self._engine

    @torch.no_grad()
    def estimate_kp(self, image):
        """image [B,3,256,256] -> {'value': [B,15,2], 'jacobian': [B,15,2,2]}"""
        return self.engine().estimate_kp(image.float())

    def _source64(self, source_image):
        key = (source_image.data_ptr(), source_image._version, tuple(source_image.shape))
        if key != self._src_key:
            self._src64 = self.engine().source_down(source_image.float())
            self._src_key = key
            self._src_ref = source_image      # keep alive so the address cannot be recycled under the cached key
        return self._src64

    @torch.no_grad()
    def estimate_motion_w_kp(self, kp_source, kp_driving, source_image):
        """-> dict with the reference's keys (deformation [B,64,64,2], occlusion_map [B,1,64,64],
        driving_kp_heatmap [B,15,64,64], sparse_motion, kp_driving, kp_source, ...)."""
        eng = self.engine()
        src64 = self._source64(source_image)
        B = kp_driving["value"].shape[0]
        if src64.shape[0] not in (1, B):
            raise ValueError("source batch must be 1 or equal to the driving batch")
        r = eng.dense_motion(src64, kp_driving, kp_source, want_aux=self.aux_outputs)
        out = {"deformation": r["deformation"], "occlusion_map": r["occlusion_nhwc"].view(B, 1, 64, 64),
               "sparse_motion": r["sparse_motion"], "_heat_nhwc": r["heat_nhwc"],
               "kp_driving": kp_driving, "kp_source": kp_source}
        out["driving_kp_heatmap"] = ops.nhwc_to_nchw(r["heat_nhwc"])
        if self.aux_outputs:
            out["mask"] = ops.nhwc_to_nchw(r["mask_nhwc"])
            hg = ops.nhwc_to_nchw(r["hg_in_nhwc"]).view(B, -1, 4, 64, 64)
            out["kp_heatmap"] = hg[:, :, 0]
            out["sparse_deformed"] = hg[:, :, 1:4]
            s = ops.nhwc_to_nchw(src64)
            out["source"] = s if s.shape[0] == B else s.expand(B, -1, -1, -1)
        return out

    @torch.no_grad()
    def forward(self, driving_image, source_image, only_return_kp_driving=False, relative=False):
        kp_driving = self.estimate_kp(driving_image)
        if only_return_kp_driving:
            return kp_driving
        kp_source = self.estimate_kp(source_image)
        return self.estimate_motion_w_kp(kp_source, kp_driving, source_image)
