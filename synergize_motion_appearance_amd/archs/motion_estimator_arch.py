"""`Motion_Estimator_keypoint_aware` -- drop-in for reference
`basicsr/archs/motion_estimator_arch.py:14-51` (+ KPDetector / DenseMotionNetwork it owns),
executing `engine_motion.MotionEngine` (HIP) instead of ATen modules."""
import torch

from ..engine_motion import MotionEngine, KPEngine, DenseEngine
from ..manifest import motion_estimator_manifest, kp_detector_manifest, dense_motion_manifest
from .. import ops
from ..registry import ARCH_REGISTRY
from ._base import HipArch


def _dense_outputs(r, src64, B, kp_driving, aux):
    """engine result -> the reference's out_dict (archs/dense_motion_arch.py:118-161), NCHW where it is NCHW there."""
    Fg = r["deformation"].shape[1]
    out = {"deformation": r["deformation"], "occlusion_map": r["occlusion_nhwc"].view(B, 1, Fg, Fg),
           "sparse_motion": r["sparse_motion"], "_heat_nhwc": r["heat_nhwc"]}
    out["driving_kp_heatmap"] = ops.nhwc_to_nchw(r["heat_nhwc"])
    if aux:
        out["mask"] = ops.nhwc_to_nchw(r["mask_nhwc"])
        hg = ops.nhwc_to_nchw(r["hg_in_nhwc"]).view(B, -1, 4, Fg, Fg)
        out["kp_heatmap"] = hg[:, :, 0]
        out["sparse_deformed"] = hg[:, :, 1:4]
        s = ops.nhwc_to_nchw(src64)
        out["source"] = s if s.shape[0] == B else s.expand(B, -1, -1, -1)
    return out


@ARCH_REGISTRY.register()
class KPDetector(HipArch):
    """standalone keypoint detector (reference `archs/keypoint_detector_arch.py:13-86`), same constructor signature;
    `forward(x, isSource=False)` -> {'value': [B,K,2], 'jacobian': [B,K,2,2]} on the HIP plan of rows A1-A3."""

    def __init__(self, block_expansion, num_kp, num_channels, max_features, num_blocks, temperature,
                 estimate_jacobian=False, scale_factor=1, single_jacobian_map=False, pad=0, model_path=None):
        if scale_factor != 0.25 or not estimate_jacobian or single_jacobian_map or pad != 0:
            raise NotImplementedError("only the options/test.yml keypoint-detector configuration has a HIP plan")
        self._cfg = ({"num_kp": num_kp, "num_channels": num_channels},
                     {"block_expansion": block_expansion, "max_features": max_features, "num_blocks": num_blocks,
                      "temperature": temperature, "estimate_jacobian": True, "scale_factor": scale_factor})
        super().__init__(kp_detector_manifest(*self._cfg))
        from ..synth import antialias_kernel
        with torch.no_grad():
            self.down.weight.copy_(antialias_kernel(num_channels, scale_factor))
        if model_path is not None:
            ck = torch.load(model_path, map_location="cpu")["kp_detector"]
            self.load_state_dict({k.replace("module.", ""): v for k, v in ck.items()})

    def engine(self):
        if self._engine is None:
            self._engine = KPEngine(self._params_on_device(), "", *self._cfg, mfma16=self.compute_dtype == "bf16")
        return self._engine

    @torch.no_grad()
    def forward(self, x, isSource=False):
        return self.engine().estimate_kp(x.float())


@ARCH_REGISTRY.register()
class DenseMotionNetwork(HipArch):
    """standalone dense-motion network (reference `archs/dense_motion_arch.py:12-161`), same constructor signature;
    `forward(source_image, kp_driving, kp_source)` -> the reference's out_dict on the HIP plan of rows A4-A6b."""

    def __init__(self, block_expansion, num_blocks, max_features, num_kp, num_channels, estimate_occlusion_map=False,
                 scale_factor=1, kp_variance=0.01, multi_mask=False, occlusion_num=5, model_path=None):
        if scale_factor != 0.25 or multi_mask or not estimate_occlusion_map:
            raise NotImplementedError("only the options/test.yml dense-motion configuration has a HIP plan")
        self._cfg = ({"num_kp": num_kp, "num_channels": num_channels},
                     {"block_expansion": block_expansion, "max_features": max_features, "num_blocks": num_blocks,
                      "estimate_occlusion_map": True, "scale_factor": scale_factor, "kp_variance": kp_variance})
        super().__init__(dense_motion_manifest(*self._cfg))
        from ..synth import antialias_kernel
        with torch.no_grad():
            self.down.weight.copy_(antialias_kernel(num_channels, scale_factor))
        if model_path is not None:
            pre = "module.dense_motion_network."
            ck = torch.load(model_path, map_location="cpu")["generator"]
            self.load_state_dict({k[len(pre):]: v for k, v in ck.items() if k.startswith(pre)})

    def engine(self):
        if self._engine is None:
            self._engine = DenseEngine(self._params_on_device(), "", *self._cfg, mfma16=self.compute_dtype == "bf16")
        return self._engine

    @torch.no_grad()
    def forward(self, source_image, kp_driving, kp_source):
        eng = self.engine()
        src64 = eng.source_down(source_image.float())
        B = kp_driving["value"].shape[0]
        return _dense_outputs(eng.dense_motion(src64, kp_driving, kp_source, want_aux=True), src64, B, kp_driving, True)


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
            self._engine = MotionEngine(self._params_on_device(), *self._cfg, mfma16=self.compute_dtype == "bf16")
            self._src_key = None
        return self._engine

    @torch.no_grad()
    def estimate_kp(self, image):
        """image [B,3,256,256] -> {'value': [B,15,2], 'jacobian': [B,15,2,2]}"""
        return self.engine().estimate_kp(image.float())

    def _source64(self, source_image):
        eng = self.engine()
        key = (tuple(source_image.shape), ops.fingerprint(source_image.float()))     # content key (see AppMotionCompFormer.encode_source)
        if key != self._src_key:
            self._src64 = eng.source_down(source_image.float())
            self._src_key = key
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
        out = _dense_outputs(r, src64, B, kp_driving, self.aux_outputs)
        out.update({"kp_driving": kp_driving, "kp_source": kp_source})
        return out

    @torch.no_grad()
    def forward(self, driving_image, source_image, only_return_kp_driving=False, relative=False):
        kp_driving = self.estimate_kp(driving_image)
        if only_return_kp_driving:
            return kp_driving
        kp_source = self.estimate_kp(source_image)
        return self.estimate_motion_w_kp(kp_source, kp_driving, source_image)
