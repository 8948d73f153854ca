"""Training-mode execution of the motion estimator on the HIP path with a gradient tape (reference
`Motion_Estimator_keypoint_aware.forward` archs/motion_estimator_arch.py:42-51 under `.train()`,
models/appmotioncomp_model.py:162, 305): keypoint detector (archs/keypoint_detector_arch.py:60-86) on the driving and the source
frames, dense motion (archs/dense_motion_arch.py:118-161).

What `.train()` changes against the inference engine (`engine_motion.py`): every BatchNorm of the two hourglasses normalises with the
statistics of the CURRENT batch (and moves its running buffers), so nothing can be folded into the convolutions; the hourglass runs
conv -> `bn_relu` -> pool / nearest-x2 as separate differentiable ops, and the skip concatenations are explicit `cat`s."""
import torch

from . import ops
from . import train_ops as T
from .manifest import hourglass_channels


class MotionTrainEngine:
    def __init__(self, common, dense, kp, buffers):
        """buffers: {name: tensor} of the module's buffers (BatchNorm running statistics, the two anti-alias kernels) -- updated in place."""
        self.K = common["num_kp"]
        self.nc = common["num_channels"]
        self.kp_cfg, self.dm_cfg = dict(kp), dict(dense)
        self.buf = buffers
        self.var = dense.get("kp_variance", 0.01)

    def _hourglass(self, tp, pre, x, nb):
        """Hourglass (utils/motion_estimator_util.py:440-492, 551-563): encoder outs [x, d0, ..], decoder with skip concats."""
        def block(p, t, up):
            h = T.conv(tp, t, p + ".conv.weight", p + ".conv.bias", up2=up)
            # num_batches_tracked stays untouched, as in the reference: its BatchNorm2d overrides forward() and calls F.batch_norm directly
            # (sync_batchnorm/batchnorm.py:48-53), which moves running_mean / running_var but never the counter
            return T.bn_relu(tp, h, p + ".norm.weight", p + ".norm.bias", self.buf.get(p + ".norm.running_mean"), self.buf.get(p + ".norm.running_var"))
        outs = [x]
        for i in range(nb):
            outs.append(T.avgpool2(tp, block(f"{pre}.encoder.down_blocks.{i}", outs[-1], False)))
        out = outs.pop()
        for j in range(nb):
            out = block(f"{pre}.decoder.up_blocks.{j}", out, True)
            out = T.cat(tp, [out, outs.pop()])
        return out

    def kp_detector(self, tp, image_nchw):
        """-> (value [B,K,2], jacobian [B,K,2,2]) tape tensors."""
        pre = "kp_detector."
        x = tp.stop(ops.antialias_down(image_nchw, self.buf[pre + "down.weight"].reshape(self.nc, 13, 13).contiguous()))
        fm = self._hourglass(tp, pre + "predictor", x, self.kp_cfg["num_blocks"])                   # [B,64,64,35]
        logits = T.conv(tp, fm, pre + "kp.weight", pre + "kp.bias", pad=(0, 0))                     # 7x7 valid -> [B,58,58,15]
        jm = T.conv(tp, fm, pre + "jacobian.weight", pre + "jacobian.bias", pad=(0, 0))             # [B,58,58,60]
        return T.kp_head(tp, logits, jm, self.K, self.kp_cfg["temperature"])

    def dense_motion(self, tp, source_nchw, kp_d, kp_s):
        """kp_d / kp_s: (value, jacobian) tape tensors -> (deformation [B,64,64,2], occlusion [B,64,64], drv_heat [B,64,64,K], aux)."""
        pre = "dense_motion_network."
        src64 = tp.stop(ops.antialias_down(source_nchw, self.buf[pre + "down.weight"].reshape(self.nc, 13, 13).contiguous()))
        hg_in, sparse, heat = T.sparse_motion(tp, src64, kp_d[0], kp_d[1], kp_s[0], kp_s[1], self.K, self.var)
        pred = self._hourglass(tp, pre + "hourglass", hg_in, self.dm_cfg["num_blocks"])            # [B,64,64,128]
        mask_l = T.conv(tp, pred, pre + "mask.weight", pre + "mask.bias")                           # 7x7 pad 3 -> [B,64,64,16]
        occ_l = T.conv(tp, pred, pre + "occlusion.weight", pre + "occlusion.bias")                  # [B,64,64,1]
        deform, occ = T.mask_deformation(tp, T.cat(tp, [mask_l, occ_l]), sparse, self.K + 1)
        return deform, occ, heat, {"sparse_motion": sparse, "src64": src64}
