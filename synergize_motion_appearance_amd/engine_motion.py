"""HIP execution plan for the motion estimator (rows A1-A6b of SURVEY.md section 8a):
keypoint detector (`archs/keypoint_detector_arch.py:60-86`) and dense motion
(`archs/dense_motion_arch.py:118-161`) of the reference, on NHWC activations.

Design points (MI355X-first, not a translation):
  * BatchNorm(eval) is folded into the preceding conv weights once at pack time;
  * every hourglass skip-concat buffer is allocated once per call and its two halves are
    written in place by their producers (decoder conv / avg-pool), so torch.cat never runs;
  * nearest x2 upsampling is folded into the consuming conv's gather (`up2`);
  * heatmaps + 16 sparse motions + 16 sparse warps are one kernel that writes the 64-channel
    hourglass input directly; mask softmax + flow blend are one kernel;
  * B driving frames run per launch (the reference loops B=1): the deep hourglass layers are
    weight-bandwidth bound at B=1 (207 MB of fp32 weights per frame).
"""

import torch

from . import ops
from .manifest import hourglass_channels
from .ops import Conv, ACT_RELU

HEADS_X3 = ops._knob("SMX_HEADS_X3", 1)      # configs[2]: the 7x7 heads in bf16x3 arithmetic on the bf16 MFMA (0 = fp32 implicit GEMM)


def _fold_bn(P, pre):
    """conv + BN(eval, eps 1e-5) -> one conv (sync_batchnorm/batchnorm.py:48-53 semantics)."""
    w, b = P[pre + ".conv.weight"], P[pre + ".conv.bias"]
    g, beta = P[pre + ".norm.weight"], P[pre + ".norm.bias"]
    rm, rv = P[pre + ".norm.running_mean"], P[pre + ".norm.running_var"]
    s = g / torch.sqrt(rv + 1e-5)
    return Conv.from_torch(w * s.view(-1, 1, 1, 1), (b - rm) * s + beta)


class HourglassPlan:
    def __init__(self, P, pre, block_expansion, in_features, num_blocks, max_features, mfma16=False):
        # mfma16 (configs[2]): the hourglass convolutions run on the bf16 MFMA (fp32-stored activations converted while
        # staging, bf16 weights, fp32 accumulate and fp32 outputs); the 7x7 keypoint / jacobian / mask / occlusion heads and
        # everything downstream of them (softmax expectation, sparse motions, flow blend) stay fp32
        self.m16 = mfma16
        self.down_ch, self.up_ch, self.out_filters = hourglass_channels(block_expansion, in_features, num_blocks, max_features)
        self.nb = num_blocks
        self.in_features = in_features
        self.down = [_fold_bn(P, f"{pre}.encoder.down_blocks.{i}") for i in range(num_blocks)]
        self.up = [_fold_bn(P, f"{pre}.decoder.up_blocks.{i}") for i in range(num_blocks)]

    def alloc_input(self, B, res, device):
        """the last concat buffer [B,res,res,cy+in] ; its tail slice is the hourglass input."""
        cy = self.up_ch[-1][1]
        c = cy + self.in_features
        if c % 4:       # zero pad channels up to a multiple of 4: the consuming 7x7 head gathers float4s (Cin 35 -> 36)
            buf = torch.zeros((B, res, res, c + (-c) % 4), device=device, dtype=torch.float32)
        else:
            buf = torch.empty((B, res, res, c), device=device, dtype=torch.float32)
        return buf, buf[..., cy:c]

    def run(self, final_buf, B, res):
        """final_buf from alloc_input with its tail filled -> same buffer, fully written."""
        nb = self.nb
        dev = final_buf.device
        # concat buffers cat[j] for decoder level j: [up_out_j | skip = encoder out (nb-1-j)]
        cats = [None] * nb
        cats[nb - 1] = final_buf
        for j in range(nb - 1):
            r = res >> (nb - 1 - j)
            cy = self.up_ch[j][1]
            cskip = self.down_ch[nb - 2 - j][1]
            cats[j] = torch.empty((B, r, r, cy + cskip), device=dev, dtype=torch.float32)
        # encoder: outs[0] = input slice, outs[i] (i>=1) pooled into the skip slice of cat[nb-1-i]
        cur = final_buf[..., self.up_ch[-1][1]:self.up_ch[-1][1] + self.in_features]
        r = res
        for i in range(nb):
            y = ops.conv(cur, self.down[i], act=ACT_RELU, mfma16=self.m16, out_dtype=torch.float32)
            r //= 2
            if i < nb - 1:
                j = nb - 2 - i
                dst = cats[j][..., self.up_ch[j][1]:]
                cur = ops.avgpool2(y, out=dst)
            else:
                cur = ops.avgpool2(y)
        # decoder
        out = cur
        for j in range(nb):
            ops.conv(out, self.up[j], out=cats[j][..., :self.up_ch[j][1]], up2=True, act=ACT_RELU, mfma16=self.m16)
            out = cats[j]
        return out


class KPEngine:
    """packed weights + plan of KPDetector (rows A1-A3); `pre` = parameter-name prefix ('' for the standalone arch)."""

    def __init__(self, P, pre, common, kp, mfma16=False):
        self.num_kp = common["num_kp"]
        self.temperature = kp["temperature"]
        self.hg = HourglassPlan(P, pre + "predictor", kp["block_expansion"], common["num_channels"], kp["num_blocks"], kp["max_features"], mfma16)
        # kp (15) and jacobian (60) heads share their input and geometry (7x7 valid): one conv with
        # N = 60 + 15 + 1 pad -> [jac | kp | 0]; 76 keeps the float4 reads of the jacobian maps aligned
        cin = P[pre + "kp.weight"].shape[1]
        padc = lambda w: torch.nn.functional.pad(w, (0, 0, 0, 0, 0, (-cin) % 4))      # zero weights for the pad channels  # noqa: E731
        kpc = Conv.from_torch(padc(P[pre + "kp.weight"]), P[pre + "kp.bias"])
        jc = Conv.from_torch(padc(P[pre + "jacobian.weight"]), P[pre + "jacobian.bias"])
        zero = Conv(torch.zeros_like(kpc.w[:1]), torch.zeros_like(kpc.b[:1]), 7, 7, kpc.cin, 1)
        self.head_conv = Conv.cat([jc, kpc, zero])
        self.n_jac = jc.cout
        self.down = P[pre + "down.weight"].reshape(common["num_channels"], 13, 13).contiguous()

    def estimate_kp(self, image_nchw):
        B = image_nchw.shape[0]
        r = image_nchw.shape[-1] // 4                         # scale_factor 0.25: 64 at img_size 256, 128 at 512 (N4)
        buf, inp = self.hg.alloc_input(B, r, image_nchw.device)
        ops.antialias_down(image_nchw, self.down, out=inp)
        fm = self.hg.run(buf, B, r)                           # [B,64,64,35 (+1 zero pad)]
        # 7x7 valid -> [B,58,58,76] = [jac 60 | kp 15 | 0]; configs[2]: bf16x3 on the bf16 MFMA (fp32-grade products, region-direct)
        heads = ops.conv7_x3(fm, self.head_conv, pad=0) if (self.hg.m16 and HEADS_X3) else ops.conv(fm, self.head_conv, pad=(0, 0))
        value, jac = ops.kp_head(heads[..., self.n_jac:self.n_jac + self.num_kp], heads[..., :self.n_jac],
                                 self.num_kp, self.temperature)
        return {"value": value, "jacobian": jac}


class DenseEngine:
    """packed weights + plan of DenseMotionNetwork (rows A4-A6b)."""

    def __init__(self, P, pre, common, dense, mfma16=False):
        self.num_kp = common["num_kp"]
        self.hg = HourglassPlan(P, pre + "hourglass", dense["block_expansion"],
                                (self.num_kp + 1) * (common["num_channels"] + 1), dense["num_blocks"], dense["max_features"], mfma16)
        # mask (16) and occlusion (1) heads: one stacked 7x7 conv, N = 17
        self.mo_conv = Conv.cat([Conv.from_torch(P[pre + "mask.weight"], P[pre + "mask.bias"]),
                                 Conv.from_torch(P[pre + "occlusion.weight"], P[pre + "occlusion.bias"])])
        self.down = P[pre + "down.weight"].reshape(common["num_channels"], 13, 13).contiguous()
        self.kp_variance = dense.get("kp_variance", 0.01)

    def source_down(self, source_nchw):
        """frame-invariant: anti-aliased 64x64 source, NHWC (dense_motion_arch.py:119-120)."""
        return ops.antialias_down(source_nchw, self.down)

    def dense_motion(self, src64, kp_driving, kp_source, want_aux=False):
        B = kp_driving["value"].shape[0]
        r = src64.shape[1]                                    # 64 (128 at img_size 512)
        buf, inp = self.hg.alloc_input(B, r, src64.device)
        sparse, heat = ops.sparse_motion(src64, kp_driving["value"], kp_driving["jacobian"].reshape(B, -1, 4),
                                         kp_source["value"], kp_source["jacobian"].reshape(kp_source["value"].shape[0], -1, 4),
                                         inp, B, self.num_kp, self.kp_variance)
        pred = self.hg.run(buf, B, r)                         # [B,64,64,128]
        # 7x7 pad 3 -> [B,64,64,17] = [mask 16 | occlusion logit]
        mlog = ops.conv7_x3(pred, self.mo_conv, pad=3) if (self.hg.m16 and HEADS_X3) else ops.conv(pred, self.mo_conv)
        deformation, mask, occ = ops.mask_deformation(mlog, sparse, want_mask=want_aux, K1=self.num_kp + 1, fused_occ=True)
        out = {"deformation": deformation, "occlusion_nhwc": occ, "heat_nhwc": heat, "sparse_motion": sparse}
        if want_aux:
            out["mask_nhwc"] = mask
            out["hg_in_nhwc"] = inp
        return out


class MotionEngine:
    """Motion_Estimator_keypoint_aware = KPDetector + DenseMotionNetwork."""

    def __init__(self, P, common, dense, kp, mfma16=False):
        self.kp = KPEngine(P, "kp_detector.", common, kp, mfma16)
        self.dm = DenseEngine(P, "dense_motion_network.", common, dense, mfma16)
        self.num_kp = common["num_kp"]

    def estimate_kp(self, image_nchw):
        return self.kp.estimate_kp(image_nchw)

    def source_down(self, source_nchw):
        return self.dm.source_down(source_nchw)

    def dense_motion(self, src64, kp_driving, kp_source, want_aux=False):
        return self.dm.dense_motion(src64, kp_driving, kp_source, want_aux)
