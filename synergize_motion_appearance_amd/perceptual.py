"""MultiScalePyramidPerceptualLoss (reference `losses/losses.py:293-387`) on the HIP training tape: an anti-aliased image pyramid
(scales 1, 1/2, 1/4, 1/8), the VGG19 feature stack up to relu5_1 on every level of prediction and target, sum of weighted L1 feature
distances.  The VGG19 is a FROZEN feature extractor: its 13 convolutions run forward (fused ReLU) and data-gradient only
(`train_ops.conv(frozen=True)`), 3x3 ones on the fused Winograd kernel.

VGG19 itself is torchvision's (`models.vgg19(pretrained=True).features[:30]`, archs/vgg_arch.py:167-200: slices at relu1_1, relu2_1, relu3_1,
relu4_1, relu5_1) -- a third-party dependency absent from the reference tree and from this image; its published layout (configuration "E")
is restated in `VGG19_FEATURES`.  Its ImageNet weights are a download: pass them as a state dict keyed like torchvision's
(`features.<n>.weight` / `.bias`) or like the reference module's (`slice<k>.<n>.weight`); there is no built-in default and no silent
random fallback.
"""
import math

import torch

from . import train_ops as T
from .lib import ACT_RELU

# torchvision cfg "E" up to features[29]: (index, kind, cin, cout); the reference's five outputs are taken after these ReLU indices
VGG19_FEATURES = [(0, "conv", 3, 64), (2, "conv", 64, 64), (4, "pool", 0, 0), (5, "conv", 64, 128), (7, "conv", 128, 128), (9, "pool", 0, 0),
                  (10, "conv", 128, 256), (12, "conv", 256, 256), (14, "conv", 256, 256), (16, "conv", 256, 256), (18, "pool", 0, 0),
                  (19, "conv", 256, 512), (21, "conv", 512, 512), (23, "conv", 512, 512), (25, "conv", 512, 512), (27, "pool", 0, 0),
                  (28, "conv", 512, 512)]
VGG19_TAPS = (0, 5, 10, 19, 28)                                   # conv index whose ReLU output is h_relu{1..5}
_SLICE_OF = {0: 1, 2: 2, 5: 2, 7: 3, 10: 3, 12: 4, 14: 4, 16: 4, 19: 4, 21: 5, 23: 5, 25: 5, 28: 5}
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def vgg19_param_shapes(style="torchvision"):
    """[(name, shape)] of the 13 convolutions, torchvision (`features.N.*`) or reference-module (`sliceK.N.*`) naming."""
    out = []
    for n, kind, cin, cout in VGG19_FEATURES:
        if kind == "conv":
            pre = f"features.{n}" if style == "torchvision" else f"slice{_SLICE_OF[n]}.{n}"
            out += [(pre + ".weight", (cout, cin, 3, 3)), (pre + ".bias", (cout,))]
    return out


def antialias_kernel2d(scale):
    """the Gaussian of AntiAliasInterpolation2d (losses/losses.py:345-377): sigma = (1/scale - 1)/2, K = 2 round(4 sigma) + 1, normalised
    -> ([K, K] tensor, subsampling step)."""
    sigma = (1.0 / scale - 1.0) / 2.0
    K = 2 * round(sigma * 4) + 1
    ax = torch.arange(K, dtype=torch.float32)
    g1 = torch.exp(-(ax - (K - 1) / 2) ** 2 / (2 * sigma ** 2))
    k = g1.view(-1, 1) * g1.view(1, -1)
    return (k / k.sum()).contiguous(), int(1 / scale)


class PerceptualLoss:
    def __init__(self, vgg_state, scales=(1, 0.5, 0.25, 0.125), loss_weights=(1.0, 1.0, 1.0, 1.0, 1.0), device="cuda", prefix="vgg19."):
        """vgg_state: {name: tensor} with torchvision (`features.N.weight`) or reference (`sliceK.N.weight`) keys."""
        dev = torch.device(device)
        self.scales, self.loss_weights, self.prefix = [float(s) for s in scales], [float(w) for w in loss_weights], prefix
        if len(self.loss_weights) != len(VGG19_TAPS):
            raise ValueError("perceptual loss: five loss_weights (relu1_1 .. relu5_1)")
        self.P = {}
        for n, kind, cin, cout in VGG19_FEATURES:
            if kind != "conv":
                continue
            for leaf, shape in (("weight", (cout, cin, 3, 3)), ("bias", (cout,))):
                keys = (f"features.{n}.{leaf}", f"slice{_SLICE_OF[n]}.{n}.{leaf}")
                src = next((vgg_state[k] for k in keys if k in vgg_state), None)
                if src is None:
                    raise KeyError(f"perceptual loss: VGG19 state dict has neither {keys[0]} nor {keys[1]}")
                if tuple(src.shape) != shape:
                    raise ValueError(f"perceptual loss: {keys[0]} has shape {tuple(src.shape)}, VGG19 needs {shape}")
                self.P[f"{prefix}features.{n}.{leaf}"] = src.detach().to(dev, torch.float32).contiguous()
        std = torch.tensor(IMAGENET_STD, dtype=torch.float32)
        self.in_scale = (1.0 / std).to(dev)
        self.in_shift = (-torch.tensor(IMAGENET_MEAN, dtype=torch.float32) / std).to(dev)
        self.pyr = {}
        for s in self.scales:
            if s != 1:
                if abs(1 / s - round(1 / s)) > 1e-9:
                    raise ValueError("perceptual loss: pyramid scales must be 1 / integer")
                k, step = antialias_kernel2d(s)
                self.pyr[s] = (k.to(dev), step)

    def features(self, tp, x):
        """x NHWC [B,H,W,3] in the network's [-1, 1] range -- the reference feeds it to the ImageNet normalisation as is (vgg_arch.py:203)."""
        h = T.chan_affine(tp, x, self.in_scale, self.in_shift)
        outs = []
        for n, kind, _, _ in VGG19_FEATURES:
            if kind == "pool":
                h = T.maxpool2(tp, h)
            else:
                pre = f"{self.prefix}features.{n}"
                h = T.conv(tp, h, pre + ".weight", pre + ".bias", act=ACT_RELU, frozen=True)
                if n in VGG19_TAPS:
                    outs.append(h)
        return outs

    def __call__(self, tp, pred, target, weight=1.0):
        """pred: tape tensor NHWC; target: NHWC, takes no gradient.  -> [1] device scalar (weight * the reference's value_total)."""
        tgt = tp.stop(target)
        terms = []
        for s in self.scales:
            if s == 1:
                xp, xt = pred, tgt
            else:
                k, step = self.pyr[s]
                xp, xt = T.antialias(tp, pred, k, step), tp.stop(T.antialias(tp, tgt, k, step))
            fp, ft = self.features(tp, xp), self.features(tp, xt)
            for i, w in enumerate(self.loss_weights):
                if w != 0:
                    terms.append((T.l1_loss(tp, fp[i], tp.stop(ft[i]), w * weight), 1.0))
        return T.weighted_sum(tp, terms)


def synthetic_vgg19_state(style="torchvision"):
    """name-keyed synthetic VGG19 weights (benchmarks / tests: the real ones are a download): He-scaled, biases small."""
    from .synth import synth_tensor
    out = {}
    for name, shape in vgg19_param_shapes(style):
        t = synth_tensor("vgg19." + name.split(".", 1)[1] if style != "torchvision" else "vgg19." + name, shape)
        if name.endswith(".weight"):
            t = t * math.sqrt(2.0 / (shape[1] * 9)) / max(float(t.std()), 1e-6)
        else:
            t = t * 0.05
        out[name] = t
    return out
