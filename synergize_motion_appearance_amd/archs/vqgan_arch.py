"""`VQGANDiscriminator` -- the third name the shipped ymls register (reference `basicsr/archs/vqgan_arch.py:535-575`,
train.yml `network_d`): a PatchGAN stack conv4x4/s2 + LeakyReLU, (n_layers-1) x [conv4x4/s2 + BN + LeakyReLU],
conv4x4/s1 + BN + LeakyReLU, conv4x4/s1 -> 1 logit map.  Same constructor signature and state_dict layout
(`main.{i}.*`); the forward runs on the HIP implicit-GEMM kernel with BatchNorm in EVAL form (running statistics
folded into the conv).  The reference only ever calls it inside the training step, in train mode
(batch statistics + autograd): that form lives on the training tape (`trainer.TrainStep._disc`, the GAN branch of
`optimize_parameters`); calling this module's forward in train mode raises."""
import torch

from .. import ops
from ..ops import Conv, ACT_LRELU02, ACT_NONE
from ..registry import ARCH_REGISTRY
from ._base import HipArch


def discriminator_plan(nc, ndf, n_layers):
    """[(index in `main`, cin, cout, stride, has_bias, has_bn)] of the conv layers."""
    plan, idx, mult = [(0, nc, ndf, 2, True, False)], 2, 1
    for n in range(1, n_layers):
        prev, mult = mult, min(2 ** n, 8)
        plan.append((idx, ndf * prev, ndf * mult, 2, False, True))
        idx += 3
    prev, mult = mult, min(2 ** n_layers, 8)
    plan.append((idx, ndf * prev, ndf * mult, 1, False, True))
    plan.append((idx + 3, ndf * mult, 1, 1, True, False))
    return plan


def discriminator_manifest(nc, ndf, n_layers):
    out = []
    for i, cin, cout, _, has_bias, has_bn in discriminator_plan(nc, ndf, n_layers):
        out.append((f"main.{i}.weight", (cout, cin, 4, 4)))
        if has_bias:
            out.append((f"main.{i}.bias", (cout,)))
        if has_bn:
            out += [(f"main.{i + 1}.weight", (cout,)), (f"main.{i + 1}.bias", (cout,)), (f"main.{i + 1}.running_mean", (cout,)),
                    (f"main.{i + 1}.running_var", (cout,)), (f"main.{i + 1}.num_batches_tracked", ())]
    return out


@ARCH_REGISTRY.register()
class VQGANDiscriminator(HipArch):
    def __init__(self, nc=3, ndf=64, n_layers=4, model_path=None):
        self._plan = discriminator_plan(nc, ndf, n_layers)
        super().__init__(discriminator_manifest(nc, ndf, n_layers))
        with torch.no_grad():
            for i, _, _, _, _, has_bn in self._plan:
                if has_bn:                                   # nn.BatchNorm2d defaults
                    self.main._modules[str(i + 1)].weight.fill_(1.0)
                    self.main._modules[str(i + 1)].running_var.fill_(1.0)
        if model_path is not None:
            chkpt = torch.load(model_path, map_location="cpu")
            key = "params_d" if "params_d" in chkpt else "params" if "params" in chkpt else None
            if key is None:
                raise ValueError("Wrong params!")
            self.load_state_dict(chkpt[key])

    def engine(self):
        if self._engine is None:
            P = self._params_on_device()
            convs = []
            for i, cin, cout, stride, has_bias, has_bn in self._plan:
                w = P[f"main.{i}.weight"]
                b = P[f"main.{i}.bias"] if has_bias else torch.zeros(cout, device=w.device)
                if has_bn:                                   # eval BatchNorm (eps 1e-5) folded into the conv
                    s = P[f"main.{i + 1}.weight"] / torch.sqrt(P[f"main.{i + 1}.running_var"] + 1e-5)
                    w, b = w * s.view(-1, 1, 1, 1), (b - P[f"main.{i + 1}.running_mean"]) * s + P[f"main.{i + 1}.bias"]
                convs.append((Conv.from_torch(w, b), stride))
            self._engine = convs
        return self._engine

    @torch.no_grad()
    def forward(self, x):
        if self.training:
            raise NotImplementedError("VQGANDiscriminator in train mode (batch-statistics BatchNorm + gradients) runs inside the training "
                                      "step (trainer.TrainStep(net_d=...)._disc, SURVEY row N2); call .eval() for the stand-alone HIP forward")
        convs = self.engine()
        h = ops.nchw_to_nhwc(x.float())
        for k, (cv, stride) in enumerate(convs):
            Hh, Ww = h.shape[1], h.shape[2]
            oh, ow = (Hh + 2 - 4) // stride + 1, (Ww + 2 - 4) // stride + 1
            h = ops.conv(h, cv, stride=stride, pad=(1, 1), out_hw=(oh, ow), act=ACT_NONE if k == len(convs) - 1 else ACT_LRELU02)
        return ops.nhwc_to_nchw(h)
