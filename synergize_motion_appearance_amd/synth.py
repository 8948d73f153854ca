"""Deterministic synthetic weights and clips (no checkpoint, no dataset, no network).

Both sides of every parity check fill a state_dict from the *name and shape* of each
entry, so no tensor file has to travel: the golden generator applies `synth_state_dict`
to the reference modules' own state_dict, the tests/bench apply it to ours.  The names
are identical on both sides by the checkpoint contract (SURVEY.md section 8b;
reference `basicsr/demo.py:46-72` strict load).

Default PyTorch init must not be used for fixtures: BN is identity, jacobians are exactly
I, position embeddings are zero (reference `archs/appmotioncodebook_arch.py:266-267`,
`archs/keypoint_detector_arch.py:33-34`).
"""
import math
import zlib

import torch

__all__ = ["synth_tensor", "synth_state_dict", "synth_input", "synth_clip", "synth_keypoints", "antialias_kernel"]


def antialias_kernel(channels: int = 3, scale: float = 0.25) -> torch.Tensor:
    """The fixed Gaussian buffer `down.weight` ([C,1,13,13] for scale 0.25).

    sigma = (1/scale - 1)/2, size = 2*round(4 sigma)+1, separable, normalised to sum 1
    (reference `utils/motion_estimator_util.py:603-632`)."""
    sigma = (1.0 / scale - 1.0) / 2.0
    size = 2 * round(sigma * 4) + 1
    ax = torch.arange(size, dtype=torch.float32)
    mean = (size - 1) / 2
    g1 = torch.exp(-(ax - mean) ** 2 / (2 * sigma ** 2))
    k = g1.view(-1, 1) * g1.view(1, -1)
    k = k / torch.sum(k)
    return k.view(1, 1, size, size).repeat(channels, 1, 1, 1)


def _gen(name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode("utf-8")) & 0x7FFFFFFF)
    return g


_BIG_MEAN_BIAS = (".conv1.bias", ".conv2.bias", ".conv_out.bias", ".conv.bias")
_BIG_MEAN_SCOPE = ("encoder.blocks.", "generator.blocks.", "fuse_convs_dict.", "to_motion.1.", "motion_emb.2.")


def synth_tensor(name: str, shape, dtype=torch.float32, style: str = "default") -> torch.Tensor:
    """Value of state_dict entry `name` with `shape` -- a pure function of (name, shape, style).

    style "checkpoint": the statistics a TRAINED checkpoint has and the default synthesis avoids on purpose -- convolutions in front
    of a GroupNorm carry a large common bias (4 + 0.3 N: |group mean| >> std at the normalisation, where E[x^2] - mean^2 cancels in
    fp32), BatchNorm running statistics far from (0, 1) (mean ~ N, var in [e^-1, e^1]), and the codebooks keep the reference's
    own init U(-1/K, 1/K) (archs/vqgan_arch.py:31: top-2 distance gaps down to fp32 noise, i.e. near-ties)."""
    shape = tuple(shape)
    g = _gen(name)
    leaf = name.rsplit(".", 1)[-1]
    if style == "checkpoint":
        if leaf == "running_var":
            return torch.exp(2.0 * torch.rand(shape, generator=g) - 1.0).to(dtype)
        if leaf == "running_mean":
            return (1.0 * torch.randn(shape, generator=g)).to(dtype)
        if name.endswith("embedding.weight"):
            return ((2.0 * torch.rand(shape, generator=g) - 1.0) / shape[0]).to(dtype)
        if name.startswith(_BIG_MEAN_SCOPE) and name.endswith(_BIG_MEAN_BIAS):
            return (4.0 + 0.3 * torch.randn(shape, generator=g)).to(dtype)
    elif style != "default":
        raise ValueError(f"unknown synthesis style {style!r}")
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if name.endswith("down.weight"):
        return antialias_kernel(shape[0], 0.25).to(dtype)
    if leaf == "running_var":
        return (0.75 + 0.5 * torch.rand(shape, generator=g)).to(dtype)
    if leaf == "running_mean":
        return (0.1 * torch.randn(shape, generator=g)).to(dtype)
    if name.startswith("position_emb_"):
        return (0.2 * torch.randn(shape, generator=g)).to(dtype)
    if name.endswith("embedding.weight"):  # codebooks: N(0,1) keeps VQ top-2 gaps wide
        return torch.randn(shape, generator=g).to(dtype)
    if name == "kp_detector.jacobian.bias":
        base = torch.tensor([1.0, 0.0, 0.0, 1.0]).repeat(shape[0] // 4)
        return (base + 0.05 * torch.randn(shape, generator=g)).to(dtype)
    if len(shape) == 1:
        is_norm = ("norm" in name) or name.startswith("to_motion.2.") or \
            name.endswith("blocks.17.weight") or name.endswith("blocks.17.bias")
        if leaf == "weight" and is_norm:
            return (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
        return (0.05 * torch.randn(shape, generator=g)).to(dtype)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    gain = 1.0
    if name == "kp_detector.jacobian.weight":
        gain = 0.2      # jacobians = I + O(0.1): invertible, non-trivial
    elif name.startswith("refine.conv2.") or name.startswith("refine.convo2."):
        gain = 0.5      # residual flow of a fraction of a pixel @64^2, like a trained net
    elif name == "generator.blocks.18.weight":
        gain = 0.3      # keep most output pixels inside [-1,1] (uint8 path not all-saturated)
    return (torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))).to(dtype)


def synth_input(name: str, shape) -> torch.Tensor:
    """N(0,1) test input keyed by name (activations, not weights)."""
    return torch.randn(tuple(shape), generator=_gen("input:" + name))


def synth_state_dict(manifest, style: str = "default"):
    """manifest: iterable of (name, shape) or a state_dict -> OrderedDict name -> tensor."""
    from collections import OrderedDict
    items = manifest.items() if hasattr(manifest, "items") else manifest
    out = OrderedDict()
    for name, v in items:
        shape = tuple(v.shape) if hasattr(v, "shape") else tuple(v)
        out[name] = synth_tensor(name, shape, style=style)
    return out


def _smooth_image(g: torch.Generator, size: int = 256) -> torch.Tensor:
    """Face-like statistics: low-frequency blobs + a little texture, in [-1, 1]."""
    img = torch.zeros(3, size, size)
    for res, amp in ((4, 0.55), (8, 0.35), (16, 0.2), (32, 0.1), (64, 0.04)):
        z = torch.randn(1, 3, res, res, generator=g)
        img += amp * torch.nn.functional.interpolate(z, size=(size, size), mode="bicubic",
                                                     align_corners=True)[0]
    img += 0.01 * torch.randn(3, size, size, generator=g)
    return img.clamp_(-1.0, 1.0)


def synth_clip(n_frames: int, seed: int = 123, size: int = 256):
    """(source [3,S,S], driving [n,3,S,S]) fp32 in [-1,1] like `demo.py:177-185`.

    Driving frame t = source warped by a smooth seeded affine trajectory + 0.02 noise, so
    keypoints move coherently (pure-noise frames give degenerate keypoints)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    src = _smooth_image(g, size)
    phase = 2 * math.pi * torch.rand(6, generator=g)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
    base = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1)            # [S,S,3]
    frames = []
    for t in range(n_frames):
        w = 2 * math.pi * (t + 1) / max(n_frames, 30)
        d = 0.06 * torch.sin(w + phase)
        theta = torch.tensor([[1.0 + d[0], d[1], 0.8 * d[2]],
                              [d[3], 1.0 + d[4], 0.8 * d[5]]])
        grid = (base @ theta.t()).unsqueeze(0)                            # [1,S,S,2]
        f = torch.nn.functional.grid_sample(src.unsqueeze(0), grid, mode="bilinear",
                                            padding_mode="border", align_corners=True)[0]
        f = f + 0.02 * torch.randn(3, size, size, generator=g)
        frames.append(f.clamp_(-1.0, 1.0))
    return src.contiguous(), torch.stack(frames).contiguous()


def synth_keypoints(batch: int, num_kp: int = 15, seed: int = 7):
    """Synthetic-keypoint mode: kp_s ~ U(-0.6,0.6), kp_d = kp_s+0.08 N,
    jacobians I + 0.1 N. About 2% of flow samples land outside [-1,1]."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    eye = torch.eye(2).view(1, 1, 2, 2)
    kp_s = {"value": (torch.rand(1, num_kp, 2, generator=g) * 1.2 - 0.6).repeat(batch, 1, 1),
            "jacobian": (eye + 0.1 * torch.randn(1, num_kp, 2, 2, generator=g)).repeat(batch, 1, 1, 1)}
    kp_d = {"value": kp_s["value"] + 0.08 * torch.randn(batch, num_kp, 2, generator=g),
            "jacobian": eye + 0.1 * torch.randn(batch, num_kp, 2, 2, generator=g)}
    return kp_s, kp_d
